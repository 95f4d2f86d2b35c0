"""torch.autograd wrappers around the libhmx.so C ABI.

Every function here requires CUDA (ROCm) fp32 tensors and calls the HIP kernels; there is
no eager/CPU fallback -- a missing library or a wrong device raises.
"""
import os

import torch

from . import hmx

CHECK_IDS = os.environ.get("HMX_CHECK_IDS", "0") == "1"


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise hmx.HmxError("hmx ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _f32c(t):
    if t.dtype != torch.float32:
        raise hmx.HmxError(f"expected float32, got {t.dtype}")
    return t.contiguous()


def _i64c(t):
    if t.dtype != torch.int64:
        t = t.long()
    return t.contiguous()


def _validate_ids(ids, n_rows, what):
    """Opt-in host check reproducing nn.Embedding's IndexError (costs a device sync)."""
    if CHECK_IDS and ids.numel():
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= n_rows:
            raise IndexError(f"{what}: index out of range in self (min {lo}, max {hi}, rows {n_rows})")


_FUSE_TABLE_GRAD = False


def set_fused_table_grad(flag=True):
    """Opt-in: the backward of the row gathers (and of the fused loss's hard-negative gather) adds straight into
    `weight.grad` of a leaf table instead of returning a dense (rows, d) tensor for autograd to add.  Same sums in
    the same order (0 + a + b + ... per row), but one zero-fill per table and step instead of one per gather plus a
    dense add each (2 GB of traffic per extra gather at 4 M x 64).  Leave it off when gradients are taken with
    torch.autograd.grad(), which only sees returned values."""
    global _FUSE_TABLE_GRAD
    _FUSE_TABLE_GRAD = bool(flag)


class TableGrad:
    """Book-keeping of a table's dense gradient in fused-table-gradient mode.  The buffer outlives the step: one byte
    per row (`flags`) records which rows the step's scatter-adds wrote, the optimizer (optim.HmxAdamW) reads only
    those rows, sets them back to zero and clears their flags, and declares the buffer `clean` -- it is then handed out
    again as the next step's zero-filled gradient without a fill.  Anything else that writes the buffer must go
    through scatter_add_rows / the fused loss (they set the flags) or call mark_dense()."""

    def __init__(self, param):
        self.buf = torch.zeros_like(param, memory_format=torch.contiguous_format)
        self.flags = torch.zeros(param.shape[0], dtype=torch.uint8, device=param.device)
        self.clean = False          # True: buf is all zero and flags are all clear (set by the optimizer)
        self.dense = False          # True: rows were written without flags; consumers must read every row

    def matches(self, param):
        return self.buf.shape == param.shape and self.buf.device == param.device

    def mark_dense(self):
        self.dense = True


_TABLE_GRADS = {}        # id(param) -> (weakref to the parameter, TableGrad)
_FLAGS_BY_PTR = {}       # data_ptr of a tracked buffer -> TableGrad


def table_grad_state(param):
    """The TableGrad whose buffer is param.grad right now, or None."""
    ent = _TABLE_GRADS.get(id(param))
    if ent is None or ent[0]() is not param or param.grad is None or param.grad.data_ptr() != ent[1].buf.data_ptr():
        return None
    return ent[1]


def _into_param_grad(param):
    """The tensor a fused table gradient should be added into, or None when the standard path applies."""
    if not (_FUSE_TABLE_GRAD and isinstance(param, torch.nn.Parameter) and param.is_leaf and param.requires_grad):
        return None
    if param.grad is None:
        import weakref
        ent = _TABLE_GRADS.get(id(param))
        st = ent[1] if (ent is not None and ent[0]() is param and ent[1].matches(param)) else None
        if st is None:
            for k in [k for k, (ref, _) in _TABLE_GRADS.items() if ref() is None]:      # entries of parameters that are gone
                _FLAGS_BY_PTR.pop(_TABLE_GRADS.pop(k)[1].buf.data_ptr(), None)
            st = TableGrad(param)
            _TABLE_GRADS[id(param)] = (weakref.ref(param), st)
            _FLAGS_BY_PTR[st.buf.data_ptr()] = st
        elif not st.clean:         # the last step's gradient was not consumed by an optimizer that cleans up
            st.buf.zero_()
            st.flags.zero_()
        st.clean, st.dense = False, False
        param.grad = st.buf
    g = param.grad
    return g if (g.is_contiguous() and g.dtype == torch.float32) else None


def _row_flags_of(out):
    """The row flags that go with a tracked table-gradient buffer (None for any other tensor)."""
    st = _FLAGS_BY_PTR.get(out.data_ptr()) if out is not None else None
    return st.flags if (st is not None and st.buf.shape == out.shape) else None


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def scatter_add_rows(grad_rows, ids, n_rows, padding_idx=-1, token_type=None, want_type=0, out=None):
    """Dense (n_rows, d) gradient: out[ids[i]] += grad_rows[i] (deterministic).  When `out` is a table gradient kept
    by TableGrad, the rows written are flagged."""
    _need_cuda(grad_rows, ids, token_type)
    d = grad_rows.shape[-1]
    g = _f32c(grad_rows).view(-1, d)
    ids = _i64c(ids).view(-1)
    tt = _i64c(token_type).view(-1) if token_type is not None else None
    n = ids.numel()
    assert g.shape[0] == n
    if out is None:
        out = torch.zeros(n_rows, d, dtype=torch.float32, device=g.device)
    L = hmx.lib()
    ws = workspace(L.hmx_scatter_add_rows_workspace_bytes(n, n_rows, d), g.device)
    hmx.check(L.hmx_scatter_add_rows_f32(hmx.ptr(g), hmx.ptr(ids), hmx.ptr(tt), want_type, n, d, hmx.ptr(out), n_rows,
                                         padding_idx, hmx.ptr(_row_flags_of(out)), hmx.ptr(ws), ws.numel(),
                                         hmx.current_stream()), "hmx_scatter_add_rows_f32")
    return out


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, ids, padding_idx):
        _need_cuda(weight, ids)
        w = _f32c(weight)
        idc = _i64c(ids)
        _validate_ids(idc, w.shape[0], "gather_rows")
        out = torch.empty(*idc.shape, w.shape[1], dtype=torch.float32, device=w.device)
        hmx.check(hmx.lib().hmx_gather_rows_f32(hmx.ptr(w), w.shape[0], w.shape[1], hmx.ptr(idc), idc.numel(),
                                                hmx.ptr(out), hmx.current_stream()), "hmx_gather_rows_f32")
        ctx.save_for_backward(idc)
        ctx.n_rows, ctx.padding_idx = w.shape[0], -1 if padding_idx is None else padding_idx
        ctx.param = weight
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idc,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        acc = _into_param_grad(ctx.param)
        gw = scatter_add_rows(grad_out, idc, ctx.n_rows, ctx.padding_idx, out=acc)
        return (None if acc is not None else gw), None, None


def gather_rows(weight, ids, padding_idx=None):
    """out[..., :] = weight[ids[...], :]  (nn.Embedding forward; dense deterministic backward)."""
    return _GatherRows.apply(weight, ids, padding_idx)


@torch.no_grad()
def gather_rows_bf16(weight, ids):
    """out[..., :] = bf16(weight[ids[...], :]) (round to nearest even) as a torch.bfloat16 tensor; no gradient."""
    _need_cuda(weight, ids)
    w = _f32c(weight.detach())
    idc = _i64c(ids)
    _validate_ids(idc, w.shape[0], "gather_rows_bf16")
    out = torch.empty(*idc.shape, w.shape[1], dtype=torch.bfloat16, device=w.device)
    hmx.check(hmx.lib().hmx_gather_rows_bf16(hmx.ptr(w), w.shape[0], w.shape[1], hmx.ptr(idc), idc.numel(),
                                             hmx.ptr(out), hmx.current_stream()), "hmx_gather_rows_bf16")
    return out


@torch.no_grad()
def gather_gelu_rows(weight, ids):
    """out[..., :] = GELU(weight[ids[...], :]) in one pass (candidate-embedding generation; no gradient)."""
    _need_cuda(weight, ids)
    w = _f32c(weight.detach())
    idc = _i64c(ids)
    _validate_ids(idc, w.shape[0], "gather_gelu_rows")
    out = torch.empty(*idc.shape, w.shape[1], dtype=torch.float32, device=w.device)
    hmx.check(hmx.lib().hmx_gather_gelu_rows_f32(hmx.ptr(w), w.shape[0], w.shape[1], hmx.ptr(idc), idc.numel(),
                                                 hmx.ptr(out), hmx.current_stream()), "hmx_gather_gelu_rows_f32")
    return out


class _GatherRouted(torch.autograd.Function):
    @staticmethod
    def forward(ctx, item_w, user_w, ids, token_type):
        _need_cuda(item_w, user_w, ids, token_type)
        iw, uw = _f32c(item_w), _f32c(user_w)
        idc, tt = _i64c(ids), _i64c(token_type)
        assert idc.shape == tt.shape and iw.shape[1] == uw.shape[1]
        out = torch.empty(*idc.shape, iw.shape[1], dtype=torch.float32, device=iw.device)
        hmx.check(hmx.lib().hmx_gather_routed_f32(hmx.ptr(iw), iw.shape[0], hmx.ptr(uw), uw.shape[0], iw.shape[1],
                                                  hmx.ptr(idc), hmx.ptr(tt), idc.numel(), hmx.ptr(out),
                                                  hmx.current_stream()), "hmx_gather_routed_f32")
        ctx.save_for_backward(idc, tt)
        ctx.rows = (iw.shape[0], uw.shape[0])
        ctx.params = (item_w, user_w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idc, tt = ctx.saved_tensors
        grads = []
        for which in (0, 1):
            if not ctx.needs_input_grad[which]:
                grads.append(None)
                continue
            acc = _into_param_grad(ctx.params[which])
            g = scatter_add_rows(grad_out, idc, ctx.rows[which], 0, tt, which + 1, out=acc)
            grads.append(None if acc is not None else g)
        return grads[0], grads[1], None, None


def gather_routed(item_weight, user_weight, ids, token_type):
    """embedding.py:436-450: item rows where token_type==1, user rows where ==2, zeros elsewhere."""
    return _GatherRouted.apply(item_weight, user_weight, ids, token_type)


# --------------------------------------------------------------------------- fused InfoNCE
FUSE_NCE_QUERY_GRAD = True    # the forward's key walk also accumulates d loss / d q (False: the backward walks the keys again)


class _FusedNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, table, hard_ids, inb_idx, temperature, normalize, hard_groups, n_hard, inb_skip_self):
        _need_cuda(q, k, table, hard_ids, inb_idx)
        q, k = _f32c(q), _f32c(k)
        m, d = q.shape
        assert k.shape == (m, d)
        hard_ld = 0 if hard_ids is None else hard_ids.shape[1]
        n_hard = hard_ld if n_hard is None else int(n_hard)
        n_inb = 0 if inb_idx is None else inb_idx.shape[1]
        tab = _f32c(table.detach()) if (table is not None and n_hard > 0) else None
        hid = _i64c(hard_ids) if n_hard > 0 else None
        flags = 0
        iid = None
        if n_inb > 0:
            if inb_idx.dtype == torch.int32:
                iid, flags = inb_idx.contiguous(), 1 | (2 if inb_skip_self else 0)
            else:
                if inb_skip_self:
                    raise hmx.HmxError("fused_nce_rows: inb_skip_self needs int32 draws")
                iid = _i64c(inb_idx)
        dev = q.device
        logits = torch.empty(m, 1 + n_hard + n_inb, dtype=torch.float32, device=dev)
        row_loss, lse, q_inv, k_inv = (torch.empty(m, dtype=torch.float32, device=dev) for _ in range(4))
        n_rows = 0 if tab is None else tab.shape[0]
        # with a gradient wanted, the forward's key walk also accumulates d loss_i / d q_i; the backward then needs no walk
        dq_unit = torch.empty_like(q) if (FUSE_NCE_QUERY_GRAD and any(ctx.needs_input_grad[:3])) else None
        groups = None
        if hard_groups is not None and n_hard > 0:
            gs, period = hard_groups
            gs = _i64c(gs)
            _need_cuda(gs)
            if not (1 <= int(period) <= min(64, n_hard, hard_ld)) or gs.numel() < 2:
                raise hmx.HmxError("fused_nce_rows: hard_groups needs 1 <= period <= min(64, n_hard, row length) and >= 1 group")
            groups = (gs, int(period))
        elif n_hard != hard_ld:
            raise hmx.HmxError("fused_nce_rows: hard_ids rows shorter than n_hard need hard_groups (repeating slots)")
        if groups is not None:
            hmx.check(hmx.lib().hmx_nce_fwd_periodic_f32(hmx.ptr(q), hmx.ptr(k), m, d, hmx.ptr(tab), n_rows, hmx.ptr(hid),
                                                         n_hard, groups[1], hmx.ptr(iid), n_inb, hard_ld, flags,
                                                         float(temperature), int(bool(normalize)), hmx.ptr(logits),
                                                         hmx.ptr(row_loss), hmx.ptr(lse), hmx.ptr(q_inv), hmx.ptr(k_inv),
                                                         hmx.ptr(dq_unit), hmx.current_stream()), "hmx_nce_fwd_periodic_f32")
        else:
            hmx.check(hmx.lib().hmx_nce_fwd_f32(hmx.ptr(q), hmx.ptr(k), m, d, hmx.ptr(tab), n_rows, hmx.ptr(hid), n_hard,
                                                hmx.ptr(iid), n_inb, hard_ld, flags, float(temperature),
                                                int(bool(normalize)), hmx.ptr(logits), hmx.ptr(row_loss), hmx.ptr(lse),
                                                hmx.ptr(q_inv), hmx.ptr(k_inv), hmx.ptr(dq_unit), hmx.current_stream()),
                      "hmx_nce_fwd_f32")
        ctx.save_for_backward(q, k, tab, hid, iid, logits, lse, q_inv, k_inv, dq_unit)
        ctx.cfg = (float(temperature), int(bool(normalize)), n_rows, n_hard, n_inb, hard_ld, flags)
        ctx.table_param = table
        ctx.groups = groups
        return row_loss

    @staticmethod
    def backward(ctx, grad_rows):
        q, k, tab, hid, iid, logits, lse, q_inv, k_inv, dq_unit = ctx.saved_tensors
        temperature, normalize, n_rows, n_hard, n_inb, hard_ld, flags = ctx.cfg
        m, d = q.shape
        dev = q.device
        g = _f32c(grad_rows)
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        want_table = n_hard > 0 and ctx.needs_input_grad[2]
        # The kernel needs a destination that is zero on entry (it normalises the touched rows in place).  With
        # fused table gradients on and no gradient accumulated yet, that destination is the table's own .grad,
        # created here (this backward runs before the gathers' backward, the loss being the last op of forward).
        fused_dst = None
        if want_table and _FUSE_TABLE_GRAD and isinstance(ctx.table_param, torch.nn.Parameter) \
                and ctx.table_param.is_leaf and ctx.table_param.grad is None:
            fused_dst = _into_param_grad(ctx.table_param)
        dtable = fused_dst
        if dtable is None and n_hard > 0:
            dtable = torch.zeros(n_rows, d, dtype=torch.float32, device=dev)
        L = hmx.lib()
        coef = logits   # overwritten with the key-gradient coefficients: backward may run once
        if ctx.groups is not None:
            gs, period = ctx.groups
            ng = gs.numel() - 1
            ws = workspace(L.hmx_nce_bwd_grouped_workspace_bytes(m, d, n_rows, n_hard, n_inb, ng, period), dev)
            hmx.check(L.hmx_nce_bwd_grouped_f32(hmx.ptr(q), hmx.ptr(k), m, d, hmx.ptr(tab), n_rows, hmx.ptr(hid), n_hard,
                                                hmx.ptr(iid), n_inb, hard_ld, flags, temperature, normalize, hmx.ptr(coef),
                                                hmx.ptr(lse), hmx.ptr(q_inv), hmx.ptr(k_inv), hmx.ptr(g), hmx.ptr(dq_unit),
                                                hmx.ptr(dq), hmx.ptr(dk), hmx.ptr(dtable), hmx.ptr(_row_flags_of(dtable)),
                                                hmx.ptr(gs), ng, period, hmx.ptr(ws), ws.numel(), hmx.current_stream()),
                      "hmx_nce_bwd_grouped_f32")
        else:
            ws = workspace(L.hmx_nce_bwd_workspace_bytes(m, d, n_rows, n_hard, n_inb), dev)
            hmx.check(L.hmx_nce_bwd_f32(hmx.ptr(q), hmx.ptr(k), m, d, hmx.ptr(tab), n_rows, hmx.ptr(hid), n_hard,
                                        hmx.ptr(iid), n_inb, hard_ld, flags, temperature, normalize, hmx.ptr(coef),
                                        hmx.ptr(lse), hmx.ptr(q_inv), hmx.ptr(k_inv), hmx.ptr(g), hmx.ptr(dq_unit),
                                        hmx.ptr(dq), hmx.ptr(dk), hmx.ptr(dtable), hmx.ptr(_row_flags_of(dtable)),
                                        hmx.ptr(ws), ws.numel(), hmx.current_stream()), "hmx_nce_bwd_f32")
        return (dq, dk, (dtable if (want_table and fused_dst is None) else None), None, None, None, None, None, None,
                None)


def fused_nce_rows(q, k, table, hard_ids, inb_idx, temperature=0.05, normalize=True, hard_groups=None, n_hard=None,
                   inb_skip_self=False):
    """Per-row InfoNCE loss over [positive k_i | table[hard_ids[i]] | k[inb_idx[i]]] (hydra3.py:845-974 with the
    sampling done by the caller).  q, k: (M, d); hard_ids: (M, n_hard) item-table rows or None; inb_idx:
    (M, n_inb) rows of k (int64 or int32) or None.  Gradients flow to q, k and table.  backward() may run once.
    hard_groups = (group_start (G + 1,) int64 offsets into the M rows, period): the caller's promise that inside each
    group of consecutive rows hard_ids[i, s] == hard_ids[first row of the group, s % period] (the reference's tiling
    of each user's mined ids); the table gradient is then built per group instead of per (row, slot) entry.  With
    groups, hard_ids may hold just the first `period` (or more) columns and n_hard says how many slots they stand
    for, so the tiled copy is never built.
    inb_skip_self: inb_idx holds int32 draws r in [0, M - 1) and row i's key is row r + (r >= i) (hydra3.py:824-829)."""
    return _FusedNCE.apply(q, k, table, hard_ids, inb_idx, temperature, normalize, hard_groups, n_hard, inb_skip_self)


# --------------------------------------------------------------------------- scoring + top-K
def _dpad(d):
    p = 16
    while p < d:
        p *= 2
    return p


_PREPARED_BY_PTR = {}     # data_ptr of a table -> weakref of its PreparedTable (the optimizer keeps the copy current)


def prepared_copy_of(param):
    """The live PreparedTable of this parameter's storage, if one exists (optim.HmxAdamW writes it during its step)."""
    ref = _PREPARED_BY_PTR.get(param.data_ptr())
    pt = ref() if ref is not None else None
    if pt is None or pt.table.data_ptr() != param.data_ptr() or pt.table.shape != param.shape:
        return None
    return pt


class PreparedTable:
    """bf16 copy (row stride = next power of two >= max(16, d)) and largest row norm of an fp32
    table shard, as the bf16 coarse scoring pass needs them.  Re-run refresh() after the table's
    weights change."""

    def __init__(self, table):
        _need_cuda(table)
        self.table = _f32c(table)
        n, d = self.table.shape
        if d > 128:
            raise hmx.HmxError("bf16 coarse scoring supports d <= 128")
        self.bf16 = torch.empty(n, _dpad(d), dtype=torch.int16, device=table.device)
        self.max_row_norm = torch.zeros(1, dtype=torch.float32, device=table.device)
        self.synced_version = None      # table._version at which an optimizer step last rewrote this copy itself
        self.refresh()
        _PREPARED_BY_PTR[self.table.data_ptr()] = __import__("weakref").ref(self)

    def refresh(self, table=None):
        if table is not None:
            self.table = _f32c(table)
        n, d = self.table.shape
        hmx.check(hmx.lib().hmx_table_prepare_bf16(hmx.ptr(self.table), n, d, hmx.ptr(self.bf16),
                                                   hmx.ptr(self.max_row_norm), hmx.current_stream()),
                  "hmx_table_prepare_bf16")
        return self


def score_topk_exact(queries, table, k, exclude=None, id_offset=0):
    """Exact top-k by inner product (fp32 MFMA, fma-chain order; score desc then id asc)."""
    _need_cuda(queries, table, exclude)
    q, t = _f32c(queries), _f32c(table)
    nq, d = q.shape
    ex = _i64c(exclude) if exclude is not None else None
    ids = torch.empty(nq, k, dtype=torch.int64, device=q.device)
    sc = torch.empty(nq, k, dtype=torch.float32, device=q.device)
    L = hmx.lib()
    ws = workspace(L.hmx_score_topk_workspace_bytes(nq, t.shape[0], d, k), q.device)
    hmx.check(L.hmx_score_topk_f32(hmx.ptr(q), nq, hmx.ptr(t), t.shape[0], d, k, hmx.ptr(ex),
                                   0 if ex is None else ex.shape[1], id_offset, hmx.ptr(ids), hmx.ptr(sc),
                                   hmx.ptr(ws), ws.numel(), hmx.current_stream()), "hmx_score_topk_f32")
    return ids, sc


def score_topk(queries, prepared, k, exclude=None, id_offset=0, return_stats=False):
    """Top-k by inner product: bf16 MFMA coarse pass + exact re-score, with the exact fp32 kernel
    re-run for the (rare) queries whose shortlist could not be proven complete.  Same result as
    score_topk_exact."""
    _need_cuda(queries, exclude)
    q = _f32c(queries)
    t = prepared.table
    nq, d = q.shape
    ex = _i64c(exclude) if exclude is not None else None
    ids = torch.empty(nq, k, dtype=torch.int64, device=q.device)
    sc = torch.empty(nq, k, dtype=torch.float32, device=q.device)
    unc = torch.empty(nq, dtype=torch.int32, device=q.device)
    L = hmx.lib()
    ws = workspace(L.hmx_score_topk_workspace_bytes(nq, t.shape[0], d, k), q.device)
    hmx.check(L.hmx_score_topk_bf16(hmx.ptr(q), nq, hmx.ptr(t), hmx.ptr(prepared.bf16), hmx.ptr(prepared.max_row_norm),
                                    t.shape[0], d, k, hmx.ptr(ex), 0 if ex is None else ex.shape[1], id_offset,
                                    hmx.ptr(ids), hmx.ptr(sc), hmx.ptr(unc), hmx.ptr(ws), ws.numel(),
                                    hmx.current_stream()), "hmx_score_topk_bf16")
    if return_stats:                          # the count is wanted on the host: one synchronisation
        redo = unc.nonzero().flatten()
        if redo.numel():
            ids2, sc2 = score_topk_exact(q[redo], t, k, None if ex is None else ex[redo], id_offset)
            ids[redo], sc[redo] = ids2, sc2
        return ids, sc, {"uncertain": int(redo.numel())}
    # no host synchronisation: the exact kernels run over the flagged queries only (tiles without one return at once)
    hmx.check(L.hmx_score_topk_f32_flagged(hmx.ptr(q), nq, hmx.ptr(t), t.shape[0], d, k, hmx.ptr(ex),
                                           0 if ex is None else ex.shape[1], id_offset, hmx.ptr(unc), hmx.ptr(ids),
                                           hmx.ptr(sc), hmx.ptr(ws), ws.numel(), hmx.current_stream()),
              "hmx_score_topk_f32_flagged")
    return ids, sc


def touched_rows(x):
    """Ascending indices (int64) of the rows of x (n_rows, d) that hold a non-zero value: one pass over x on the HIP
    kernel, then a compaction of the per-row flags (host sync: the count)."""
    _need_cuda(x)
    x = _f32c(x)
    st = _FLAGS_BY_PTR.get(x.data_ptr())
    if st is not None and st.buf.shape == x.shape and not st.dense:
        return st.flags.nonzero().flatten()        # a table gradient with row flags: the flagged rows, no pass over x
    flags = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
    hmx.check(hmx.lib().hmx_row_any_nonzero_f32(hmx.ptr(x), x.shape[0], x.shape[1], hmx.ptr(flags),
                                                hmx.current_stream()), "hmx_row_any_nonzero_f32")
    return flags.nonzero().flatten()


def touched_flags(x):
    """One byte per row of x (n_rows, d), non-zero where the row holds (or, for a row-flagged table gradient, was
    written with) a value: touched_rows without the compaction, hence without a host sync."""
    _need_cuda(x)
    x = _f32c(x)
    st = _FLAGS_BY_PTR.get(x.data_ptr())
    if st is not None and st.buf.shape == x.shape and not st.dense:
        return st.flags
    flags = torch.empty(x.shape[0], dtype=torch.uint8, device=x.device)
    hmx.check(hmx.lib().hmx_row_any_nonzero_f32(hmx.ptr(x), x.shape[0], x.shape[1], hmx.ptr(flags),
                                                hmx.current_stream()), "hmx_row_any_nonzero_f32")
    return flags


@torch.no_grad()
def gather_rows_padded(weight, ids):
    """weight[ids] with zero rows where ids is negative or past the table (padded id lists); no gradient."""
    _need_cuda(weight, ids)
    w = _f32c(weight.detach())
    idc = _i64c(ids)
    out = torch.empty(*idc.shape, w.shape[1], dtype=torch.float32, device=w.device)
    hmx.check(hmx.lib().hmx_gather_rows_f32(hmx.ptr(w), w.shape[0], w.shape[1], hmx.ptr(idc), idc.numel(),
                                            hmx.ptr(out), hmx.current_stream()), "hmx_gather_rows_f32")
    return out


def sample_thresholds(queries, prepared, k, n_ranges=1, n_values=1):
    """(n_q,) start thresholds the bf16 pass samples from this table (-inf for launch shapes that do not sample):
    at least k sampled rows of the table score that much.  Exchanged between item-range shards (retrieval.py);
    n_ranges = how many ranges' thresholds will be combined (each samples its share).  n_values = 2 m returns
    (n_q, 2 m): the m largest sampled scores of each half of the sampled rows, for a k-th-of-the-union combine."""
    _need_cuda(queries)
    q = _f32c(queries)
    nq, d = q.shape
    t = prepared.table
    thr = torch.empty((nq,) if n_values == 1 else (nq, n_values), dtype=torch.float32, device=q.device)
    L = hmx.lib()
    ws = workspace(L.hmx_score_topk_workspace_bytes(nq, t.shape[0], d, k), q.device)
    hmx.check(L.hmx_topk_sample_thresholds_bf16(hmx.ptr(q), nq, hmx.ptr(prepared.bf16), t.shape[0], d, k, n_ranges,
                                                n_values, hmx.ptr(thr), hmx.ptr(ws), ws.numel(), hmx.current_stream()),
              "hmx_topk_sample_thresholds_bf16")
    return thr


def kth_largest(vals, k):
    """vals (S, n_q, v) -> (n_q,): the k-th largest of the S * v numbers of each query (-inf when fewer than k)."""
    _need_cuda(vals)
    v = _f32c(vals)
    S, nq, nv = v.shape
    out = torch.empty(nq, dtype=torch.float32, device=v.device)
    hmx.check(hmx.lib().hmx_kth_largest_f32(hmx.ptr(v), S, nq * nv, nq, nv, k, hmx.ptr(out), hmx.current_stream()),
              "hmx_kth_largest_f32")
    return out


def score_topk_seeded(queries, prepared, k, start_thresholds, id_offset=0):
    """The bf16 pass from given start thresholds, completeness left to the caller: -> (ids, scores, bounds, failed).
    A list may hold fewer than k rows (ids -1, scores -inf); bounds[q] is the score below which rows of this table
    may have been dropped; failed[q] != 0 marks queries of a launch that gave up.  No host synchronisation."""
    _need_cuda(queries, start_thresholds)
    q, thr = _f32c(queries), _f32c(start_thresholds)
    nq, d = q.shape
    assert thr.shape == (nq,)
    t = prepared.table
    ids = torch.empty(nq, k, dtype=torch.int64, device=q.device)
    sc = torch.empty(nq, k, dtype=torch.float32, device=q.device)
    bounds = torch.empty(nq, dtype=torch.float32, device=q.device)
    failed = torch.empty(nq, dtype=torch.int32, device=q.device)
    L = hmx.lib()
    ws = workspace(L.hmx_score_topk_workspace_bytes(nq, t.shape[0], d, k), q.device)
    hmx.check(L.hmx_score_topk_bf16_seeded(hmx.ptr(q), nq, hmx.ptr(t), hmx.ptr(prepared.bf16),
                                           hmx.ptr(prepared.max_row_norm), t.shape[0], d, k, id_offset, hmx.ptr(thr),
                                           hmx.ptr(ids), hmx.ptr(sc), hmx.ptr(bounds), hmx.ptr(failed), hmx.ptr(ws),
                                           ws.numel(), hmx.current_stream()), "hmx_score_topk_bf16_seeded")
    return ids, sc, bounds, failed


def topk_merge(cand_ids, cand_scores):
    """(S, n_q, k) per-shard lists -> global (n_q, k) with the same order / tie-break."""
    _need_cuda(cand_ids, cand_scores)
    ci, cs = _i64c(cand_ids), _f32c(cand_scores)
    S, nq, k = ci.shape
    ids = torch.empty(nq, k, dtype=torch.int64, device=ci.device)
    sc = torch.empty(nq, k, dtype=torch.float32, device=ci.device)
    hmx.check(hmx.lib().hmx_topk_merge(hmx.ptr(ci), hmx.ptr(cs), S, nq, k, hmx.ptr(ids), hmx.ptr(sc),
                                       hmx.current_stream()), "hmx_topk_merge")
    return ids, sc
