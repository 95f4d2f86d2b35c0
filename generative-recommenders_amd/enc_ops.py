"""torch.autograd wrappers for the encoder kernels of libhmx.so.

All tensors are CUDA fp32; token-major (T, features).  No CPU fallback.
"""
import ctypes
import math

import torch

from . import hmx
from .ops import _f32c, _i64c, _need_cuda, workspace

N_STATE = 16
DT_RANK = 16
P_DBC = DT_RANK + 2 * N_STATE


def _s():
    return hmx.current_stream()


def _L():
    return hmx.lib()


def _empty(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def colsum(P):
    """(rows, cols) -> (cols): deterministic column sums."""
    P = _f32c(P)
    rows, cols = P.shape
    out = _empty(cols, like=P)
    nbytes = _L().hmx_colsum_workspace_bytes(rows, cols)
    ws = workspace(nbytes, P.device) if nbytes else None
    hmx.check(_L().hmx_colsum_f32(hmx.ptr(P), rows, cols, cols, hmx.ptr(out), hmx.ptr(ws), nbytes, _s()),
              "hmx_colsum_f32")
    return out


def combine(xs, coefs, rowmask=None):
    """(sum_i coefs[i] * xs[i]) * rowmask[:, None] for up to four (T, d) tensors."""
    xs = [_f32c(x) for x in xs]
    T, d = xs[0].shape
    out = _empty(T, d, like=xs[0])
    p = [hmx.ptr(x) for x in xs] + [None] * (4 - len(xs))
    c = list(coefs) + [0.0] * (4 - len(xs))
    hmx.check(_L().hmx_combine_f32(p[0], p[1], p[2], p[3], c[0], c[1], c[2], c[3], hmx.ptr(rowmask), T, d,
                                   hmx.ptr(out), _s()), "hmx_combine_f32")
    return out


class _Combine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rowmask, c0, c1, c2, c3, *xs):
        ctx.coefs = (c0, c1, c2, c3)[:len(xs)]
        ctx.save_for_backward(rowmask) if rowmask is not None else None
        ctx.has_mask = rowmask is not None
        return combine(xs, ctx.coefs, rowmask)

    @staticmethod
    def backward(ctx, g):
        rm = ctx.saved_tensors[0] if ctx.has_mask else None
        g = _f32c(g)
        grads = []
        for i, c in enumerate(ctx.coefs):
            if not ctx.needs_input_grad[5 + i]:
                grads.append(None)
            elif rm is None and c == 1.0:
                grads.append(g)
            else:
                grads.append(combine([g], [c], rm))
        return (None, None, None, None, None, *grads)


def combine_ad(xs, coefs, rowmask=None):
    c = list(coefs) + [0.0] * (4 - len(xs))
    return _Combine.apply(rowmask, c[0], c[1], c[2], c[3], *xs)


# ------------------------------------------------------------------------------- RMSNorm
class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, eps):
        _need_cuda(x, scale)
        x, scale = _f32c(x), _f32c(scale)
        T, d = x.shape
        y, rstd = _empty(T, d, like=x), _empty(T, like=x)
        hmx.check(_L().hmx_rmsnorm_fwd(hmx.ptr(x), hmx.ptr(scale), T, d, eps, hmx.ptr(y), hmx.ptr(rstd), _s()),
                  "hmx_rmsnorm_fwd")
        ctx.save_for_backward(x, scale, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, rstd = ctx.saved_tensors
        dy = _f32c(dy)
        T, d = x.shape
        nb = _L().hmx_rmsnorm_bwd_blocks(T)
        dx, part = _empty(T, d, like=x), _empty(nb, d, like=x)
        hmx.check(_L().hmx_rmsnorm_bwd(hmx.ptr(x), hmx.ptr(scale), hmx.ptr(rstd), hmx.ptr(dy), T, d, hmx.ptr(dx),
                                       hmx.ptr(part), _s()), "hmx_rmsnorm_bwd")
        return dx, colsum(part), None


def rmsnorm(x, scale, eps=1e-6):
    return _RMSNorm.apply(x, scale, eps)


# ------------------------------------------------------------------------------- deferred weight gradients
# Opt-in (set_direct_param_grad): the weight / bias gradients of the projections are not returned to autograd one
# launch at a time; their operands are queued and up to 24 products run as ONE launch that adds straight into
# `param.grad`.  Same sums (fp32 MFMA, fixed order per launch shape); a backward pass of the encoder then has a dozen
# weight-gradient launches instead of some 180.  Like ops.set_fused_table_grad it bypasses autograd's accumulation,
# so torch.autograd.grad() does not see these gradients; leave it off there.
_DIRECT_PARAM_GRAD = False
_WGRAD_BATCH = 24
_WGRAD_TILES = 1024      # output tiles one batched launch can hold


def set_direct_param_grad(flag=True):
    global _DIRECT_PARAM_GRAD
    _wgrad_queue.flush()
    _DIRECT_PARAM_GRAD = bool(flag)


class GradArena:
    """One flat zero-filled buffer per backward pass for the gradients of a module's dense parameters: `.grad`
    tensors are views into it, created when a parameter first receives a gradient (parameters that receive none
    keep grad = None, as under plain autograd).  One fill instead of one per parameter."""

    def __init__(self, params):
        import weakref
        self.offsets, self.total = {}, 0
        for q in params:
            self.offsets[id(q)] = (self.total, q.numel(), weakref.ref(q))   # id() alone could be a recycled one
            self.total += (q.numel() + 63) // 64 * 64
        self.buf = None
        self.views = []
        self.handed = set()      # slots handed out from the current buffer

    def alive(self):
        return any(slot[2]() is not None for slot in self.offsets.values())

    def target(self, param):
        slot = self.offsets.get(id(param))
        if slot is None or slot[2]() is not param:
            return None
        # a new buffer once every view of the old one has been dropped (zero_grad(set_to_none=True))
        if self.buf is None or all(v() is None for v in self.views) or self.buf.device != param.device:
            self.buf = torch.zeros(self.total, dtype=torch.float32, device=param.device)
            self.views = []
            self.handed = set()
        g = self.buf[slot[0]:slot[0] + slot[1]].view(param.shape)
        if slot[0] in self.handed:
            # this parameter's slice was in use before (its .grad was dropped while another view of the buffer is
            # still held, e.g. a partial zero_grad or a logger keeping a gradient): it holds the old gradient
            g.zero_()
        self.handed.add(slot[0])
        import weakref
        self.views.append(weakref.ref(g))
        return g


_ARENAS = []


def register_grad_arena(module, exclude=()):
    """Dense (non-table) parameters of `module` get their .grad from one flat buffer in direct-gradient mode."""
    skip = {id(q) for q in exclude}
    arena = GradArena([q for q in module.parameters() if id(q) not in skip])
    _ARENAS[:] = [a for a in _ARENAS if a.alive()]
    _ARENAS.append(arena)
    return arena


def _param_grad_target(param):
    """The tensor a direct weight gradient is added into (created zero-filled if the parameter has none yet), or None
    when the standard autograd path applies."""
    if not (_DIRECT_PARAM_GRAD and isinstance(param, torch.nn.Parameter) and param.is_leaf and param.requires_grad
            and param.dtype == torch.float32 and param.is_contiguous()):
        return None
    if param.grad is None:
        g = None
        for arena in _ARENAS:
            g = arena.target(param)
            if g is not None:
                break
        param.grad = g if g is not None else torch.zeros_like(param)
    g = param.grad
    return g if (g.is_contiguous() and g.dtype == torch.float32 and g.device == param.device) else None


class _WgradQueue:
    def __init__(self):
        self.jobs, self.keep, self.targets, self.tiles = [], [], set(), 0

    def add(self, dy, dy_off, ldy, x, x_off, ldx, in_mask, T, N, K, dw, dw_off, ldw, db, db_off, groups=(0, 0, 0, 0)):
        """dW[dw_off ...] += dY^T X (and dbias += column sums of dY); offsets are in elements.  groups: (n_group,
        n_group_padded, k_group, k_group_padded) of hmx_wgrad_desc -- widened operands against a true weight."""
        tw = dw.data_ptr() + 4 * dw_off
        tb = db.data_ptr() + 4 * db_off if db is not None else None
        tiles = -(-N // 32) * -(-(K + (tb is not None)) // 32)       # 32 x 32 output tiles, bias as one more column
        if (tw in self.targets or (tb is not None and tb in self.targets) or len(self.jobs) >= _WGRAD_BATCH
                or self.tiles + tiles > _WGRAD_TILES):
            self.flush()
        self.tiles += tiles
        d = hmx.WgradDesc(dy.data_ptr() + 4 * dy_off, ldy, x.data_ptr() + 4 * x_off, ldx,
                          in_mask.data_ptr() if in_mask is not None else None, T, N, K, tw, ldw, tb, 1, *groups)
        self.jobs.append(d)
        self.keep.append((dy, x, in_mask, dw, db))
        self.targets.add(tw)
        if tb is not None:
            self.targets.add(tb)
        self.stream = _s()
        self.device = dy.device
        if len(self.jobs) == 1:
            # first job since the queue was last emptied: ask for a flush when the backward pass that is executing now
            # has finished.  (Registered per filling, not once per pass: a pass that raises never runs its callbacks,
            # and a flag that only a callback clears would then stay set for good.)
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
            except RuntimeError:
                self.flush()          # not inside a backward pass: nothing to wait for

    def discard(self):
        """Drops queued products without running them: what a backward pass that raised left behind."""
        self.jobs, self.keep, self.targets, self.tiles = [], [], set(), 0

    def flush(self):
        if not self.jobs:
            return
        n = len(self.jobs)
        arr = (hmx.WgradDesc * n)(*self.jobs)
        nbytes = _L().hmx_linear_wgrad_batch_workspace_bytes(arr, n)
        ws = workspace(nbytes, self.device)
        rc = _L().hmx_linear_wgrad_batch_f32(arr, n, hmx.ptr(ws), ws.numel(), self.stream)
        self.jobs, self.keep, self.targets, self.tiles = [], [], set(), 0
        hmx.check(rc, "hmx_linear_wgrad_batch_f32")


_wgrad_queue = _WgradQueue()


def flush_deferred_wgrads():
    """Run the queued weight-gradient products now (a backward pass does this by itself when it ends)."""
    _wgrad_queue.flush()


def discard_stale_wgrads():
    """Called where a new training forward pass starts: a backward pass that completed has flushed its queue, so
    anything still queued belongs to a pass that raised part-way; its products must not be added to the gradients of
    the step that starts now."""
    _wgrad_queue.discard()


# ------------------------------------------------------------------------------- linear
def _linear_raw(x_ptr, ldx, w_ptr, w_rs, w_cs, bias, T, N, K, y_ptr, ldy, r1=None, c1=0.0, r2=None, c2=0.0,
                in_mask=None, out_mask=None, accumulate=False):
    hmx.check(_L().hmx_linear_f32(x_ptr, ldx, w_ptr, w_rs, w_cs, hmx.ptr(bias), T, N, K, y_ptr, ldy,
                                  hmx.ptr(r1), N if r1 is not None else 0, c1, hmx.ptr(r2), N if r2 is not None else 0,
                                  c2, hmx.ptr(in_mask), hmx.ptr(out_mask), 1 if accumulate else 0, _s()),
              "hmx_linear_f32")


def _heads_linear_raw(x_ptr, ldx, x_hs, ws, w_rs, w_cs, T, N, K, y_ptr, ldy, y_hs, accumulate=False):
    """All heads of a block-diagonal projection in one launch (hmx_heads_linear_f32); ws: one weight tensor per head."""
    for h0 in range(0, len(ws), 8):               # the launch takes up to eight heads
        part = ws[h0:h0 + 8]
        ptrs = (ctypes.c_void_p * len(part))(*[w.data_ptr() for w in part])
        hmx.check(_L().hmx_heads_linear_f32(x_ptr + 4 * h0 * x_hs, ldx, x_hs, ptrs, w_rs, w_cs, T, N, K,
                                            y_ptr + 4 * h0 * y_hs, ldy, y_hs, len(part), 1 if accumulate else 0, _s()),
                  "hmx_heads_linear_f32")


def _wgrad_raw(dy_ptr, ldy, x_ptr, ldx, in_mask, T, N, K, dw_ptr, ldw, db, device):
    ws = workspace(_L().hmx_linear_wgrad_workspace_bytes(T, N, K), device)
    hmx.check(_L().hmx_linear_wgrad_f32(dy_ptr, ldy, x_ptr, ldx, hmx.ptr(in_mask), T, N, K, dw_ptr, ldw, hmx.ptr(db),
                                        hmx.ptr(ws), ws.numel(), _s()), "hmx_linear_wgrad_f32")


class _Linear(torch.autograd.Function):
    """y = ((x * in_mask) W[:, col_off:col_off + K]^T + b + c1 r1 + c2 r2) * out_mask   (at most one of the masks)."""

    @staticmethod
    def forward(ctx, x, weight, bias, in_mask, out_mask, r1, c1, r2, c2, col_off=0, rows=None):
        _need_cuda(x, weight)
        assert in_mask is None or out_mask is None
        x, w_full = _f32c(x), _f32c(weight)
        row0, row1 = rows if rows is not None else (0, w_full.shape[0])
        w = w_full[row0:row1]                       # the output rows used (a view: rows of a row-major matrix)
        b_used = bias[row0:row1] if (bias is not None and rows is not None) else bias
        T, K = x.shape
        N, ldw = w.shape
        assert ldw >= col_off + K
        r1 = _f32c(r1) if r1 is not None else None
        r2 = _f32c(r2) if r2 is not None else None
        y = _empty(T, N, like=x)
        _linear_raw(hmx.ptr(x), K, hmx.ptr(w) + 4 * col_off, ldw, 1, b_used, T, N, K, hmx.ptr(y), N, r1, c1, r2, c2,
                    in_mask, out_mask)
        ctx.save_for_backward(x, w, in_mask, out_mask)
        ctx.c = (c1, c2)
        ctx.col_off = col_off
        ctx.rows = (row0, row1, w_full.shape[0])
        ctx.has_bias = bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, in_mask, out_mask = ctx.saved_tensors
        T, K = x.shape
        N, ldw = w.shape
        row0, row1, n_all = ctx.rows
        g = _f32c(dy)
        if out_mask is not None:
            g = combine([g], [1.0], out_mask)
        need = ctx.needs_input_grad
        dx = dw = db = dr1 = dr2 = None
        if need[0]:
            dx = _empty(T, K, like=x)
            _linear_raw(hmx.ptr(g), N, hmx.ptr(w) + 4 * ctx.col_off, 1, ldw, None, T, K, N, hmx.ptr(dx), K,
                        out_mask=in_mask)
        want_db = ctx.has_bias and need[2]
        tw = _param_grad_target(ctx.wparam) if need[1] else None
        tb = _param_grad_target(ctx.bparam) if want_db else None
        if tw is not None and (tb is not None or not want_db):
            # direct mode: queued, added into .grad by a batched launch; nothing is returned to autograd
            _wgrad_queue.add(g, 0, N, x, 0, K, in_mask, T, N, K, tw, row0 * ldw + ctx.col_off, ldw, tb, row0)
            want_db = False
        elif need[1]:
            whole = (row0, row1) == (0, n_all)
            dw = torch.zeros(n_all, ldw, dtype=w.dtype, device=w.device) if (ldw > K or not whole) else torch.empty_like(w)
            db_full = (torch.zeros(n_all, dtype=w.dtype, device=w.device) if not whole else _empty(N, like=x)) if want_db else None
            _wgrad_raw(hmx.ptr(g), N, hmx.ptr(x), K, in_mask, T, N, K, hmx.ptr(dw) + 4 * (row0 * ldw + ctx.col_off), ldw,
                       db_full[row0:row1] if db_full is not None else None, x.device)
            db = db_full
            want_db = False
        if want_db:              # frozen weight, trainable bias: the bias gradient is the column sum of dy
            db = colsum(g)
            if (row0, row1) != (0, n_all):
                full = torch.zeros(n_all, dtype=db.dtype, device=db.device)
                full[row0:row1] = db
                db = full
        if need[5]:
            dr1 = g if ctx.c[0] == 1.0 else combine([g], [ctx.c[0]])
        if need[7]:
            dr2 = g if ctx.c[1] == 1.0 else combine([g], [ctx.c[1]])
        return dx, dw, db, None, None, dr1, None, dr2, None, None, None


def linear(x, weight, bias=None, in_mask=None, out_mask=None, r1=None, c1=0.0, r2=None, c2=0.0, col_off=0, rows=None):
    """nn.Linear forward on bf16 MFMA; uses the x.shape[1] columns of `weight` that start at column `col_off`
    (a column block of a wider projection: the other columns see no input and get a zero gradient).  rows = (r0, r1):
    only output rows r0 .. r1 of `weight` / `bias` are used (the q or the k / v block of an attention in_proj): the
    parameters are passed whole, so their gradients go straight into the full .grad instead of through slice views."""
    return _Linear.apply(x, weight, bias, in_mask, out_mask, r1, float(c1), r2, float(c2), int(col_off),
                         None if rows is None else (int(rows[0]), int(rows[1])))


class _HeadsLinear(torch.autograd.Function):
    """Block-diagonal projection: y[:, h*n_out:(h+1)*n_out] = x[:, h*k_in:(h+1)*k_in] W_h^T for every head h, with
    W_h = params[h][r0:r1] (rows r0..r1 of that head's parameter)."""

    @staticmethod
    def forward(ctx, x, r0, r1, *params):
        _need_cuda(x)
        x = _f32c(x)
        ws = [_f32c(q)[r0:r1] for q in params]
        H = len(ws)
        n_out, k_in = ws[0].shape
        T = x.shape[0]
        assert x.shape[1] == H * k_in
        y = _empty(T, H * n_out, like=x)
        _heads_linear_raw(x.data_ptr(), H * k_in, k_in, ws, k_in, 1, T, n_out, k_in, y.data_ptr(), H * n_out, n_out)
        ctx.save_for_backward(x, *ws)
        ctx.params, ctx.rows = params, (r0, r1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, *ws = ctx.saved_tensors
        g = _f32c(dy)
        H = len(ws)
        n_out, k_in = ws[0].shape
        T = x.shape[0]
        r0, r1 = ctx.rows
        dx = _empty(T, H * k_in, like=x) if ctx.needs_input_grad[0] else None
        dws = []
        if dx is not None:
            _heads_linear_raw(g.data_ptr(), H * n_out, n_out, ws, 1, k_in, T, k_in, n_out, dx.data_ptr(), H * k_in, k_in)
        for h, w in enumerate(ws):
            gp = g.data_ptr() + 4 * h * n_out
            if not ctx.needs_input_grad[3 + h]:
                dws.append(None)
                continue
            tw = _param_grad_target(ctx.params[h])
            if tw is not None:
                _wgrad_queue.add(g, h * n_out, H * n_out, x, h * k_in, H * k_in, None, T, n_out, k_in, tw, r0 * k_in,
                                 k_in, None, 0)
                dws.append(None)
                continue
            full = ctx.params[h]
            dw = torch.zeros_like(full) if (r0, r1) != (0, full.shape[0]) else torch.empty_like(full)
            _wgrad_raw(gp, H * n_out, x.data_ptr() + 4 * h * k_in, H * k_in, None, T, n_out, k_in,
                       dw.data_ptr() + 4 * r0 * k_in, k_in, None, x.device)
            dws.append(dw)
        return (dx, None, None, *dws)


def heads_linear(x, weights, rows=None):
    """weights: one (n, k) parameter per head; rows = (r0, r1) restricts every head to those rows of its parameter
    (the x / z halves of in_proj) without slicing the parameter under autograd."""
    r0, r1 = rows if rows is not None else (0, weights[0].shape[0])
    return _HeadsLinear.apply(x, r0, r1, *weights)


# ------------------------------------------------------------------------------- pointwise
_ROPE_CACHE = {}


def rope_tables(L, hd, device, pairs=None):
    """cos/sin of t * 10000^(-2i/hd), t = 0..L-1 (hydra3.py:318-328), computed once on the CPU in fp32.
    pairs > hd / 2: rows widened to `pairs` entries with the identity rotation (heads in a widened feature layout)."""
    pairs = hd // 2 if pairs is None else pairs
    key = (L, hd, pairs, str(device))
    if key not in _ROPE_CACHE:
        theta = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        ang = torch.arange(L, dtype=torch.float32)[:, None] * theta[None, :]
        ang = torch.nn.functional.pad(ang, (0, pairs - hd // 2))
        _ROPE_CACHE[key] = (torch.cos(ang).to(device).contiguous(), torch.sin(ang).to(device).contiguous())
    return _ROPE_CACHE[key]


class _SiluRope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, L, hd):
        _need_cuda(x1)
        x1 = _f32c(x1)
        T, d = x1.shape
        cs, sn = rope_tables(L, hd, x1.device)
        s, y2 = _empty(T, d, like=x1), _empty(T, d, like=x1)
        hmx.check(_L().hmx_silu_rope_fwd(hmx.ptr(x1), hmx.ptr(cs), hmx.ptr(sn), T, L, d, hd, hmx.ptr(s), hmx.ptr(y2),
                                         _s()), "hmx_silu_rope_fwd")
        ctx.save_for_backward(x1)
        ctx.dims = (L, hd)
        return s, y2

    @staticmethod
    def backward(ctx, ds, dy2):
        (x1,) = ctx.saved_tensors
        L, hd = ctx.dims
        T, d = x1.shape
        cs, sn = rope_tables(L, hd, x1.device)
        ds = _f32c(ds) if ds is not None else torch.zeros_like(x1)
        dy2 = _f32c(dy2) if dy2 is not None else torch.zeros_like(x1)
        dx1 = torch.empty_like(x1)
        hmx.check(_L().hmx_silu_rope_bwd(hmx.ptr(x1), hmx.ptr(cs), hmx.ptr(sn), hmx.ptr(ds), hmx.ptr(dy2), T, L, d, hd,
                                         hmx.ptr(dx1), _s()), "hmx_silu_rope_bwd")
        return dx1, None, None


def silu_rope(x1, L, hd):
    return _SiluRope.apply(x1, L, hd)


class _Binary(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c, mode):
        _need_cuda(a, b)
        a, b = _f32c(a), _f32c(b)
        out = torch.empty_like(a)
        hmx.check(_L().hmx_binary_fwd(hmx.ptr(a), hmx.ptr(b), a.numel(), c, mode, hmx.ptr(out), _s()), "hmx_binary_fwd")
        ctx.save_for_backward(a, b)
        ctx.cm = (c, mode)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f32c(g)
        da, db = torch.empty_like(a), torch.empty_like(b)
        hmx.check(_L().hmx_binary_bwd(hmx.ptr(a), hmx.ptr(b), hmx.ptr(g), a.numel(), ctx.cm[0], ctx.cm[1], hmx.ptr(da),
                                      hmx.ptr(db), _s()), "hmx_binary_bwd")
        return da, db, None, None


def mul_scale(a, b, c):
    return _Binary.apply(a, b, float(c), 0)


def swiglu(gate, up):
    return _Binary.apply(gate, up, 1.0, 1)


# ------------------------------------------------------------------------------- conv + scan
class _ConvSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, L):
        _need_cuda(x, w, b)
        x, w, b = _f32c(x), _f32c(w), _f32c(b)
        T, C = x.shape
        xc = torch.empty_like(x)
        hmx.check(_L().hmx_conv_silu_fwd(hmx.ptr(x), hmx.ptr(w), hmx.ptr(b), T, L, C, hmx.ptr(xc), _s()),
                  "hmx_conv_silu_fwd")
        ctx.save_for_backward(x, w, b)
        ctx.L = L
        return xc

    @staticmethod
    def backward(ctx, dxc):
        x, w, b = ctx.saved_tensors
        T, C = x.shape
        B = T // ctx.L
        dxc = _f32c(dxc)
        dx, part = torch.empty_like(x), _empty(B, C * 5, like=x)
        hmx.check(_L().hmx_conv_silu_bwd(hmx.ptr(x), hmx.ptr(w), hmx.ptr(b), hmx.ptr(dxc), B, ctx.L, C, hmx.ptr(dx),
                                         hmx.ptr(part), _s()), "hmx_conv_silu_bwd")
        pg = colsum(part)                    # [dw (C, 4) | db (C)]
        return dx, pg[:4 * C].view(C, 4), pg[4 * C:], None


_ZEROS = {}


def _zeros(n, device):
    """A shared read-only block of n zeros."""
    key = (str(device),)
    z = _ZEROS.get(key)
    if z is None or z.numel() < n:
        z = _ZEROS[key] = torch.zeros(max(n, 4096), dtype=torch.float32, device=device)
    return z[:n]


class _HeadParams(torch.autograd.Function):
    """The per-head conv / scan parameters of a block, stacked over heads: one copy launch for all of them; the
    gradients go back to the heads as slices of the stacked gradients (no launch).  rows > a head's own channel
    count: every head's block is followed by zero rows up to `rows` channels (widened layout; zero parameters keep
    the added channels at zero through conv and scan)."""

    @staticmethod
    def forward(ctx, H, rows, *params):
        groups = [params[i:i + H] for i in range(0, len(params), H)]
        pieces = []
        for q in params:
            pieces.append(q.detach().reshape(-1))
            extra = (rows - q.shape[0]) * (q.numel() // q.shape[0])
            if extra:
                pieces.append(_zeros(extra, q.device))
        flat = torch.cat(pieces)
        outs, at = [], 0
        for g in groups:
            n = H * rows * (g[0].numel() // g[0].shape[0])
            outs.append(flat[at:at + n].view(H * rows, *g[0].shape[1:]))
            at += n
        ctx.H, ctx.rows = H, rows
        ctx.shapes = [q.shape for q in params]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        H, rows, out = ctx.H, ctx.rows, []
        for gi, g in enumerate(grads):
            for h in range(H):
                shape = ctx.shapes[gi * H + h]
                out.append(None if g is None else g[h * rows:h * rows + shape[0]].view(shape))
        return (None, None, *out)


def head_params(blocks, rows=None):
    """(conv weight (C, 4), conv bias (C), W_dt (C, 16), b_dt (C), A_log (C, 16), D (C)) of the heads' MambaBlocks,
    channels of head h at rows h * rows ... h * rows + ED (rows = ED unless given: see _HeadParams)."""
    per = [[m.conv1d.weight.view(m.conv1d.weight.shape[0], -1) for m in blocks], [m.conv1d.bias for m in blocks],
           [m.ssm.dt_proj_layer.weight for m in blocks], [m.ssm.dt_proj_layer.bias for m in blocks],
           [m.ssm.A_log for m in blocks], [m.ssm.D for m in blocks]]
    ED = per[1][0].shape[0]
    rows = ED if rows is None else rows
    if len(blocks) == 1 and rows == ED:
        return tuple(g[0] for g in per)
    return _HeadParams.apply(len(blocks), rows, *[q for g in per for q in g])


def conv_silu(x, w, b, L):
    """x (T, C), w (C, 4), b (C): causal depthwise conv over each length-L sequence, then SiLU."""
    return _ConvSilu.apply(x, w, b, L)


class _Scan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xc, z, dbc, Wdt, bdt, A_log, D, L, ED, H):
        _need_cuda(xc, z, dbc)
        xc, dbc, Wdt, bdt, A_log, D = (_f32c(t) for t in (xc, dbc, Wdt, bdt, A_log, D))
        z = _f32c(z) if z is not None else None
        T, C = xc.shape
        B = T // L
        if ED < 32 or ED & (ED - 1) or C != ED * H:
            raise hmx.HmxError(f"selective scan needs 4*head_dim to be a power of two >= 32 (got {ED})")
        keep = any(ctx.needs_input_grad)
        hs = _empty(max(int(_L().hmx_scan_state_floats(B, L, C)), 1), like=xc) if keep else None
        yg = torch.empty_like(xc)
        hmx.check(_L().hmx_scan_fwd(hmx.ptr(xc), hmx.ptr(z), hmx.ptr(dbc), hmx.ptr(Wdt), hmx.ptr(bdt), hmx.ptr(A_log),
                                    hmx.ptr(D), B, L, C, ED, H, hmx.ptr(yg), hmx.ptr(hs), _s()), "hmx_scan_fwd")
        if keep:
            ctx.save_for_backward(xc, z, dbc, Wdt, bdt, A_log, D, hs)
        ctx.dims = (B, L, C, ED, H)
        return yg

    @staticmethod
    def backward(ctx, dyg):
        xc, z, dbc, Wdt, bdt, A_log, D, hs = ctx.saved_tensors
        B, L, C, ED, H = ctx.dims
        T = B * L
        dyg = _f32c(dyg)
        gph = _L().hmx_scan_groups_per_head(ED)
        dxc = torch.empty_like(xc)
        dz = torch.empty_like(z) if z is not None else None
        dpart = _empty(T * H, gph, P_DBC, like=xc)
        ppart = _empty(B, C * 34, like=xc)
        hmx.check(_L().hmx_scan_bwd(hmx.ptr(xc), hmx.ptr(z), hmx.ptr(dbc), hmx.ptr(Wdt), hmx.ptr(bdt), hmx.ptr(A_log),
                                    hmx.ptr(D), hmx.ptr(hs), hmx.ptr(dyg), B, L, C, ED, H, hmx.ptr(dxc), hmx.ptr(dz),
                                    hmx.ptr(dpart), hmx.ptr(ppart), _s()), "hmx_scan_bwd")
        if gph == 1:
            ddbc = dpart.view(T, H * P_DBC)
        else:
            ddbc = _empty(T, H * P_DBC, like=xc)
            hmx.check(_L().hmx_sum_groups_f32(hmx.ptr(dpart), T * H, gph, P_DBC, hmx.ptr(ddbc), _s()),
                      "hmx_sum_groups_f32")
        pg = colsum(ppart)                   # [dA_log (C, 16) | dW_dt (C, 16) | db_dt (C) | dD (C)]
        return (dxc, dz, ddbc, pg[16 * C:32 * C].view(C, 16), pg[32 * C:33 * C], pg[:16 * C].view(C, 16), pg[33 * C:],
                None, None, None)


class _ProjScan(torch.autograd.Function):
    """deltaBC projection of every head followed by the gated scan, as one node: in the backward pass the projection's
    input gradient is added into the scan's dxc by the product itself (xc feeds both, autograd would add them)."""

    @staticmethod
    def forward(ctx, xc, z, Wdt, bdt, A_log, D, L, ED, H, EDr, *Wdbc):
        """ED: channels per head of xc / z / the stacked parameters; EDr <= ED: a head's true channel count, the
        width of its deltaBC weight (48, EDr) -- channels EDr .. ED of a head are the zero channels of a widened
        layout and do not enter the projection."""
        _need_cuda(xc, z)
        xc, Wdt, bdt, A_log, D = (_f32c(t) for t in (xc, Wdt, bdt, A_log, D))
        z = _f32c(z) if z is not None else None
        ws = [_f32c(q) for q in Wdbc]
        T, C = xc.shape
        B = T // L
        if ED < 32 or ED & (ED - 1) or C != ED * H or len(ws) != H or any(w.shape != (P_DBC, EDr) for w in ws):
            raise hmx.HmxError(f"selective scan needs 4*head_dim to be a power of two >= 32 (got {ED})")
        dbc = _empty(T, H * P_DBC, like=xc)
        _heads_linear_raw(xc.data_ptr(), C, ED, ws, EDr, 1, T, P_DBC, EDr, dbc.data_ptr(), H * P_DBC, P_DBC)
        keep = any(ctx.needs_input_grad)
        hs = _empty(max(int(_L().hmx_scan_state_floats(B, L, C)), 1), like=xc) if keep else None
        yg = torch.empty_like(xc)
        hmx.check(_L().hmx_scan_fwd(hmx.ptr(xc), hmx.ptr(z), hmx.ptr(dbc), hmx.ptr(Wdt), hmx.ptr(bdt), hmx.ptr(A_log),
                                    hmx.ptr(D), B, L, C, ED, H, hmx.ptr(yg), hmx.ptr(hs), _s()), "hmx_scan_fwd")
        if keep:
            ctx.save_for_backward(xc, z, dbc, Wdt, bdt, A_log, D, hs, *ws)
        ctx.dims = (B, L, C, ED, H, EDr)
        ctx.params = Wdbc
        return yg

    @staticmethod
    def backward(ctx, dyg):
        xc, z, dbc, Wdt, bdt, A_log, D, hs, *ws = ctx.saved_tensors
        B, L, C, ED, H, EDr = ctx.dims
        T = B * L
        dyg = _f32c(dyg)
        gph = _L().hmx_scan_groups_per_head(ED)
        dxc = torch.empty_like(xc)
        dz = torch.empty_like(z) if z is not None else None
        dpart = _empty(T * H, gph, P_DBC, like=xc)
        ppart = _empty(B, C * 34, like=xc)
        hmx.check(_L().hmx_scan_bwd(hmx.ptr(xc), hmx.ptr(z), hmx.ptr(dbc), hmx.ptr(Wdt), hmx.ptr(bdt), hmx.ptr(A_log),
                                    hmx.ptr(D), hmx.ptr(hs), hmx.ptr(dyg), B, L, C, ED, H, hmx.ptr(dxc), hmx.ptr(dz),
                                    hmx.ptr(dpart), hmx.ptr(ppart), _s()), "hmx_scan_bwd")
        if gph == 1:
            ddbc = dpart.view(T, H * P_DBC)
        else:
            ddbc = _empty(T, H * P_DBC, like=xc)
            hmx.check(_L().hmx_sum_groups_f32(hmx.ptr(dpart), T * H, gph, P_DBC, hmx.ptr(ddbc), _s()),
                      "hmx_sum_groups_f32")
        dws = []
        _heads_linear_raw(ddbc.data_ptr(), H * P_DBC, P_DBC, ws, 1, EDr, T, EDr, P_DBC, dxc.data_ptr(), C, ED,
                          accumulate=True)
        for h, w in enumerate(ws):
            gp = ddbc.data_ptr() + 4 * h * P_DBC
            if not ctx.needs_input_grad[10 + h]:
                dws.append(None)
                continue
            tw = _param_grad_target(ctx.params[h])
            if tw is not None:
                _wgrad_queue.add(ddbc, h * P_DBC, H * P_DBC, xc, h * ED, C, None, T, P_DBC, EDr, tw, 0, EDr, None, 0)
                dws.append(None)
                continue
            dw = torch.empty_like(w)
            _wgrad_raw(gp, H * P_DBC, xc.data_ptr() + 4 * h * ED, C, None, T, P_DBC, EDr, hmx.ptr(dw), EDr, None, xc.device)
            dws.append(dw)
        pg = colsum(ppart)                   # [dA_log (C, 16) | dW_dt (C, 16) | db_dt (C) | dD (C)]
        return (dxc, dz, pg[16 * C:32 * C].view(C, 16), pg[32 * C:33 * C], pg[:16 * C].view(C, 16), pg[33 * C:],
                None, None, None, None, *dws)


def _scan_width(ED):
    """The scan kernels group a head's channels in powers of two (>= 32)."""
    p = 32
    while p < ED:
        p *= 2
    return p


def _pad_channels(t, H, ED, EDp, channel_dim):
    """An axis of H * ED channels (head-major) widened to H * EDp with zeros (differentiable)."""
    F = torch.nn.functional
    if channel_dim == 0:                                   # (H * ED, ...) parameters
        rest = t.shape[1:]
        pad = (0, 0) * len(rest) + (0, EDp - ED)
        return F.pad(t.reshape(H, ED, *rest), pad).reshape(H * EDp, *rest)
    T = t.shape[0]                                         # (T, H * ED) activations
    return F.pad(t.reshape(T, H, ED), (0, EDp - ED)).reshape(T, H * EDp)


def _padded_scan(xc, z, Wdt, bdt, A_log, D, L, ED, H, dbc=None, Wdbc=None):
    """Inner widths that are not a power of two (the reference's default hidden size 60 gives 120): the head's channels
    are widened with zero channels -- zero input, zero gate, zero parameters, so they stay zero and add nothing to the
    shared B / C / dt gradients -- and the result is cut back.  Same values as the unpadded recurrence."""
    EDp = _scan_width(ED)
    T = xc.shape[0]
    xcp = _pad_channels(xc, H, ED, EDp, 1)
    zp = _pad_channels(z, H, ED, EDp, 1) if z is not None else None
    Wdt, bdt, A_log, D = (_pad_channels(t, H, ED, EDp, 0) for t in (Wdt, bdt, A_log, D))
    if Wdbc is not None:
        y = _ProjScan.apply(xcp, zp, Wdt, bdt, A_log, D, L, EDp, H, ED, *Wdbc)
    else:
        y = _Scan.apply(xcp, zp, dbc, Wdt, bdt, A_log, D, L, EDp, H)
    return y.reshape(T, H, EDp)[..., :ED].reshape(T, H * ED)


def proj_scan_gated(xc, z, Wdbc, Wdt, bdt, A_log, D, L, ED, H, ED_real=None):
    """selective_scan_gated(xc, z, heads_linear(xc, Wdbc), ...): Wdbc is one (48, ED) deltaBC weight per head.
    ED_real < ED: inputs and stacked parameters are already in a widened layout (ED channels per head, the last
    ED - ED_real of them zero) and the deltaBC weights are the true (48, ED_real) ones."""
    if ED_real is not None and ED_real != ED:
        return _ProjScan.apply(xc, z, Wdt, bdt, A_log, D, L, ED, H, ED_real, *Wdbc)
    if ED != _scan_width(ED):
        return _padded_scan(xc, z, Wdt, bdt, A_log, D, L, ED, H, Wdbc=Wdbc)
    return _ProjScan.apply(xc, z, Wdt, bdt, A_log, D, L, ED, H, ED, *Wdbc)


def selective_scan_gated(xc, z, dbc, Wdt, bdt, A_log, D, L, ED, H):
    """(scan(xc; dbc, params) + D xc) * SiLU(z) for all heads (z = None: no gate); see hmx_scan_fwd."""
    if ED != _scan_width(ED):
        return _padded_scan(xc, z, Wdt, bdt, A_log, D, L, ED, H, dbc=dbc)
    return _Scan.apply(xc, z, dbc, Wdt, bdt, A_log, D, L, ED, H)


class _Attn1Q(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, drop, L, H):
        _need_cuda(q, kv)
        q, kv = _f32c(q), _f32c(kv)
        B, d = q.shape
        assert kv.shape == (B * L, 2 * d)
        drop = _f32c(drop) if drop is not None else None
        p = _empty(B, H, L, like=q)
        out = _empty(B, d, like=q)
        hmx.check(_L().hmx_attn1q_fwd(hmx.ptr(q), hmx.ptr(kv), hmx.ptr(drop), B, L, d, H, hmx.ptr(p), hmx.ptr(out), _s()),
                  "hmx_attn1q_fwd")
        ctx.save_for_backward(q, kv, drop, p)
        ctx.dims = (B, L, d, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, drop, p = ctx.saved_tensors
        B, L, d, H = ctx.dims
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        hmx.check(_L().hmx_attn1q_bwd(hmx.ptr(q), hmx.ptr(kv), hmx.ptr(drop), hmx.ptr(p), hmx.ptr(_f32c(dout)), B, L, d, H,
                                      hmx.ptr(dq), hmx.ptr(dkv), _s()), "hmx_attn1q_bwd")
        return dq, dkv, None, None, None


def attn1q(q, kv, drop, L, H):
    """One query per sequence over L projected [k | v] rows (hydra3.py:528-533); drop: (B, H, L) multiplier or None."""
    return _Attn1Q.apply(q, kv, drop, L, H)


class _PScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, X):
        A, X = A.contiguous(), X.contiguous()
        Bn, L = A.shape[0], A.shape[1]
        C = A[0, 0].numel() if A.numel() else 0
        H = torch.empty_like(X)
        hmx.check(_L().hmx_pscan_fwd(hmx.ptr(A), hmx.ptr(X), Bn, L, C, hmx.ptr(H), _s()), "hmx_pscan_fwd")
        ctx.save_for_backward(A, H)
        return H

    @staticmethod
    def backward(ctx, dH):
        A, H = ctx.saved_tensors
        dH = dH.contiguous()
        Bn, L = A.shape[0], A.shape[1]
        C = A[0, 0].numel() if A.numel() else 0
        dA, dX = torch.empty_like(A), torch.empty_like(A)
        hmx.check(_L().hmx_pscan_bwd(hmx.ptr(A), hmx.ptr(H), hmx.ptr(dH), Bn, L, C, hmx.ptr(dA), hmx.ptr(dX), _s()),
                  "hmx_pscan_bwd")
        return dA, dX


def pscan(A, X):
    """hydra3.py:104: H[:, t] = A[:, t] * H[:, t-1] + X[:, t] for A, X of shape (B, L, ED, N)."""
    if not (A.is_cuda and X.is_cuda) or A.dtype != torch.float32 or X.dtype != torch.float32 or A.shape != X.shape:
        raise hmx.HmxError("pscan expects two CUDA float32 tensors of the same (B, L, ...) shape")
    return _PScan.apply(A, X)


def selective_scan(x, delta, A, B, C, D):
    """hydra3.py:107-123 with the reference's own decomposition: deltaA = exp(delta (x) A), BX = delta * B * x,
    hs = pscan(deltaA, BX) on HIP, y = <hs, C> + D * x.  (The model itself uses the fused scan kernel, which never
    forms the (B, L, ED, N) tensors; this free function keeps the reference's call shape.)"""
    dA = torch.exp(delta.unsqueeze(-1) * A)
    bx = (delta.unsqueeze(-1) * B.unsqueeze(2)) * x.unsqueeze(-1)
    hs = pscan(dA, bx)
    return (hs * C.unsqueeze(2)).sum(-1) + D * x


def seq_index(ids, ts, time_span):
    """(B, L) ids / timestamps -> padding mask (T,) float, position ids (T,), time bins (T,) (no gradient)."""
    _need_cuda(ids, ts)
    ids, ts = _i64c(ids), _i64c(ts)
    B, L = ids.shape
    T = B * L
    mask = torch.empty(T, dtype=torch.float32, device=ids.device)
    pos = torch.empty(T, dtype=torch.int64, device=ids.device)
    tb = torch.empty(T, dtype=torch.int64, device=ids.device)
    hmx.check(_L().hmx_seq_index(hmx.ptr(ids), hmx.ptr(ts), T, L, time_span, hmx.ptr(mask), hmx.ptr(pos), hmx.ptr(tb),
                                 _s()), "hmx_seq_index")
    return mask, pos, tb
