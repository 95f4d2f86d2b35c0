"""torch.autograd wrappers of the fused block halves of libhmx.so (hmx_block_*).

A Hydra block (hydra3.py:392-457) runs as   pre half -> conv / deltaBC / scan (enc_ops) -> post half.
The two halves are one kernel each per direction; see include/hmx.h for what each covers.  CUDA fp32 only.
"""
import ctypes

import torch

from . import enc_ops as E
from . import hmx
from .ops import _f32c, workspace


def supported(d, n_heads):
    """The fused halves take this width as it is (no widening)."""
    return bool(hmx.lib().hmx_block_supported(d, n_heads))


PAD_TO_FUSED = True      # widths whose head_dim is not a multiple of 32 run on the fused halves in a widened layout


class Layout:
    """Feature layout of a (d, H) block on the fused halves: every head's hd = d / H features are followed by zero
    features up to hdp = the next multiple of 32, its 4 hd inner channels by zero channels up to 4 hdp
    (d = 60, H = 2: 64 features, 256 channels).  dp == d: nothing is widened."""
    _cache = {}

    def __init__(self, d, H, dp):
        self.d, self.H, self.dp = d, H, dp
        self.hd, self.hdp = d // H, dp // H
        self.ED, self.EDp = 4 * self.hd, 4 * self.hdp
        self.widened = dp != d

    @staticmethod
    def of(d, H):
        """Layout of a (d, H) block, or None when the fused halves do not take it."""
        key = (d, H, PAD_TO_FUSED)
        if key not in Layout._cache:
            dp = int(hmx.lib().hmx_block_padded_width(d, H))
            if dp == 0 or (dp != d and not PAD_TO_FUSED):
                Layout._cache[key] = None
            else:
                Layout._cache[key] = Layout(d, H, dp)
        return Layout._cache[key]

    def widen(self, x2):
        """(T, d) -> (T, dp), zeros in the added features (differentiable)."""
        if not self.widened:
            return x2
        T = x2.shape[0]
        return torch.nn.functional.pad(x2.reshape(T, self.H, self.hd), (0, self.hdp - self.hd)).reshape(T, self.dp)

    def narrow(self, x2p):
        """(T, dp) -> (T, d): drops the added features."""
        if not self.widened:
            return x2p
        T = x2p.shape[0]
        return x2p.reshape(T, self.H, self.hdp)[..., :self.hd].reshape(T, self.d)


class PackedBlock:
    """bf16 hi / lo fragment images of a block's projection weights (both orientations), packed by one launch.
    Valid for the weights' current values: built once per forward pass and reused by that pass's backward."""

    def __init__(self, blk, lay):
        mhm = blk.multiheadMamba
        heads = list(mhm.mamba)
        self.d, self.H, self.d_real = lay.dp, lay.H, lay.d
        w = hmx.BlockWeights()
        keep = []
        for name, q in (("W1", mhm.linear_layer1.weight), ("W2", mhm.linear_layer2.weight), ("Wg", blk.gate.weight),
                        ("Wu", blk.up.weight), ("Wd", blk.down.weight), ("scale1", blk.norm1.scale),
                        ("b1", mhm.linear_layer1.bias), ("b2", mhm.linear_layer2.bias), ("scale2", blk.norm2.scale)):
            t = _f32c(q.detach())
            keep.append(t)
            setattr(w, name, t.data_ptr())
        for h, m in enumerate(heads):
            a, b = _f32c(m.in_proj.weight.detach()), _f32c(m.out_proj.weight.detach())
            keep += [a, b]
            w.Win[h], w.Wout[h] = a.data_ptr(), b.data_ptr()
        L = hmx.lib()
        dev = keep[0].device
        self.buf = torch.empty(L.hmx_block_packed_bytes(self.d, self.H), dtype=torch.uint8, device=dev)
        hmx.check(L.hmx_block_pack_f32(ctypes.byref(w), self.d, self.H, self.d_real, hmx.ptr(self.buf),
                                       hmx.current_stream()), "hmx_block_pack_f32")


def _args(T, L, d, H, first, eps, packed, **ptrs):
    a = hmx.BlockArgs()
    a.T, a.L, a.d, a.H, a.first, a.eps = T, L, d, H, int(first), eps
    a.d_real = packed.d_real
    a.packed = packed.buf.data_ptr()
    for k, v in ptrs.items():
        setattr(a, k, None if v is None else v.data_ptr())
    return a


def _scale_grad(part):
    return E.colsum(part)


def _square_jobs(lay, pi, dy, x, mask, T, bi=None):
    """Weight-gradient job of a (d, d) projection whose operands dy, x are (T, dp).  Widened operands are read as they
    are (whole aligned rows); the product's outputs map to the true weight by head groups (hdp columns, hd of them
    real)."""
    if not lay.widened:
        return [(pi, dy, 0, lay.d, x, 0, lay.d, mask, T, lay.d, lay.d, 0, lay.d, bi, 0)]
    hd, hdp, d, dp = lay.hd, lay.hdp, lay.d, lay.dp
    return [(pi, dy, 0, dp, x, 0, dp, mask, T, dp, dp, 0, d, bi, 0, (hd, hdp, hd, hdp))]


def _weight_grads(jobs, params, device):
    """jobs: (param index, dy, dy_off, ldy, x, x_off, ldx, in_mask, T, N, K, dw_off, ldw, bias param index or None,
    bias offset).  Direct-gradient mode queues them (added into .grad by the batched launch); otherwise one batched
    launch writes fresh gradient tensors, returned by position (None where the mode handled it or no gradient is
    needed)."""
    grads = [None] * len(params)
    immediate, descs, keep = [], [], []
    for job in jobs:
        (pi, dy, dy_off, ldy, x, x_off, ldx, mask, T, N, K, dw_off, ldw, bi, db_off), groups = job[:15], tuple(job[15:16])
        groups = groups[0] if groups else (0, 0, 0, 0)
        q = params[pi]
        if q is None or not q.requires_grad:
            continue
        bq = params[bi] if bi is not None else None
        want_b = bq is not None and bq.requires_grad
        tw = E._param_grad_target(q)
        tb = E._param_grad_target(bq) if want_b else None
        if tw is not None and (tb is not None or not want_b):
            E._wgrad_queue.add(dy, dy_off, ldy, x, x_off, ldx, mask, T, N, K, tw, dw_off, ldw, tb, db_off, groups)
            continue
        if grads[pi] is None:
            grads[pi] = torch.zeros_like(q, memory_format=torch.contiguous_format)
        if want_b and grads[bi] is None:
            grads[bi] = torch.zeros_like(bq)
        immediate.append((dy, dy_off, ldy, x, x_off, ldx, mask, T, N, K, grads[pi], dw_off, ldw, grads[bi] if want_b else None,
                          db_off, groups))
    if immediate:
        for (dy, dy_off, ldy, x, x_off, ldx, mask, T, N, K, dw, dw_off, ldw, db, db_off, groups) in immediate:
            descs.append(hmx.WgradDesc(dy.data_ptr() + 4 * dy_off, ldy, x.data_ptr() + 4 * x_off, ldx,
                                       mask.data_ptr() if mask is not None else None, T, N, K,
                                       dw.data_ptr() + 4 * dw_off, ldw,
                                       db.data_ptr() + 4 * db_off if db is not None else None, 1, *groups))
        for i in range(0, len(descs), 24):
            chunk = descs[i:i + 24]
            arr = (hmx.WgradDesc * len(chunk))(*chunk)
            lib = hmx.lib()
            ws = workspace(lib.hmx_linear_wgrad_batch_workspace_bytes(arr, len(chunk)), device)
            hmx.check(lib.hmx_linear_wgrad_batch_f32(arr, len(chunk), hmx.ptr(ws), ws.numel(), hmx.current_stream()),
                      "hmx_linear_wgrad_batch_f32")
    return grads


class _Link:
    """Carries dm (the gradient of the RMSNorm2 input) from the post half's backward to the pre half's backward,
    which owns both residual paths of the block input (x itself, and in the first block the normalised n)."""
    dm = None


def _scale_grad(part, lay):
    return lay.narrow(E.colsum(part).unsqueeze(0)).reshape(lay.d)


class _BlockPre(torch.autograd.Function):
    """All activations (T, dp) / (T, H * EDp) in the layout `lay`; parameters in their true shapes."""

    @staticmethod
    def forward(ctx, x, drop, mask, packed, link, lay, L, first, eps, scale1, W1, b1, *Win):
        x = _f32c(x)
        T, d = x.shape
        H = lay.H
        C = 4 * d
        cos, sin = E.rope_tables(L, lay.hd, x.device, pairs=lay.hdp // 2)
        mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=x.device)   # noqa: E731
        rstd1, x_in, z, y2 = mk(T), mk(T, C), mk(T, C), mk(T, d)
        sc, b = _f32c(scale1.detach()), _f32c(b1.detach())
        a = _args(T, L, d, H, first, eps, packed, mask=mask, x=x, drop=drop, scale1=sc, b1=b, rope_cos=cos, rope_sin=sin,
                  rstd1=rstd1, x_in=x_in, z=z, y2=y2)
        hmx.check(hmx.lib().hmx_block_pre_fwd(ctypes.byref(a), hmx.current_stream()), "hmx_block_pre_fwd")
        ctx.save_for_backward(x, drop, mask, sc, b)
        ctx.cfg = (L, lay, first, eps, packed, link)
        ctx.params = (scale1, W1, b1) + tuple(Win)
        ctx.mark_non_differentiable(rstd1)
        ctx.set_materialize_grads(False)      # backward takes None for absent gradients: no (T) zeros built for rstd1
        return x_in, z, y2, rstd1

    @staticmethod
    def backward(ctx, dx_in, dz, dy2, _):
        x, drop, mask, sc, b = ctx.saved_tensors
        L, lay, first, eps, packed, link = ctx.cfg
        T, d = x.shape
        H = lay.H
        C = 4 * d
        dev = x.device
        mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)   # noqa: E731
        zeros = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)   # noqa: E731
        dx_in = _f32c(dx_in) if dx_in is not None else zeros(T, C)
        dz = _f32c(dz) if dz is not None else zeros(T, C)
        dy2 = _f32c(dy2) if dy2 is not None else zeros(T, d)
        dm = link.dm if link.dm is not None else zeros(T, d)
        link.dm = None
        cos, sin = E.rope_tables(L, lay.hd, dev, pairs=lay.hdp // 2)
        rows = int(hmx.lib().hmx_block_scale_partial_rows(T))
        dx, s, dx1, n, part = mk(T, d), mk(T, d), mk(T, d), mk(T, d), mk(rows, d)   # the launch zeroes the rows its waves do not write
        a = _args(T, L, d, H, first, eps, packed, mask=mask, x=x, drop=drop, scale1=sc, b1=b, rope_cos=cos, rope_sin=sin,
                  dx_in=dx_in, dz=dz, dy2=dy2, dm=dm, dx=dx, s=s, dx1=dx1, n=n, dscale_part=part)
        hmx.check(hmx.lib().hmx_block_pre_bwd(ctypes.byref(a), hmx.current_stream()), "hmx_block_pre_bwd")
        params = ctx.params            # (scale1, W1, b1, Win_0 ...)
        hd, hdp, ED, EDp = lay.hd, lay.hdp, lay.ED, lay.EDp
        jobs = _square_jobs(lay, 1, dx1, n, mask, T, bi=2)
        for h in range(H):             # in_proj (2 ED, hd): rows [x | z]
            jobs.append((3 + h, dx_in, h * EDp, C, s, h * hdp, d, None, T, ED, hd, 0, hd, None, 0))
            jobs.append((3 + h, dz, h * EDp, C, s, h * hdp, d, None, T, ED, hd, ED * hd, hd, None, 0))
        g = _weight_grads(jobs, params, dev)
        g[0] = _scale_grad(part, lay) if params[0].requires_grad else None
        return (dx, None, None, None, None, None, None, None, None) + tuple(g)


class _BlockPost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, yg, y2, x, drop, rstd1, mask, packed, link, lay, L, first, eps, scale1, scale2, W2, b2, Wg, Wu, Wd,
                *Wout):
        yg, y2, x = _f32c(yg), _f32c(y2), _f32c(x)
        T, d = x.shape
        mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=x.device)   # noqa: E731
        out, m, rstd2 = mk(T, d), mk(T, d), mk(T)
        sc1, sc2, bb = _f32c(scale1.detach()), _f32c(scale2.detach()), _f32c(b2.detach())
        a = _args(T, L, d, lay.H, first, eps, packed, mask=mask, x=x, drop=drop, scale1=sc1, rstd1=rstd1, yg=yg, y2=y2,
                  b2=bb, scale2=sc2, out=out, m=m, rstd2=rstd2)
        hmx.check(hmx.lib().hmx_block_post_fwd(ctypes.byref(a), hmx.current_stream()), "hmx_block_post_fwd")
        ctx.save_for_backward(yg, y2, mask, m, rstd2, sc2)
        ctx.cfg = (L, lay, first, eps, packed, link)
        ctx.params = (scale2, W2, b2, Wg, Wu, Wd) + tuple(Wout)
        return out

    @staticmethod
    def backward(ctx, dout):
        yg, y2, mask, m, rstd2, sc2 = ctx.saved_tensors
        L, lay, first, eps, packed, link = ctx.cfg
        T, d = m.shape
        H = lay.H
        C = 4 * d
        dev = m.device
        dout = _f32c(dout)
        mk = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)   # noqa: E731
        rows = int(hmx.lib().hmx_block_scale_partial_rows(T))
        dyg, dy2, dm = mk(T, C), mk(T, d), mk(T, d)
        hmid, n2, dg, du, pp, dy1 = mk(T, d), mk(T, d), mk(T, d), mk(T, d), mk(T, d), mk(T, d)
        part = mk(rows, d)             # the launch zeroes the rows its waves do not write
        a = _args(T, L, d, H, first, eps, packed, mask=mask, yg=yg, y2=y2, scale2=sc2, m=m, rstd2=rstd2, dout=dout,
                  dyg=dyg, dy2=dy2, dm=dm, hmid=hmid, n2=n2, dg=dg, du=du, p=pp, dy1=dy1, dscale_part=part)
        hmx.check(hmx.lib().hmx_block_post_bwd(ctypes.byref(a), hmx.current_stream()), "hmx_block_post_bwd")
        link.dm = dm
        params = ctx.params            # (scale2, W2, b2, Wg, Wu, Wd, Wout_0 ...)
        hd, hdp, ED, EDp = lay.hd, lay.hdp, lay.ED, lay.EDp
        jobs = (_square_jobs(lay, 1, dm, pp, None, T, bi=2) + _square_jobs(lay, 3, dg, n2, None, T) +
                _square_jobs(lay, 4, du, n2, None, T) + _square_jobs(lay, 5, dout, hmid, mask, T))
        for h in range(H):             # out_proj (hd, ED)
            jobs.append((6 + h, dy1, h * hdp, d, yg, h * EDp, C, None, T, hd, ED, 0, ED, None, 0))
        g = _weight_grads(jobs, params, dev)
        g[0] = _scale_grad(part, lay) if params[0].requires_grad else None
        # x and (first block) n receive their share of dm in the pre half's backward
        return (dyg, dy2, None, None, None, None, None, None, None, None, None, None, None) + tuple(g)


class drop_pool:
    """While active, the dropout multipliers of the fused blocks are slices of one buffer drawn by a single launch
    (n_elems: how many the blocks inside will ask for in total; a block that finds too few left draws its own)."""
    _active = None

    def __init__(self, n_elems, pdrop, device, enabled=True):
        self.n, self.p, self.device, self.enabled = int(n_elems), float(pdrop), device, enabled and pdrop > 0.0
        self.buf, self.at = None, 0

    def __enter__(self):
        if self.enabled:
            from .feature_ops import dropout_multiplier
            self.buf = dropout_multiplier((self.n,), self.p, self.device)
            self.prev, drop_pool._active = drop_pool._active, self
        return self

    def __exit__(self, *exc):
        if self.enabled:
            drop_pool._active = self.prev
        return False

    @staticmethod
    def take(T, d, pdrop, device):
        pool = drop_pool._active
        n = T * d
        if pool is not None and pool.p == pdrop and pool.at + n <= pool.n and pool.buf.device == device:
            out = pool.buf[pool.at:pool.at + n].view(T, d)
            pool.at += (n + 3) // 4 * 4
            return out
        from .feature_ops import dropout_multiplier
        return dropout_multiplier((T, d), pdrop, device)


def hydra_block(blk, x2, mask1, L, first, lay):
    """One HydraBlockFirst / HydraBlock on the fused halves.  x2 (T, lay.dp) in the layout `lay` (the added features
    of a widened layout zero), mask1 (T) or None -> (T, lay.dp), added features zero again."""
    mhm = blk.multiheadMamba
    heads = list(mhm.mamba)
    H = len(heads)
    T, d = x2.shape
    assert d == lay.dp and H == lay.H
    packed = PackedBlock(blk, lay)
    link = _Link()
    drop = None
    pdrop = blk.dropout.p
    if blk.training and pdrop > 0.0:
        fixed = blk.fixed_drop_multiplier
        if fixed is not None:
            fixed = fixed.reshape(T, -1)
            drop = _f32c(lay.widen(fixed) if fixed.shape[1] == lay.d else fixed)
        else:
            drop = drop_pool.take(T, d, pdrop, x2.device)
    eps = blk.norm1.eps
    x_in, z, y2, rstd1 = _BlockPre.apply(x2, drop, mask1, packed, link, lay, L, first, eps, blk.norm1.scale,
                                         mhm.linear_layer1.weight, mhm.linear_layer1.bias,
                                         *[m.in_proj.weight for m in heads])
    conv_w, conv_b, w_dt, b_dt, a_log, dskip = E.head_params(heads, rows=lay.EDp)
    xc = E.conv_silu(x_in, conv_w, conv_b, L)
    yg = E.proj_scan_gated(xc, z, [m.ssm.deltaBC_layer.weight for m in heads], w_dt, b_dt, a_log, dskip, L, lay.EDp, H,
                           ED_real=lay.ED)
    return _BlockPost.apply(yg, y2, x2, drop, rstd1, mask1, packed, link, lay, L, first, blk.norm2.eps, blk.norm1.scale,
                            blk.norm2.scale, mhm.linear_layer2.weight, mhm.linear_layer2.bias, blk.gate.weight,
                            blk.up.weight, blk.down.weight, *[m.out_proj.weight for m in heads])
