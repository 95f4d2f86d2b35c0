"""torch.autograd wrappers of the feature-fusion row kernels of libhmx.so (csrc/features.hip).

CUDA fp32 tensors only; no CPU fallback.
"""
import torch

from . import enc_ops as E
from . import hmx, ops
from .enc_ops import _L, _empty, _s, colsum, linear
from .ops import _f32c, _i64c, _need_cuda

RELU, GELU = 0, 1


class _GatherSumRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, ids, padding_idx):
        _need_cuda(weight, ids)
        w, idc = _f32c(weight), _i64c(ids)
        bag = idc.shape[-1]
        n = 1
        for extent in idc.shape[:-1]:     # bags, also when they are empty (bag == 0: every output row is zero)
            n *= extent
        out = torch.empty(*idc.shape[:-1], w.shape[1], dtype=torch.float32, device=w.device)
        hmx.check(_L().hmx_gather_sum_rows_f32(hmx.ptr(w), w.shape[0], w.shape[1], hmx.ptr(idc), n, bag, hmx.ptr(out),
                                               _s()), "hmx_gather_sum_rows_f32")
        ctx.save_for_backward(idc)
        ctx.n_rows, ctx.padding_idx, ctx.param = w.shape[0], -1 if padding_idx is None else padding_idx, weight
        return out

    @staticmethod
    def backward(ctx, g):
        (idc,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None, None
        bag, d = idc.shape[-1], g.shape[-1]
        g2 = _f32c(g).view(-1, d)
        if g2.shape[0] == 0 or bag == 0:
            acc = ops._into_param_grad(ctx.param)
            return (None if acc is not None else torch.zeros(ctx.n_rows, d, device=g.device)), None, None
        # every id of a bag receives the bag's gradient row: rows repeated by the HIP row gather, then the
        # deterministic scatter-add
        src = torch.arange(g2.shape[0] * bag, device=g.device, dtype=torch.int64) // max(bag, 1)
        rows = ops.gather_rows(g2, src)
        acc = ops._into_param_grad(ctx.param)
        gw = ops.scatter_add_rows(rows, idc.reshape(-1), ctx.n_rows, ctx.padding_idx, out=acc)
        return (None if acc is not None else gw), None, None


def gather_sum_rows(weight, ids, padding_idx=None):
    """out[..., :] = sum_a weight[ids[..., a], :] (nn.Embedding followed by .sum over the bag axis)."""
    return _GatherSumRows.apply(weight, ids, padding_idx)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        _need_cuda(x)
        x = _f32c(x)
        out = torch.empty_like(x)
        hmx.check(_L().hmx_act_fwd(hmx.ptr(x), x.numel(), mode, hmx.ptr(out), _s()), "hmx_act_fwd")
        ctx.save_for_backward(x)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.empty_like(x)
        hmx.check(_L().hmx_act_bwd(hmx.ptr(x), hmx.ptr(g), x.numel(), ctx.mode, hmx.ptr(dx), _s()), "hmx_act_bwd")
        return dx, None


def relu(x):
    return _Act.apply(x, RELU)


def gelu(x):
    return _Act.apply(x, GELU)


class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, q):
        _need_cuda(p, q)
        p, q = _f32c(p), _f32c(q)
        assert p.shape == q.shape
        out = torch.empty_like(p)
        hmx.check(_L().hmx_gate_fwd(hmx.ptr(p), hmx.ptr(q), p.numel(), hmx.ptr(out), _s()), "hmx_gate_fwd")
        ctx.save_for_backward(p, q)
        return out

    @staticmethod
    def backward(ctx, g):
        p, q = ctx.saved_tensors
        g = _f32c(g)
        dp, dq = torch.empty_like(p), torch.empty_like(q)
        hmx.check(_L().hmx_gate_bwd(hmx.ptr(p), hmx.ptr(q), hmx.ptr(g), p.numel(), hmx.ptr(dp), hmx.ptr(dq), _s()),
                  "hmx_gate_bwd")
        return dp, dq


def sigmoid_gate(p, q):
    """p * sigmoid(q)."""
    return _Gate.apply(p, q)


class _GateMix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, a, l):
        _need_cuda(z, a, l)
        z, a, l = _f32c(z), _f32c(a), _f32c(l)
        assert z.shape == a.shape == l.shape
        out = torch.empty_like(z)
        hmx.check(_L().hmx_gate_mix_fwd(hmx.ptr(z), hmx.ptr(a), hmx.ptr(l), z.numel(), hmx.ptr(out), _s()), "hmx_gate_mix_fwd")
        ctx.save_for_backward(z, a, l)
        return out

    @staticmethod
    def backward(ctx, g):
        z, a, l = ctx.saved_tensors
        g = _f32c(g)
        dz, da, dl = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        hmx.check(_L().hmx_gate_mix_bwd(hmx.ptr(z), hmx.ptr(a), hmx.ptr(l), hmx.ptr(g), z.numel(), hmx.ptr(dz), hmx.ptr(da),
                                        hmx.ptr(dl), _s()), "hmx_gate_mix_bwd")
        return dz, da, dl


def gate_mix(z, a, l):
    """sigmoid(z) * a + (1 - sigmoid(z)) * l."""
    return _GateMix.apply(z, a, l)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _need_cuda(x, weight, bias)
        x, w, b = _f32c(x), _f32c(weight), _f32c(bias)
        d = x.shape[-1]
        T = x.numel() // d
        y, mean, rstd = torch.empty_like(x), _empty(T, like=x), _empty(T, like=x)
        hmx.check(_L().hmx_layernorm_fwd(hmx.ptr(x), hmx.ptr(w), hmx.ptr(b), T, d, eps, hmx.ptr(y), hmx.ptr(mean),
                                         hmx.ptr(rstd), _s()), "hmx_layernorm_fwd")
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dy = _f32c(dy)
        d = x.shape[-1]
        T = x.numel() // d
        nb = _L().hmx_layernorm_bwd_blocks(T)
        dx, part = torch.empty_like(x), _empty(nb, 2 * d, like=x)
        hmx.check(_L().hmx_layernorm_bwd(hmx.ptr(x), hmx.ptr(w), hmx.ptr(mean), hmx.ptr(rstd), hmx.ptr(dy), T, d,
                                         hmx.ptr(dx), hmx.ptr(part), _s()), "hmx_layernorm_bwd")
        sums = colsum(part)
        return dx, sums[:d], sums[d:], None


def layernorm(x, weight, bias, eps=1e-5):
    return _LayerNorm.apply(x, weight, bias, eps)


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, drop, B, Lq, Lk, H):
        _need_cuda(q, kv, drop)
        q, kv = _f32c(q), _f32c(kv)
        drop = _f32c(drop) if drop is not None else None
        d = q.shape[1]
        assert q.shape[0] == B * Lq and kv.shape == (B * Lk, 2 * d) and (drop is None or drop.numel() == B * H * Lq * Lk)
        p, out = _empty(B, H, Lq, Lk, like=q), _empty(B * Lq, d, like=q)
        hmx.check(_L().hmx_attn_fwd(hmx.ptr(q), hmx.ptr(kv), hmx.ptr(drop), B, Lq, Lk, d, H, hmx.ptr(p), hmx.ptr(out),
                                    _s()), "hmx_attn_fwd")
        ctx.save_for_backward(q, kv, drop, p)
        ctx.dims = (B, Lq, Lk, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, drop, p = ctx.saved_tensors
        B, Lq, Lk, H = ctx.dims
        dout = _f32c(dout)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ds, wv = torch.empty_like(p), torch.empty_like(p)
        hmx.check(_L().hmx_attn_bwd(hmx.ptr(q), hmx.ptr(kv), hmx.ptr(drop), hmx.ptr(p), hmx.ptr(dout), B, Lq, Lk,
                                    q.shape[1], H, hmx.ptr(dq), hmx.ptr(dkv), hmx.ptr(ds), hmx.ptr(wv), _s()),
                  "hmx_attn_bwd")
        return dq, dkv, None, None, None, None, None


def attention(q, kv, drop, B, Lq, Lk, H):
    """Multi-head attention core on projected rows: q (B * Lq, d), kv (B * Lk, 2 d) = [k | v] -> (B * Lq, d);
    drop (B, H, Lq, Lk) or None multiplies the softmax weights."""
    return _Attention.apply(q, kv, drop, B, Lq, Lk, H)


def dropout_multiplier(shape, p_drop, device):
    """Tensor of `shape` holding 0 with probability p_drop and 1 / (1 - p_drop) otherwise (what nn.Dropout multiplies
    its input by in training).  The Philox seed of a draw comes from torch's CPU generator, so torch.manual_seed
    makes a run repeatable; no gradient (a constant of the step)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise hmx.HmxError("dropout_multiplier: the HIP kernels need a GPU device")
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())        # a fresh key per draw; the stream offset stays 0
    with torch.cuda.device(dev):
        hmx.check(_L().hmx_dropout_multiplier_f32(seed, 0, out.numel(), float(p_drop), hmx.ptr(out), _s()),
                  "hmx_dropout_multiplier_f32")
    return out


def dropout(x, p_drop, training):
    """nn.Dropout / F.dropout on (.., d) rows: x times a drawn multiplier (identity in eval mode or at p = 0)."""
    if not training or p_drop <= 0.0:
        return x
    return E.mul_scale(x, dropout_multiplier(x.shape, p_drop, x.device), 1.0)


def row_inv_norm(x2, eps=1e-12):
    """1 / max(|x_t|, eps) per row of (T, K) data: the row scale of F.normalize(p=2), used as the input row scale of
    the projection that follows (no gradient: the rows are input data)."""
    _need_cuda(x2)
    x2 = _f32c(x2)
    out = torch.empty(x2.shape[0], dtype=torch.float32, device=x2.device)
    hmx.check(_L().hmx_row_inv_norm_f32(hmx.ptr(x2), x2.shape[0], x2.shape[1], float(eps), hmx.ptr(out), _s()),
              "hmx_row_inv_norm_f32")
    return out


_KCHUNK = 256      # widest reduction of one hmx_linear_f32 launch


class _WideLinear(torch.autograd.Function):
    """y = (x * in_scale) W[:, col_off:col_off + K]^T + b (+ r1) for any K: the reduction runs in column chunks of
    256, each launch adding onto the one before (fixed order)."""

    @staticmethod
    def forward(ctx, x, weight, bias, in_scale, r1, col_off):
        _need_cuda(x, weight)
        x, w = _f32c(x), _f32c(weight)
        T, K = x.shape
        N, ldw = w.shape
        assert ldw >= col_off + K
        r1 = _f32c(r1) if r1 is not None else None
        y = _empty(T, N, like=x)
        for c0 in range(0, K, _KCHUNK):
            kc = min(_KCHUNK, K - c0)
            first = c0 == 0
            E._linear_raw(hmx.ptr(x) + 4 * c0, K, hmx.ptr(w) + 4 * (col_off + c0), ldw, 1, bias if first else None,
                          T, N, kc, hmx.ptr(y), N, r1 if first else None, 1.0 if (first and r1 is not None) else 0.0,
                          in_mask=in_scale, accumulate=not first)
        ctx.save_for_backward(x, w, in_scale)
        ctx.col_off = col_off
        ctx.has_bias = bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, in_scale = ctx.saved_tensors
        T, K = x.shape
        N, ldw = w.shape
        co = ctx.col_off
        g = _f32c(dy)
        need = ctx.needs_input_grad
        dx = dw = db = None
        if need[0]:
            if in_scale is not None:
                raise hmx.HmxError("no gradient with respect to rows that enter a projection through a row scale "
                                   "(they are input data)")
            dx = _empty(T, K, like=x)
            for c0 in range(0, K, _KCHUNK):
                kc = min(_KCHUNK, K - c0)
                for n0 in range(0, N, _KCHUNK):
                    nc = min(_KCHUNK, N - n0)
                    E._linear_raw(hmx.ptr(g) + 4 * n0, N, hmx.ptr(w) + 4 * (n0 * ldw + co + c0), 1, ldw, None, T, kc, nc,
                                  hmx.ptr(dx) + 4 * c0, K, accumulate=n0 > 0)
        want_db = ctx.has_bias and need[2]
        tw = E._param_grad_target(ctx.wparam) if need[1] else None
        tb = E._param_grad_target(ctx.bparam) if want_db else None
        if tw is not None and (tb is not None or not want_db):
            for c0 in range(0, K, _KCHUNK):
                kc = min(_KCHUNK, K - c0)
                E._wgrad_queue.add(g, 0, N, x, c0, K, in_scale, T, N, kc, tw, co + c0, ldw, tb if c0 == 0 else None, 0)
            want_db = False
        elif need[1]:
            dw = torch.zeros_like(w) if ldw > K else torch.empty_like(w)
            db = _empty(N, like=x) if want_db else None
            for c0 in range(0, K, _KCHUNK):
                kc = min(_KCHUNK, K - c0)
                E._wgrad_raw(hmx.ptr(g) + 0, N, hmx.ptr(x) + 4 * c0, K, in_scale, T, N, kc,
                             hmx.ptr(dw) + 4 * (co + c0), ldw, db if c0 == 0 else None, x.device)
            want_db = False
        if want_db:
            db = colsum(g)
        return dx, dw, db, None, (g if need[4] else None), None


def linear_any(x2, weight, bias=None, r1=None, col_off=0, in_scale=None):
    """nn.Linear on a (T, K) input through hmx_linear_f32: one launch for K <= 256, column chunks of 256 added in
    order for a wider input (the raw multimodal item embeddings, 1024 to 4096 columns).  in_scale (T) multiplies
    the input rows inside the product."""
    if x2.shape[1] <= _KCHUNK and in_scale is None:
        return linear(x2, weight, bias, r1=r1, c1=1.0 if r1 is not None else 0.0, col_off=col_off)
    return _WideLinear.apply(x2, weight, bias, in_scale, r1, int(col_off))
