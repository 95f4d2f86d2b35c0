"""Input assembly of the sequence encoder as one kernel each way (hmx_input_assembly_fwd / _bwd of include/hmx.h;
hydra3.py:621-669 with the id + position half of embedding.py:613-652).  No CPU fallback."""
import ctypes

import torch

from . import block_ops
from . import hmx
from . import ops
from .ops import _f32c, _i64c, _need_cuda


def supported(d, ds):
    return bool(hmx.lib().hmx_input_assembly_supported(d, ds))


def _fill(args, **tensors):
    for k, v in tensors.items():
        setattr(args, k, None if v is None else v.data_ptr())
    return args


class _InputAssembly(torch.autograd.Function):
    """out (T, d), padding mask (T), position ids (T), time bins (T).  `small_padding_idx`: the row of the three small
    tables that receives no gradient (0 when they are the embedding tables themselves, -1 when they are projected
    copies whose padding row the caller already detached)."""

    @staticmethod
    def forward(ctx, item_w, user_w, pos_w, W, b, pre, time_rows, action_rows, dwell_rows, ids, token_type, ts, action,
                dwell, L, time_span, scale, small_padding_idx):
        _need_cuda(item_w, user_w, pos_w, W, b, time_rows, action_rows, dwell_rows, ids, token_type, ts, action, dwell)
        iw, uw, pw, Wc, bc = _f32c(item_w), _f32c(user_w), _f32c(pos_w), _f32c(W), _f32c(b)
        tr, ar, dr = _f32c(time_rows), _f32c(action_rows), _f32c(dwell_rows)
        prec = _f32c(pre) if pre is not None else None
        idc, ttc, tsc, ac, dc = (_i64c(x).reshape(-1) for x in (ids, token_type, ts, action, dwell))
        T, d, ds = idc.numel(), Wc.shape[0], iw.shape[1]
        dev = iw.device
        train = any(ctx.needs_input_grad[:9])
        out = torch.empty(T, d, dtype=torch.float32, device=dev)
        pm = torch.empty(T, dtype=torch.float32, device=dev)
        pos = torch.empty(T, dtype=torch.int64, device=dev)
        tb = torch.empty(T, dtype=torch.int64, device=dev)
        e = torch.empty(T, ds, dtype=torch.float32, device=dev) if train else None
        a = hmx.InputAssemblyArgs()
        a.T, a.L, a.ds, a.d, a.time_span = T, L, ds, d, time_span
        a.item_rows, a.user_rows, a.pos_rows = iw.shape[0], uw.shape[0], pw.shape[0]
        a.ldw, a.scale = Wc.shape[1], scale
        a.time_n, a.action_n, a.dwell_n = tr.shape[0], ar.shape[0], dr.shape[0]
        _fill(a, ids=idc, token_type=ttc, timestamps=tsc, action=ac, dwell=dc, item_table=iw, user_table=uw, pos_table=pw,
              W=Wc, bias=bc, pre=prec, time_rows=tr, action_rows=ar, dwell_rows=dr, out=out, mask=pm, pos_idx=pos,
              time_bin=tb, e=e)
        hmx.check(hmx.lib().hmx_input_assembly_fwd(ctypes.byref(a), hmx.current_stream()), "hmx_input_assembly_fwd")
        ctx.save_for_backward(idc, ttc, ac, dc, pos, tb, pm, e, Wc)
        ctx.params = (item_w, user_w, pos_w, W, b, time_rows, action_rows, dwell_rows)
        ctx.cfg = (L, time_span, scale, small_padding_idx, pre is not None)
        ctx.mark_non_differentiable(pm, pos, tb)
        ctx.set_materialize_grads(False)      # no zero tensors built for the three index outputs in every backward pass
        return out, pm, pos, tb

    @staticmethod
    def backward(ctx, dout, _pm, _pos, _tb):
        idc, ttc, ac, dc, pos, tb, pm, e, Wc = ctx.saved_tensors
        item_w, user_w, pos_w, W, b, time_rows, action_rows, dwell_rows = ctx.params
        L, time_span, scale, small_pad, has_pre = ctx.cfg
        T, d, ds = idc.numel(), Wc.shape[0], item_w.shape[1]
        if dout is None:
            return (None,) * 18
        dev = dout.device
        dout = _f32c(dout).reshape(T, d)
        g, gs = torch.empty_like(dout), torch.empty_like(dout)
        de = torch.empty(T, ds, dtype=torch.float32, device=dev)
        a = hmx.InputAssemblyArgs()
        a.T, a.L, a.ds, a.d, a.time_span = T, L, ds, d, time_span
        a.ldw, a.scale = Wc.shape[1], scale
        _fill(a, W=Wc, mask=pm, dout=dout, g=g, gs=gs, de=de)
        hmx.check(hmx.lib().hmx_input_assembly_bwd(ctypes.byref(a), hmx.current_stream()), "hmx_input_assembly_bwd")
        need = ctx.needs_input_grad
        grads = [None] * 18

        def table_grad(slot, param, rows, idx, padding_idx, token_type=None, want_type=0):
            if not need[slot]:
                return
            acc = ops._into_param_grad(param)
            r = ops.scatter_add_rows(rows, idx, param.shape[0], padding_idx, token_type, want_type, out=acc)
            grads[slot] = None if acc is not None else r

        table_grad(0, item_w, de, idc, 0, ttc, 1)         # embedding.py:436-450 routes by token type
        table_grad(1, user_w, de, idc, 0, ttc, 2)
        table_grad(2, pos_w, de, pos, 0)
        # hydra_projection: dW[:, :ds] += gs^T e, db += column sums of gs (batched / direct like every other projection)
        wg = block_ops._weight_grads([(0, gs, 0, d, e, 0, ds, None, T, d, ds, 0, Wc.shape[1], 1, 0)], [W, b], dev)
        grads[3], grads[4] = wg[0], wg[1]
        if has_pre and need[5]:
            grads[5] = gs
        table_grad(6, time_rows, g, tb, small_pad)
        table_grad(7, action_rows, g, ac, small_pad)
        table_grad(8, dwell_rows, g, dc, small_pad)
        return tuple(grads)


def input_assembly(item_w, user_w, pos_w, W, b, pre, time_rows, action_rows, dwell_rows, ids, token_type, ts, action, dwell,
                   time_span, scale, small_padding_idx=0):
    """ids / token_type / ts / action / dwell: (B, L).  Returns (seqs (B, L, d), pm (B, L, 1), pos (T,), time_bin (T,))."""
    B, L = ids.shape
    out, pm, pos, tb = _InputAssembly.apply(item_w, user_w, pos_w, W, b, pre, time_rows, action_rows, dwell_rows, ids,
                                            token_type, ts, action, dwell, L, time_span, scale, small_padding_idx)
    return out.view(B, L, -1), pm.view(B, L, 1), pos, tb
