"""Drop-in modules for hydra3.py on the MI355X kernels of libhmx.so.

Class names, constructor arguments, parameter names / shapes and forward signatures follow the reference
(hydra3.py:158-720, 978-1151) so that a reference state_dict loads with strict=True and main.py / infer.py
call sites keep working; the computation runs in the HIP kernels (enc_ops / ops).  CUDA fp32 only.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import assembly_ops
from . import block_ops
from . import enc_ops as E
from . import feature_ops
from . import hmx
from . import ops
from .enc_ops import pscan, selective_scan   # noqa: F401  (hydra3.py:104, 107: same names, same call shapes)
from .shared_embedding import SharedEmbeddingConfig, SharedEmbeddingModule

D_STATE, DT_RANK, D_CONV = 16, 16, 4
ALIGN_LOSS_ROWS = True          # the loss pads rows whose width is not a multiple of 16 floats (see compute_mixed_contrastive_loss)
FUSED_INPUT_ASSEMBLY = True     # log2feats builds its input with hmx_input_assembly_fwd where the widths allow
VALIDATE_HARD_GROUPS = False   # debug switch: verify on the device that broadcast hard-negative lists repeat per sequence
FUSED_BLOCKS = True      # HydraBlock(First) on the fused block kernels where block_ops.Layout.of(d, heads) exists (d <= 128 after widening)


class RMSNorm(nn.Module):
    """hydra3.py:221-229."""

    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.scale = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return E.rmsnorm(x.reshape(-1, x.shape[-1]), self.scale, self.eps).view(x.shape)


class SSM(nn.Module):
    """hydra3.py:158-218 (parameters; the fused scan kernel consumes them)."""

    def __init__(self, in_features, dt_rank, dim_inner, d_state):
        super().__init__()
        assert dt_rank == DT_RANK and d_state == D_STATE, "kernels are built for dt_rank = d_state = 16"
        self.dt_rank, self.dim_inner, self.d_state = dt_rank, dim_inner, d_state
        self.deltaBC_layer = nn.Linear(in_features, dt_rank + 2 * d_state, bias=False)
        self.dt_proj_layer = nn.Linear(dt_rank, dim_inner, bias=True)
        self.A_log = nn.Parameter(torch.log(torch.arange(1, d_state + 1, dtype=torch.float32)).repeat(dim_inner, 1))
        self.D = nn.Parameter(torch.ones(dim_inner))

    def forward(self, x, pscan=True):
        B, L, ED = x.shape
        xc = x.reshape(B * L, ED)
        dbc = E.heads_linear(xc, [self.deltaBC_layer.weight])
        y = E.selective_scan_gated(xc, None, dbc, self.dt_proj_layer.weight, self.dt_proj_layer.bias, self.A_log,
                                   self.D, L, ED, 1)
        return y.view(B, L, ED)


class MambaBlock(nn.Module):
    """hydra3.py:232-277."""

    def __init__(self, dim, dt_rank, dim_inner, d_state, d_conv=D_CONV):
        super().__init__()
        assert d_conv == D_CONV
        self.dim, self.dim_inner, self.d_conv = dim, dim_inner, d_conv
        self.in_proj = nn.Linear(dim, dim_inner * 2, bias=False)
        self.conv1d = nn.Conv1d(dim_inner, dim_inner, d_conv, groups=dim_inner, padding=d_conv - 1, bias=True)
        self.activation = nn.SiLU()
        self.ssm = SSM(dim_inner, dt_rank, dim_inner, d_state)
        self.out_proj = nn.Linear(dim_inner, dim, bias=False)

    def forward(self, x):
        B, L, dim = x.shape
        return mamba_heads([self], x.reshape(B * L, dim), L).view(B, L, dim)


def mamba_heads(blocks, s, L):
    """All heads' MambaBlocks at once.  s: (T, H*hd) -> (T, H*hd)."""
    ED, H = blocks[0].dim_inner, len(blocks)
    x_in = E.heads_linear(s, [m.in_proj.weight for m in blocks], rows=(0, ED))
    z = E.heads_linear(s, [m.in_proj.weight for m in blocks], rows=(ED, 2 * ED))
    conv_w, conv_b, w_dt, b_dt, a_log, dskip = E.head_params(blocks)
    xc = E.conv_silu(x_in, conv_w, conv_b, L)
    yg = E.proj_scan_gated(xc, z, [m.ssm.deltaBC_layer.weight for m in blocks], w_dt, b_dt, a_log, dskip, L, ED, H)
    return E.heads_linear(yg, [m.out_proj.weight for m in blocks])


class MultiHeadMamba(nn.Module):
    """hydra3.py:280-389."""

    def __init__(self, dim, dt_rank, dim_inner, d_state, args):
        super().__init__()
        self.args = args
        self.head_dim = dim // args.hydra_num_heads
        assert self.head_dim % 2 == 0, "Head dimension must be even for RoPE"       # hydra3.py:315
        self.mamba = nn.ModuleList([MambaBlock(self.head_dim, dt_rank, 4 * self.head_dim, d_state)
                                    for _ in range(args.hydra_num_heads)])
        self.activation = nn.SiLU()
        self.linear_layer1 = nn.Linear(dim, dim)
        self.linear_layer2 = nn.Linear(dim, dim)

    def mix(self, x2, mask1, L):
        """x2 (T, d) -> W2-input p = y1 * y2 / sqrt(H)."""
        x1 = E.linear(x2, self.linear_layer1.weight, self.linear_layer1.bias, in_mask=mask1)
        s, y2 = E.silu_rope(x1, L, self.head_dim)
        y1 = mamba_heads(list(self.mamba), s, L)
        return E.mul_scale(y1, y2, 1.0 / math.sqrt(len(self.mamba)))

    def forward(self, x, x_org, padding_mask):
        B, L, d = x.shape
        m1 = padding_mask.reshape(B * L) if padding_mask is not None else None
        p = self.mix(x.reshape(B * L, d), m1, L)
        out = E.linear(p, self.linear_layer2.weight, self.linear_layer2.bias, r1=x_org.reshape(B * L, d), c1=1.0)
        return out.view(B, L, d)


class _HydraBlockBase(nn.Module):
    first = False
    # (B, L, d) tensor used in place of a random dropout draw while training (the values nn.Dropout would multiply
    # by: 0 or 1 / (1 - p)); None = draw.  For repeating a step on recorded multipliers.
    fixed_drop_multiplier = None

    def __init__(self, dim, dt_rank, dim_inner, d_state, args):
        super().__init__()
        self.multiheadMamba = MultiHeadMamba(dim, dt_rank, dim_inner, d_state, args)
        self.gate = nn.Linear(dim, dim, bias=False)
        self.up = nn.Linear(dim, dim, bias=False)
        self.down = nn.Linear(dim, dim, bias=False)
        self.activation = nn.SiLU()
        self.dropout = nn.Dropout(p=args.hydra_dropout_rate)
        self.norm1 = RMSNorm(dim)
        self.norm2 = RMSNorm(dim)

    def fused_layout(self, device=None):
        """The feature layout in which this block runs on the fused block kernels (block_ops.Layout), or None when it
        takes the per-operation kernels (width not supported, single-bf16 mode, CPU tensors)."""
        mhm = self.multiheadMamba
        if not (FUSED_BLOCKS and hmx.SPLIT_BF16) or (device is not None and torch.device(device).type != "cuda"):
            return None
        return block_ops.Layout.of(mhm.linear_layer1.weight.shape[0], len(mhm.mamba))

    def forward(self, x, padding_mask):
        """x (B, L, d) -> (B, L, d) (hydra3.py:410-423 / 444-457).  A stack of blocks whose layout is widened
        (fused_layout().widened) may hand the widened (B, L, dp) tensor from block to block: an input of that width
        is taken as widened already and the output keeps it."""
        B, L, d = x.shape
        T = B * L
        x2 = x.reshape(T, d)
        m1 = padding_mask.reshape(T).contiguous() if padding_mask is not None else None
        mhm = self.multiheadMamba
        lay = self.fused_layout(x.device)
        if lay is not None:
            # the block as two fused token-parallel kernels around the conv / scan core
            if d == lay.dp:
                return block_ops.hydra_block(self, x2.contiguous(), m1, L, self.first, lay).view(B, L, d)
            out = block_ops.hydra_block(self, lay.widen(x2).contiguous(), m1, L, self.first, lay)
            return lay.narrow(out).view(B, L, d)
        if self.training and self.dropout.p > 0.0 and self.fixed_drop_multiplier is not None:
            xd = E.mul_scale(x2, self.fixed_drop_multiplier.reshape(T, d).contiguous(), 1.0)
        else:
            xd = feature_ops.dropout(x2, self.dropout.p, self.training)
        n = E.rmsnorm(xd, self.norm1.scale, self.norm1.eps)
        p = mhm.mix(n, m1, L)
        # mixer output x_org + W2 p + b2, plus the block's residual: n (first block) or x again (regular block)
        if self.first:
            m = E.linear(p, mhm.linear_layer2.weight, mhm.linear_layer2.bias, r1=x2, c1=1.0, r2=n, c2=1.0)
        else:
            m = E.linear(p, mhm.linear_layer2.weight, mhm.linear_layer2.bias, r1=x2, c1=2.0)
        n2 = E.rmsnorm(m, self.norm2.scale, self.norm2.eps)
        hmid = E.swiglu(E.linear(n2, self.gate.weight), E.linear(n2, self.up.weight))
        out = E.linear(hmid, self.down.weight, r1=n2 if self.first else m, c1=1.0, out_mask=m1)
        return out.view(B, L, d)


class HydraBlockFirst(_HydraBlockBase):
    """hydra3.py:392-423."""
    first = True


class HydraBlock(_HydraBlockBase):
    """hydra3.py:426-457."""
    first = False


class _AddPerSequence(torch.autograd.Function):
    """x (B, L, d) + c * v (B, d) broadcast over the positions (hydra3.py:547).  Backward: x's gradient is the incoming
    one as it is, v's is c times its sum over the positions -- autograd's own rule first scales the whole (B, L, d)
    gradient by c and then sums it."""

    @staticmethod
    def forward(ctx, x, v, c):
        ctx.c = c
        return torch.add(x, v.unsqueeze(1), alpha=c)

    @staticmethod
    def backward(ctx, g):
        dv = g.sum(1).mul_(ctx.c) if ctx.needs_input_grad[1] else None
        return (g if ctx.needs_input_grad[0] else None), dv, None


class _ScaleWindows(torch.autograd.Function):
    """(sequence[:, -sw:], sequence[:, -mw:], sequence) for the three encoders of hydra3.py:528-530.  Through plain
    slicing the backward pass builds two zero-filled (B, L, d) tensors, copies a window into each and adds all three;
    here the two window gradients are added onto the tails of the long encoder's input gradient (which this node
    owns: it is the fresh output of that encoder's backward kernel)."""

    @staticmethod
    def forward(ctx, sequence, sw, mw):
        ctx.windows = (sw, mw)
        return sequence[:, -sw:].contiguous(), sequence[:, -mw:].contiguous(), sequence.view_as(sequence)

    @staticmethod
    def backward(ctx, d_short, d_mid, d_long):
        sw, mw = ctx.windows
        some = d_long if d_long is not None else (d_mid if d_mid is not None else d_short)
        if some is None:
            return None, None, None
        if d_long is None:
            B, d = some.shape[0], some.shape[2]
            d_long = some.new_zeros(B, max(sw, mw), d)          # only reached when the long encoder is cut off
        elif not d_long.is_contiguous():
            d_long = d_long.contiguous()
        if d_short is not None:
            d_long[:, -sw:] += d_short
        if d_mid is not None:
            d_long[:, -mw:] += d_mid
        return d_long, None, None


class MultiScaleUserEncoder(nn.Module):
    """hydra3.py:460-548.  The three HydraBlocks, the attention's projections, the one-query attention, the gate's
    Linear, the gated mix, the RMSNorm and the output projection run on the HIP kernels; the final broadcast add
    over the sequence is PyTorch."""

    def __init__(self, dim, dt_rank, dim_inner, d_state, args):
        super().__init__()
        self.args, self.hidden_dim = args, dim
        self.short_term_encoder = HydraBlock(dim, dt_rank, dim_inner, d_state, args)
        self.mid_term_encoder = HydraBlock(dim, dt_rank, dim_inner, d_state, args)
        self.long_term_encoder = HydraBlock(dim, dt_rank, dim_inner, d_state, args)
        self.cross_attention = nn.MultiheadAttention(dim, args.hydra_num_heads, dropout=args.hydra_dropout_rate,
                                                     batch_first=True)
        self.fusion_gate = nn.Sequential(nn.Linear(dim * 3, dim), nn.Sigmoid())
        self.output_proj = nn.Linear(dim, dim)
        self.norm = RMSNorm(dim)

    def forward(self, sequence, padding_mask=None):
        L = sequence.shape[1]
        sw, mw = min(10, L), min(25, L)
        sm = padding_mask[:, -sw:].contiguous() if padding_mask is not None else None
        mm = padding_mask[:, -mw:].contiguous() if padding_mask is not None else None
        if sequence.requires_grad:
            short_in, mid_in, long_in = _ScaleWindows.apply(sequence, sw, mw)
        else:
            short_in, mid_in, long_in = sequence[:, -sw:].contiguous(), sequence[:, -mw:].contiguous(), sequence
        short = self.short_term_encoder(short_in, sm)
        mid = self.mid_term_encoder(mid_in, mm)
        long_ = self.long_term_encoder(long_in, padding_mask)
        lay = self.long_term_encoder.fused_layout(sequence.device)
        if lay is not None and lay.widened and long_.shape[-1] == lay.dp:     # widened input: the tail runs at d
            short, mid = (lay.narrow(t[:, -1]).unsqueeze(1) for t in (short, mid))
            long_ = lay.narrow(long_.reshape(-1, lay.dp)).view(long_.shape[0], L, lay.d)
        B, _, d = long_.shape
        sf, mf, lf = short[:, -1].contiguous(), mid[:, -1].contiguous(), long_[:, -1].contiguous()
        # nn.MultiheadAttention(query=sf, key=value=long_) spelled out on the HIP kernels: in-projections,
        # one-query attention (dropout on the weights as the module applies it in training), out-projection
        ca = self.cross_attention
        H = ca.num_heads
        W, bias = ca.in_proj_weight, ca.in_proj_bias
        q = E.linear(sf, W, bias, rows=(0, d))
        kv = E.linear(long_.reshape(B * L, d), W, bias, rows=(d, 3 * d))
        drop = None
        if self.training and ca.dropout > 0.0:
            drop = feature_ops.dropout_multiplier((B, H, L), ca.dropout, long_.device)
        att = E.linear(E.attn1q(q, kv, drop, L, H), ca.out_proj.weight, ca.out_proj.bias)
        gate = self.fusion_gate[0]        # Linear(3d -> d) on cat[sf, mf, lf], as three column blocks of the weight
        z = E.linear(sf, gate.weight, gate.bias)
        z = E.linear(mf, gate.weight, None, r1=z, c1=1.0, col_off=d)
        z = E.linear(lf, gate.weight, None, r1=z, c1=1.0, col_off=2 * d)
        mixed = feature_ops.gate_mix(z, att, lf)          # sigmoid(z) * att + (1 - sigmoid(z)) * lf
        fused = E.linear(E.rmsnorm(mixed, self.norm.scale, self.norm.eps), self.output_proj.weight, self.output_proj.bias)
        return _AddPerSequence.apply(long_, fused, 0.5)


class PositionAwareEncoder(nn.Module):
    """hydra3.py:1044-1082: constructed by Hydra (its parameters are part of the state_dict) but not called by the
    model; the forward is the reference's, on the HIP kernels."""

    def __init__(self, hidden_dim, max_len=512):
        super().__init__()
        self.hidden_dim, self.max_len = hidden_dim, max_len
        self.position_weights = nn.Parameter(torch.ones(max_len))
        self.position_attention = nn.MultiheadAttention(hidden_dim, 2, dropout=0.2, batch_first=True)

    def forward(self, sequence, positions=None):
        """sequence (B, L, d) -> sequence + self-attention over the sequence scaled per position by
        softmax(position_weights[:L]) (hydra3.py:1065-1082; `positions` is unused there as well)."""
        B, L, d = sequence.shape
        x2 = sequence.reshape(B * L, d)
        w = torch.softmax(self.position_weights[:L], dim=0)                    # L numbers
        xw = E.mul_scale(x2, w.repeat(B).unsqueeze(1).expand(B * L, d).contiguous(), 1.0)
        mha = self.position_attention
        W, bias = mha.in_proj_weight, mha.in_proj_bias
        q = E.linear(xw, W[:d], bias[:d])
        kv = E.linear(xw, W[d:], bias[d:])
        drop = None
        if self.training and mha.dropout > 0.0:
            drop = feature_ops.dropout_multiplier((B, mha.num_heads, L, L), mha.dropout, sequence.device)
        att = feature_ops.attention(q, kv, drop, B, L, L, mha.num_heads)
        out = E.linear(att, mha.out_proj.weight, mha.out_proj.bias, r1=x2, c1=1.0)
        return out.view(B, L, d)


class AdaptiveHardNegativeSampler(nn.Module):
    """hydra3.py:1024-1041."""

    def __init__(self, initial_ratio=0.3, max_ratio=0.7):
        super().__init__()
        self.initial_ratio, self.max_ratio = initial_ratio, max_ratio
        self.register_buffer('training_step', torch.tensor(0.0))

    def get_current_ratio(self, max_steps=10000):
        progress = min(float(self.training_step) / max_steps, 1.0)
        return self.initial_ratio + (self.max_ratio - self.initial_ratio) * progress

    def update_step(self):
        self.training_step += 1


class HardNegativeMiner:
    """hydra3.py:978-1007 on the fused HIP scoring + top-K: no (B, items) score matrix is formed."""

    def __init__(self, model, item_embeddings, top_k=100, prepared=None):
        self.model, self.item_embeddings, self.top_k = model, item_embeddings, top_k
        self.prepared = prepared

    @torch.no_grad()
    def mine_hard_negatives_similarity(self, user_embeddings, positive_items, exclude_items=None):
        prepared = self.prepared if self.prepared is not None else ops.PreparedTable(self.item_embeddings.detach())
        excl = [t.reshape(t.shape[0], -1) for t in (positive_items, exclude_items) if t is not None]
        excl = torch.cat(excl, dim=1) if excl else None
        ids, _ = ops.score_topk(user_embeddings.detach(), prepared, self.top_k, exclude=excl)
        return ids


def _mine_popular(self, user_embeddings, item_popularity, exclude_items=None):
    """hydra3.py:1009-1021: top_k ids per user drawn with replacement from softmax(item_popularity) (a sampling
    step, torch's multinomial)."""
    B = user_embeddings.shape[0]
    probs = F.softmax(item_popularity.float(), dim=0)
    return torch.multinomial(probs, self.top_k * B, replacement=True).view(B, self.top_k)


HardNegativeMiner.mine_hard_negatives_popular = _mine_popular


def _sample_inbatch_indices_no_self(M, rows, T, device):
    """hydra3.py:817-830."""
    if M - 1 >= 2 ** 31:
        r = torch.randint(0, M - 1, (rows.size(0), T), device=device)
        return r + (r >= rows.unsqueeze(1)).long()
    # drawn as 32-bit integers (the generator yields the same values for either width), widened by the add itself
    r = torch.randint(0, M - 1, (rows.size(0), T), device=device, dtype=torch.int32)
    out = torch.empty(rows.size(0), T, dtype=torch.int64, device=device)
    return torch.add(r, r >= rows.unsqueeze(1).to(torch.int32), out=out)


class _TakeRows(torch.autograd.Function):
    """x2[rows] for distinct, ascending `rows`: the gradient is a row copy into zeros (the general index backward
    sorts its indices to cope with repeats, which a mask cannot produce)."""

    @staticmethod
    def forward(ctx, x2, rows):
        ctx.save_for_backward(rows)
        ctx.n = x2.shape[0]
        return x2.index_select(0, rows)

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        dx = g.new_zeros(ctx.n, g.shape[1])
        dx.index_copy_(0, rows, g)
        return dx, None


class LossRows:
    """The positions of a (B, L) loss mask, found on the HOST: the row numbers (row-major), the sequence of each row
    and the first row of each sequence, uploaded together.  Boolean indexing / nonzero() on the device would make
    the host wait for the whole encoder before it can size the loss tensors; here a mask that lives on the device is
    copied to pinned host memory when this object is made (Hydra.forward makes it before the encoder runs) and only
    that small copy is waited for when the rows are needed.  main.py keeps next_token_type on the host, in which case
    nothing is copied at all."""

    def __init__(self, mask, device):
        self.device = torch.device(device)
        self.shape = tuple(mask.shape)
        self._event = None
        self._ready = None
        mask = mask.bool()
        if mask.is_cuda:
            self._host = torch.empty(mask.shape, dtype=torch.bool, pin_memory=True)
            self._host.copy_(mask, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host = mask

    def get(self):
        """-> (rows (M,), seq_of_row (M,), starts (B + 1,)) int64 on the device, M."""
        if self._ready is None:
            if self._event is not None:
                self._event.synchronize()
            import numpy as np
            m = self._host.numpy()
            B, L = m.shape
            rows = np.flatnonzero(m)
            starts = np.zeros(B + 1, dtype=np.int64)
            np.cumsum(m.sum(1), out=starts[1:])
            M = int(rows.shape[0])
            packed = torch.from_numpy(np.concatenate([rows, rows // L, starts]))
            if self.device.type == "cuda":
                packed = packed.pin_memory().to(self.device, non_blocking=True)
            self._ready = (packed[:M], packed[M:2 * M], packed[2 * M:], M)
        return self._ready


def compute_mixed_contrastive_loss(log_feats, pos_embs, loss_mask, item_table=None, hard_negatives=None, T=512,
                                   hard_ratio=0.3, temperature=0.05, normalize=True, q_chunk=64, device=None,
                                   predrawn=None, loss_rows=None):
    """hydra3.py:845-974 (InfoNCE over in-batch + hard negatives).  The negatives are drawn here exactly as the
    reference draws them (index tensors only); the loss itself -- hard-negative row gather, normalisation, the
    (M, 513) logits, log-softmax and the whole backward incl. the scatter into the item-table gradient -- is one
    fused HIP op, so no (M, 513, d) key tensor exists.  `q_chunk` only bounded the reference's memory use and has
    no effect on the value; it is accepted and ignored.
    predrawn = (hard_ids (M, n) item-table rows or None, inb_idx (M, n') rows of the masked positives or None)
    replaces the sampling: the loss is then a deterministic function of its inputs (used to compare a whole train
    step with draws recorded from the reference, whose CPU generator a GPU cannot replay).
    loss_rows: a LossRows made earlier from the same mask (Hydra.forward does, so that its host copy is long done)."""
    device = log_feats.device
    if loss_rows is None:
        loss_rows = LossRows(loss_mask, device)
    # the masked positions in row-major order, found once on the host
    rows, seq_of_row, starts, M = loss_rows.get()
    d = log_feats.shape[-1]
    if M == 0:
        return torch.tensor(0.0, device=device, requires_grad=True)
    if M == log_feats.numel() // d:
        # every position carries a loss row (distinct, ascending row numbers: rows == arange): the selection of
        # hydra3.py:866-868 is the identity, so no copy forward and no zero-filled scatter backward
        q, k = log_feats.reshape(-1, d), pos_embs.reshape(-1, d)
    else:
        q, k = _TakeRows.apply(log_feats.reshape(-1, d), rows), _TakeRows.apply(pos_embs.reshape(-1, d), rows)
    if predrawn is not None:
        hard_ids, inb = predrawn
        return ops.fused_nce_rows(q, k, item_table, hard_ids, inb, temperature, normalize).mean()
    num_hard = int(T * hard_ratio)
    num_inb = T - num_hard
    hard_ids = None
    hard_groups = None
    n_hard_slots = None
    per_sequence = False
    if hard_negatives is not None and num_hard > 0:                      # hydra3.py:905-919
        per_sequence = hard_negatives.dim() == 3 and hard_negatives.stride(1) == 0
        if per_sequence:                          # one list per sequence, broadcast over its positions
            ch = hard_negatives[:, 0].index_select(0, seq_of_row)
        else:
            ch = hard_negatives.reshape(-1, hard_negatives.shape[-1]).index_select(0, rows)
        K = ch.shape[1]
        if K >= num_hard:
            ch = torch.gather(ch, 1, torch.randint(0, K, (M, num_hard), device=device))
        elif per_sequence and K <= 64 and q.is_cuda:
            # Fewer ids than slots: the reference tiles them over the slots (hydra3.py:911-914).  The rows of a
            # sequence are consecutive in the masked order and share their list, so the tiled copy is not built: the
            # loss takes the (M, K) lists, the number of slots they stand for and where each sequence's rows start.
            # Contract of this path: hard_negatives is ONE (B, K) list per sequence broadcast over the positions
            # (stride(1) == 0 says exactly that for the tensor as it arrives); ids that differ between the positions
            # of a sequence must come as a dense (B, L, K) tensor.  VALIDATE_HARD_GROUPS re-checks it on the device.
            hard_groups = (starts, K)
            n_hard_slots = num_hard
            if VALIDATE_HARD_GROUPS:
                first = ch.index_select(0, starts[:-1].clamp(max=M - 1).index_select(0, seq_of_row))
                if not torch.equal(ch, first):
                    raise hmx.HmxError("hard negatives are not one list per sequence: pass a dense (B, L, K) tensor "
                                       "instead of a broadcast view")
        else:
            ch = ch.repeat(1, (num_hard + K - 1) // K)[:, :num_hard]
        hard_ids = ch
    inb = None
    skip_self = False
    if num_inb > 0:                                                      # hydra3.py:921-950
        if M - 1 >= num_inb and M - 1 < 2 ** 31 and q.is_cuda:
            # hydra3.py:824-829: r uniform in [0, M - 1), index r + (r >= own row).  Only r is drawn (32-bit: the
            # generator yields the same values for either width); the loss kernels apply the shift as they read it
            inb = torch.randint(0, M - 1, (M, num_inb), device=device, dtype=torch.int32)
            skip_self = True
        elif M - 1 >= num_inb:
            inb = _sample_inbatch_indices_no_self(M, torch.arange(M, device=device), num_inb, device)
        else:
            ar = torch.arange(M, device=device)
            base = ar.unsqueeze(0).repeat(M, 1)
            inb = base[base != ar.unsqueeze(1)].view(M, M - 1)
            short = num_inb - (M - 1)
            if short > 0:
                if item_table is not None:       # random table rows fill the batch's shortfall: more table keys
                    rid = torch.randint(1, item_table.shape[0], (M, short), device=device)
                    if hard_groups is not None:  # the extra random rows break the per-sequence repetition
                        hard_ids = hard_ids.repeat(1, (num_hard + hard_ids.shape[1] - 1) // hard_ids.shape[1])[:, :num_hard]
                        hard_groups, n_hard_slots = None, None
                    hard_ids = rid if hard_ids is None else torch.cat([hard_ids, rid], dim=1)
                else:
                    inb = torch.cat([inb, inb[:, -1:].expand(-1, short)], dim=1)
    table = item_table
    if ALIGN_LOSS_ROWS and q.is_cuda and d % 16 and hard_groups is not None and per_sequence:
        # Rows of a width that is not a multiple of 64 bytes (the reference's default d = 60: 240-byte rows) straddle
        # cache lines, and the loss is two walks over 36 M such rows.  Same loss on rows widened with zero columns: q
        # and k are padded, and the hard negatives -- one list of K ids per sequence -- are gathered once into a small
        # padded table of B * K rows that the loss indexes instead (row b * K + j = item row of sequence b's j-th id);
        # its gradient returns through the gather's scatter-add.  Zero columns change neither norms nor dot products.
        pad = (0, (d + 15) // 16 * 16 - d)
        q, k = F.pad(q, pad), F.pad(k, pad)
        lists = hard_negatives[:, 0]                                     # (B, K) ids
        table = F.pad(ops.gather_rows(item_table, lists.reshape(-1)), pad)
        K = lists.shape[1]
        hard_ids = (seq_of_row * K).unsqueeze(1) + torch.arange(K, device=device).unsqueeze(0)
    return ops.fused_nce_rows(q, k, table, hard_ids, inb, temperature, normalize, hard_groups=hard_groups,
                              n_hard=n_hard_slots, inb_skip_self=skip_self).mean()


class EnhancedTrainingLoss(nn.Module):
    """hydra3.py:1085-1151."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.ranking_loss_weight = getattr(args, 'ranking_loss_weight', 0.0)
        self.contrastive_loss_weight = getattr(args, 'contrastive_loss_weight', 1.0)
        self.diversity_loss_weight = getattr(args, 'diversity_loss_weight', 0.0)
        self.q_chunk = getattr(args, 'loss_q_chunk', 256)          # hydra3.py:1132
        # The reference evaluates the ranking and diversity terms whatever their weights (hydra3.py:1139-1149), i.e.
        # with the default weights 0.0 it computes them, multiplies by zero and back-propagates zeros.  False: terms
        # with weight 0.0 are skipped (same loss, same gradients); True: they are evaluated like the reference does.
        self.evaluate_all_terms = bool(getattr(args, 'evaluate_zero_weight_terms', False))
        self.adaptive_hard_sampler = AdaptiveHardNegativeSampler()

    def forward(self, user_embs, pos_embs, neg_embs, hard_negatives, loss_mask, item_table, predrawn=None,
                loss_rows=None):
        total = self.contrastive_loss_weight * compute_mixed_contrastive_loss(
            user_embs, pos_embs, loss_mask, item_table, hard_negatives, hard_ratio=0.3, T=512, temperature=0.05,
            q_chunk=self.q_chunk, predrawn=predrawn, loss_rows=loss_rows)
        self.adaptive_hard_sampler.update_step()
        if neg_embs is not None and (self.ranking_loss_weight != 0.0 or self.evaluate_all_terms):
            total = total + self.ranking_loss_weight * self.compute_ranking_loss(
                user_embs[:, -1], pos_embs[:, -1], neg_embs[:, -1], loss_mask[:, -1].to(user_embs.device))
        if self.diversity_loss_weight != 0.0 or self.evaluate_all_terms:
            total = total + self.diversity_loss_weight * self.compute_diversity_loss(user_embs[:, -1], item_table)
        return total

    def compute_ranking_loss(self, user_embs, pos_embs, neg_embs, loss_mask):
        """hydra3.py:1098-1104: BPR-style loss on the last position ((B, d)-sized: plain torch ops)."""
        rl = -F.logsigmoid((user_embs * pos_embs).sum(-1) - (user_embs * neg_embs).sum(-1))
        return (rl * loss_mask).sum() / loss_mask.sum()

    def compute_diversity_loss(self, user_embs, item_embs=None):
        """hydra3.py:1106-1116: mean absolute off-diagonal cosine similarity of the batch's user embeddings."""
        un = F.normalize(user_embs, p=2, dim=1)
        sim = un @ un.t()
        return (sim * (1 - torch.eye(sim.shape[0], device=sim.device))).abs().mean()


class Hydra(nn.Module):
    """hydra3.py:553-720."""

    def __init__(self, user_num, item_num, feat_statistics, feat_types, args, shared_embeddings=None):
        super().__init__()
        self.user_num, self.item_num, self.dev = user_num, item_num, args.device
        self.norm_first, self.maxlen, self.args = args.norm_first, args.maxlen, args
        if shared_embeddings is not None:
            self.shared_embeddings = shared_embeddings
        else:
            SharedEmbeddingConfig(args)
            self.shared_embeddings = SharedEmbeddingModule(user_num, item_num, feat_statistics, feat_types, args)
        d = args.hydra_hidden_units
        if args.shared_hidden_units != d:
            self.embedding_projection = nn.Linear(args.shared_hidden_units, d)
        else:
            self.embedding_projection = nn.Identity()
        self.hydra_projection = nn.Linear(args.shared_hidden_units * 2, d)
        if not hasattr(self.shared_embeddings, 'hydra_projection'):
            self.shared_embeddings.hydra_projection = self.hydra_projection
        self.time_span_emb = nn.Embedding(getattr(args, 'time_span', 512), d, padding_idx=0)   # unused (hydra3.py:580)
        self.hydra = nn.ModuleList()
        self.multi_scale_encoder = MultiScaleUserEncoder(d, DT_RANK, 4 * d, D_STATE, args)
        self.hidden_units = d
        for _ in range(args.hydra_num_blocks):
            self.hydra.append(HydraBlock(d, DT_RANK, 4 * d, D_STATE, args))
        self.hydra_first = HydraBlockFirst(d, DT_RANK, 4 * d, D_STATE, args)
        self.position_aware_encoder = PositionAwareEncoder(d, max_len=args.maxlen + 1)
        self.enchanced_loss = EnhancedTrainingLoss(args)
        self._prepared = None
        # direct-gradient mode (enc_ops.set_direct_param_grad): the small parameters' .grad tensors are views of one
        # flat buffer; the two big tables keep their own dense gradients
        se = self.shared_embeddings
        E.register_grad_arena(self, exclude=(se.item_emb.weight, se.user_emb.weight))

    # ---- input assembly (hydra3.py:621-669) --------------------------------------------------
    def _project_small(self, rows):
        if isinstance(self.embedding_projection, nn.Identity):
            return rows
        return E.linear(rows, self.embedding_projection.weight, self.embedding_projection.bias)

    def _project_ids(self, base, feat=None):
        """hydra_projection applied to cat[base, feat] without building the concatenation: the two halves of the
        weight's columns are applied to the two inputs (feat None = the zero block of a batch without features)."""
        y = E.linear(base, self.hydra_projection.weight, self.hydra_projection.bias)
        if feat is not None:
            y = E.linear(feat.reshape(-1, feat.shape[-1]), self.hydra_projection.weight, None, r1=y, c1=1.0,
                         col_off=base.shape[-1])
        return y

    def _small_rows(self, table):
        """A small table as the fused assembly reads it: after embedding_projection.  The padding row keeps its place
        but is cut out of the graph, as nn.Embedding(padding_idx=0) keeps its gradient at zero."""
        if isinstance(self.embedding_projection, nn.Identity):
            return table.weight
        w = table.weight
        w = torch.cat([w[:1].detach(), w[1:]], dim=0)
        return E.linear(w, self.embedding_projection.weight, self.embedding_projection.bias)

    def assemble_inputs(self, log_seqs, mask, seq_timestamp, action_type, dwell_bins, seq_feature=None):
        se = self.shared_embeddings
        B, L = log_seqs.shape
        T = B * L
        d, ds = self.args.hydra_hidden_units, se.item_emb.embedding_dim
        if (FUSED_INPUT_ASSEMBLY and log_seqs.is_cuda and assembly_ops.supported(d, ds)
                and L + 1 <= se.pos_emb.num_embeddings):
            feats = seq_feature if seq_feature is not None else []
            pre = None
            if se.has_features and not se._is_empty(feats):
                feat = se.get_feature_embeddings(feats, True, 'early')
                pre = E.linear(feat.reshape(-1, feat.shape[-1]), self.hydra_projection.weight, None, col_off=ds)
            identity = isinstance(self.embedding_projection, nn.Identity)
            seqs, pm, _, _ = assembly_ops.input_assembly(
                se.item_emb.weight, se.user_emb.weight, se.pos_emb.weight, self.hydra_projection.weight,
                self.hydra_projection.bias, pre, self._small_rows(se.time_span_emb), self._small_rows(se.action_emb),
                self._small_rows(se.dwell_time_emb), log_seqs, mask, seq_timestamp, action_type, dwell_bins,
                getattr(self.args, 'time_span', 512), d ** 0.5, small_padding_idx=0 if identity else -1)
            return seqs, pm
        pm, pos, tb = E.seq_index(log_seqs, seq_timestamp, getattr(self.args, 'time_span', 512))
        e, feat = se.embedding_parts(log_seqs.reshape(T), seq_feature if seq_feature is not None else [], mask.reshape(T),
                                     include_user=True, fusion_style='early', pos_seq=pos)
        proj = self._project_ids(e, feat)
        t_rows = self._project_small(ops.gather_rows(se.time_span_emb.weight, tb, 0))
        a_rows = self._project_small(ops.gather_rows(se.action_emb.weight, action_type.reshape(T), 0))
        d_rows = self._project_small(ops.gather_rows(se.dwell_time_emb.weight, dwell_bins.reshape(T), 0))
        seqs = E.combine_ad([proj, t_rows, a_rows, d_rows],
                            [self.args.hydra_hidden_units ** 0.5, 1.0, 1.0, 1.0], rowmask=pm)
        return seqs.view(B, L, -1), pm.view(B, L, 1)

    def log2feats(self, log_seqs, seq_feature, mask, seq_timestamp, action_type, dwell_bins):
        """hydra3.py:618-681."""
        if log_seqs.shape[1] > self.maxlen + 1:                          # hydra3.py:623-626
            cut = -self.maxlen - 1
            log_seqs, mask, seq_timestamp = log_seqs[:, cut:], mask[:, cut:], seq_timestamp[:, cut:]
            action_type, dwell_bins = action_type[:, cut:], dwell_bins[:, cut:]
            if isinstance(seq_feature, dict):
                seq_feature = {k: v[:, cut:] for k, v in seq_feature.items()}
            else:
                seq_feature = [f[cut:] for f in seq_feature]
        seqs, pm = self.assemble_inputs(log_seqs, mask, seq_timestamp, action_type, dwell_bins, seq_feature)
        B, L, d = seqs.shape
        lay = self.hydra_first.fused_layout(seqs.device)
        if lay is not None and lay.widened:
            # the whole block stack runs in the widened layout: widened once here, narrowed once in the fusion tail
            seqs = lay.widen(seqs.reshape(B * L, d)).view(B, L, lay.dp)
        # dropout multipliers of all blocks (full-length ones plus the 10- and 25-token windows) from one draw
        n_drop = (lay.dp if lay is not None else d) * B * (L * (2 + len(self.hydra)) + min(10, L) + min(25, L))
        with block_ops.drop_pool(n_drop, self.args.hydra_dropout_rate, seqs.device,
                                 enabled=self.training and lay is not None):
            seqs = self.hydra_first(seqs, padding_mask=pm)
            for blk in self.hydra:
                seqs = blk(x=seqs, padding_mask=pm)
            seqs = self.multi_scale_encoder(seqs, pm)
        return seqs * pm

    def get_user_embedding(self, seq, seq_feat, token_type, seq_timestamp, action_type, dwell_bins):
        """hydra3.py:808-815."""
        return self.log2feats(seq, seq_feat, token_type, seq_timestamp, action_type, dwell_bins)[:, -1, :]

    def predict(self, log_seqs, seq_feature, mask, seq_timestamp, action_type, dwell_bins):
        """hydra3.py:722-728 (with the dwell_bins argument the reference's call sites pass)."""
        return self.get_user_embedding(log_seqs, seq_feature, mask, seq_timestamp, action_type, dwell_bins)

    def rerank_candidates(self, user_item, seq_feature, mask, ts, action_type, candidate_ids, dwell_bins=None):
        """hydra3.py:772-788: scores (B, C) of per-user candidate lists (B, C) against the user embedding.  The
        reference's call leaves out dwell_bins (its log2feats needs it); None stands for all-zero bins."""
        if dwell_bins is None:
            dwell_bins = torch.zeros_like(user_item)
        user = self.get_user_embedding(user_item, seq_feature, mask, ts, action_type, dwell_bins)
        B, C = candidate_ids.shape
        rows = self._project_small(self.shared_embeddings.item_emb(candidate_ids.reshape(B * C)))
        return torch.bmm(user.unsqueeze(1), rows.view(B, C, -1).transpose(1, 2)).squeeze(1)

    def save_item_emb(self, item_ids, retrieval_ids, feat_dict, save_path, batch_size=1024):
        """hydra3.py:730-761 (embedding.fbin + id.u64bin of the candidate items).  The reference's body calls a
        `feat2emb` that its Hydra does not have; the files it describes are what the shared module's
        save_item_embeddings (embedding.py:654-684) writes, so the call goes there."""
        return self.shared_embeddings.save_item_embeddings(item_ids, retrieval_ids, feat_dict, save_path, batch_size)

    def prepared_item_table(self, refresh=True):
        """bf16 scoring copy of the item table (rebuilt after every optimizer step while training)."""
        w = self.shared_embeddings.item_emb.weight.detach()
        if self._prepared is None or self._prepared.table.data_ptr() != w.data_ptr():
            self._prepared = ops.PreparedTable(w)
        elif refresh and self._prepared.synced_version != w._version:
            # (an optimizer step of optim.HmxAdamW rewrites the copy from the rows it has just updated and records
            # the table's version; anything else that changed the table since then bumped the version)
            self._prepared.refresh()
        return self._prepared

    def forward(self, user_item, pos_seqs, neg_seqs, mask, next_mask, next_action_type, seq_feature, pos_feature,
                neg_feature, timestamp, action_type, dwell_bins, predrawn=None):
        """hydra3.py:683-720.  The negative-item branch (gather + projection of neg_seqs, hydra3.py:694-696, 711) and
        the diversity term (:1146-1149) enter the reference's loss with weight 0.0 by default; they are evaluated
        here only when their weights are non-zero (the loss value and every gradient are the same either way) or when
        enchanced_loss.evaluate_all_terms asks for the reference's exact sequence of operations.
        predrawn: see compute_mixed_contrastive_loss.  The mined hard-negative ids of the call are kept in
        self.last_mined_ids."""
        E.discard_stale_wgrads()
        # the loss positions are found on the host; a mask that lives on the device starts its way there now
        loss_mask = next_mask == 1                # main.py leaves next_mask on the host (hydra3.py:688)
        loss_rows = LossRows(loss_mask, user_item.device)
        log_feats = self.log2feats(user_item, seq_feature, mask, timestamp, action_type, dwell_bins)
        se = self.shared_embeddings
        item_table = se.item_emb.weight
        B, L = pos_seqs.shape
        pos_embs = self._project_ids(*se.embedding_parts(pos_seqs.reshape(B * L), pos_feature, None, False,
                                                         'early')).view(B, L, -1)
        neg_embs = None
        if self.enchanced_loss.ranking_loss_weight != 0.0 or self.enchanced_loss.evaluate_all_terms:
            neg_embs = self._project_ids(*se.embedding_parts(neg_seqs.reshape(B * L), neg_feature, None, False,
                                                             'early')).view(B, L, -1)
        loss_rows.get()      # host-side work, done while the device still has the encoder queued
        miner = HardNegativeMiner(self, item_table, top_k=50, prepared=self.prepared_item_table())
        hard = miner.mine_hard_negatives_similarity(log_feats[:, -1, :], positive_items=pos_seqs,
                                                    exclude_items=user_item)
        self.last_mined_ids = hard
        hard = hard.unsqueeze(1).expand(-1, log_feats.shape[1], -1)
        return self.enchanced_loss(user_embs=log_feats, pos_embs=pos_embs, neg_embs=neg_embs, hard_negatives=hard,
                                   loss_mask=loss_mask, item_table=item_table, predrawn=predrawn, loss_rows=loss_rows)


IDS_ONLY_FEAT_TYPES = {k: [] for k in ('user_sparse', 'item_sparse', 'user_array', 'item_array',
                                        'user_continual', 'item_continual', 'item_emb')}


def make_args(d=64, L=100, heads=2, blocks=2, dropout=0.2, device='cuda', **kw):
    from types import SimpleNamespace
    a = SimpleNamespace(device=device, hydra_hidden_units=d, hstu_hidden_units=d, shared_hidden_units=d,
                        hydra_num_heads=heads, hydra_num_blocks=blocks, hydra_dropout_rate=dropout, maxlen=L - 1,
                        norm_first=False, num_actions=5, time_span=512)
    a.__dict__.update(kw)
    return a


def build_model(n_users, n_items, d=64, L=100, heads=2, blocks=2, dropout=0.2, device='cuda', seed=0, **kw):
    """Ids-only Hydra initialised as main.py:603-607,619-622 (xavier_normal_ on >= 2-D parameters, zero padding rows)."""
    args = make_args(d, L, heads, blocks, dropout, device, **kw)
    torch.manual_seed(seed)
    se = SharedEmbeddingModule(n_users, n_items, {}, IDS_ONLY_FEAT_TYPES, args, None)
    model = Hydra(n_users, n_items, {}, IDS_ONLY_FEAT_TYPES, args, se)
    for p in model.parameters():
        if p.dim() >= 2:
            nn.init.xavier_normal_(p.data)
    for t in (se.pos_emb, se.item_emb, se.user_emb, se.action_emb):
        t.weight.data[0].zero_()
    return model.to(device)
