"""AdamW with the gradient-norm clip folded in, on the HIP kernels (hmx_grad_norm_f32, hmx_adamw_step_f32).

Replaces, in the train loop of main.py:640,709-713,

    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    optimizer.step()                                  # torch.optim.AdamW(lr, betas=(0.9, 0.98), weight_decay=...)

by `optimizer.step()` of `HmxAdamW(model.parameters(), ..., max_grad_norm=1.0)`: one pass over the gradients for the
norm, one pass over parameters / gradients / moments for the update, the clip factor applied inside the update and
never leaving the device.  It is a torch.optim.Optimizer (param_groups, state_dict with step / exp_avg / exp_avg_sq per
parameter, LR schedulers), so the rest of the loop is unchanged.

Embedding tables whose dense gradient is kept by ops.TableGrad (fused table gradients: ops.set_fused_table_grad) take
a row-wise form of the same update: only the rows the step's scatter-adds flagged have their gradient read, those rows
are set back to zero and un-flagged once used -- so the buffer is the next step's zero gradient and `zero_grad` costs
nothing for it; after step() such a table's .grad therefore reads all zero -- and the table's bf16 scoring copy
(ops.PreparedTable), if one exists, is rewritten from the updated rows in the same pass.  Values are those of the
dense update, bit for bit.  `sparse_table_grads=False` switches this off.
"""
import ctypes

import numpy as np
import torch

from . import hmx
from . import ops

_DESC = np.dtype([("p", np.uint64), ("g", np.uint64), ("m", np.uint64), ("v", np.uint64), ("n", np.int64)])


class HmxAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 keep_clipped_grads=False, sparse_table_grads=True):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("HmxAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.keep_clipped_grads = bool(keep_clipped_grads)     # also write the clipped gradients back into .grad
        self.sparse_table_grads = bool(sparse_table_grads)
        self.last_grad_norm = None                             # device tensor (norm, clip factor) of the last step
        self._tables = {}                                      # per group index: cached layout for the active set

    # moments live in two flat buffers per group, `exp_avg` / `exp_avg_sq` of a parameter are views into them
    def _init_state(self, group):
        ps = [p for p in group["params"] if "exp_avg" not in self.state[p]]
        if not ps:
            return
        for p in ps:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise hmx.HmxError("HmxAdamW needs contiguous fp32 parameters on the GPU")
        total = sum((p.numel() + 63) // 64 * 64 for p in ps)
        dev = ps[0].device
        flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        at = 0
        for p in ps:
            st = self.state[p]
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = flat_m[at:at + p.numel()].view(p.shape)
            st["exp_avg_sq"] = flat_v[at:at + p.numel()].view(p.shape)
            at += (p.numel() + 63) // 64 * 64

    def _table(self, gi, active):
        """Device descriptor table + chunk offsets for the parameters of group gi that have a gradient this step."""
        chunk = int(hmx.lib().hmx_adamw_chunk_elems())
        key = tuple(id(p) for p in active)
        cached = self._tables.get(gi)
        if cached is None or cached["key"] != key:
            if len(active) > 1023:
                raise hmx.HmxError("HmxAdamW: more than 1023 tensors in one group")
            counts = [(p.numel() + chunk - 1) // chunk for p in active]
            starts = np.zeros(len(active) + 1, dtype=np.int32)
            starts[1:] = np.cumsum(counts)
            dev = active[0].device
            cached = {"key": key, "chunks": int(starts[-1]),
                      "starts": torch.from_numpy(starts).to(dev),
                      "desc": torch.empty(len(active) * _DESC.itemsize, dtype=torch.uint8, device=dev),
                      "host": np.zeros(len(active), dtype=_DESC)}
            self._tables[gi] = cached
        # parameters and moments stay where they are; gradients may move between steps (autograd hands over fresh
        # tensors): only the entries whose pointer changed are checked and rewritten, and the table is uploaded
        # again only if something did
        host = cached["host"]
        gptr = [p.grad.data_ptr() for p in active]
        pptr = [p.data_ptr() for p in active]
        if gptr != cached.get("gptr") or pptr != cached.get("pptr"):
            old_g, old_p = cached.get("gptr"), cached.get("pptr")
            for i, p in enumerate(active):
                if old_g is not None and old_g[i] == gptr[i] and old_p[i] == pptr[i]:
                    continue
                st = self.state[p]
                g = p.grad
                if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                    raise hmx.HmxError("HmxAdamW needs contiguous fp32 gradients on the GPU")
                host[i] = (pptr[i], gptr[i], st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            cached["gptr"], cached["pptr"] = gptr, pptr
            # Through a pinned staging block, asynchronously: a copy from pageable memory would make the host wait for
            # everything the stream still holds (the whole backward pass), and the next step's launches would start
            # only then.  The host allocator keeps the block until the copy that reads it has run.
            staged = torch.from_numpy(host.view(np.uint8).copy()).pin_memory()
            cached["desc"].copy_(staged, non_blocking=True)
        return cached

    def _row_wise(self, p):
        """The TableGrad of a parameter that takes the row-wise kernel, else None."""
        if not self.sparse_table_grads or self.keep_clipped_grads or p.dim() != 2:
            return None
        tg = ops.table_grad_state(p)
        if tg is None or tg.dense or p.shape[1] % 4 != 0 or p.shape[1] > 256 or not p.is_contiguous():
            return None
        return tg

    def _table_array(self, pairs, with_copy):
        """(ctypes array of hmx_adamw_table for these (parameter, TableGrad) pairs, the PreparedTable of each or None)."""
        arr = (hmx.AdamwTable * max(len(pairs), 1))()
        copies = []
        for i, (p, tg) in enumerate(pairs):
            st = self.state[p]
            a = arr[i]
            a.p, a.g, a.m, a.v = p.data_ptr(), tg.buf.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            a.n_rows, a.d = p.shape[0], p.shape[1]
            a.row_flags = tg.flags.data_ptr()
            pt = ops.prepared_copy_of(p) if with_copy else None
            if pt is not None:
                a.bf16_out, a.bf16_ld, a.bf16_max_norm = pt.bf16.data_ptr(), pt.bf16.shape[1], pt.max_row_norm.data_ptr()
            copies.append(pt)
        return arr, copies

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}              # the loaded moments live elsewhere

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = hmx.lib()
        stream = hmx.current_stream()
        tables = []          # flat tensor lists: (group, parameters, device table, step)
        row_tables = []      # embedding tables on the row-wise kernel: (group, parameters, their TableGrad, step)
        n_row_wise = 0
        for gi, group in enumerate(self.param_groups):
            self._init_state(group)
            # one launch per (group, step count): parameters that skipped steps (no gradient) keep their own count
            by_step, rows_by_step = {}, {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                tg = self._row_wise(p) if n_row_wise < 8 else None       # one launch takes eight tables
                n_row_wise += tg is not None
                (rows_by_step if tg is not None else by_step).setdefault(int(self.state[p]["step"]), []).append(
                    (p, tg) if tg is not None else p)
            for k, (st, active) in enumerate(sorted(by_step.items())):
                tables.append((group, active, self._table((gi, k), active), st + 1))
            for st, pairs in sorted(rows_by_step.items()):
                row_tables.append((group, [p for p, _ in pairs], [t for _, t in pairs], st + 1))
        if not tables and not row_tables:
            return loss
        dev = (tables[0][1][0] if tables else row_tables[0][1][0]).device
        scale_ptr = None
        if self.max_grad_norm is not None:
            # clip_grad_norm_ takes one norm over every gradient handed to it: all groups together
            if len(tables) == 1:
                allt = tables[0][2]
            elif tables:
                allt = self._table(-1, [p for _, act, _, _ in tables for p in act])
            else:
                allt = None
            all_rows = [(p, t) for _, ps, ts, _ in row_tables for p, t in zip(ps, ts)]
            arr, _keep = self._table_array(all_rows, with_copy=False)
            n_part = (allt["chunks"] if allt else 0) + sum(int(L.hmx_adamw_table_blocks(p.shape[0])) for p, _ in all_rows)
            ws = torch.empty(max(int(L.hmx_grad_norm_workspace_bytes(n_part)), 256), dtype=torch.uint8, device=dev)
            norm = torch.empty(2, dtype=torch.float32, device=dev)
            hmx.check(L.hmx_grad_norm_f32(allt["desc"].data_ptr() if allt else None,
                                          allt["starts"].data_ptr() if allt else None, len(allt["host"]) if allt else 0,
                                          allt["chunks"] if allt else 0, arr, len(all_rows), float(self.max_grad_norm),
                                          ws.data_ptr(), ws.numel(), norm.data_ptr(), stream), "hmx_grad_norm_f32")
            self.last_grad_norm = norm
            scale_ptr = norm.data_ptr() + 4
        for group, ps, tgs, step in row_tables:
            torch._foreach_add_([self.state[p]["step"] for p in ps], 1)
            b1, b2 = group["betas"]
            arr, copies = self._table_array(list(zip(ps, tgs)), with_copy=True)
            hmx.check(L.hmx_adamw_tables_step_f32(arr, len(ps), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                  float(group["weight_decay"]), step, scale_ptr, stream),
                      "hmx_adamw_tables_step_f32")
            for p, tg, pt in zip(ps, tgs, copies):
                tg.clean = True                    # gradient rows zeroed, flags cleared: the next step's zero gradient
                if pt is not None:
                    pt.synced_version = p._version
        for group, active, t, step in tables:
            torch._foreach_add_([self.state[p]["step"] for p in active], 1)
            b1, b2 = group["betas"]
            hmx.check(L.hmx_adamw_step_f32(t["desc"].data_ptr(), t["starts"].data_ptr(), len(active), t["chunks"],
                                           float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), step, scale_ptr,
                                           1 if (self.keep_clipped_grads and scale_ptr) else 0, stream),
                      "hmx_adamw_step_f32")
        return loss
