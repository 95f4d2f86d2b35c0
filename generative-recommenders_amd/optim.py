"""AdamW with the gradient-norm clip folded in, on the HIP kernels (hmx_grad_norm_f32, hmx_adamw_step_f32).

Replaces, in the train loop of main.py:640,709-713,

    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    optimizer.step()                                  # torch.optim.AdamW(lr, betas=(0.9, 0.98), weight_decay=...)

by `optimizer.step()` of `HmxAdamW(model.parameters(), ..., max_grad_norm=1.0)`: one pass over the gradients for the
norm, one pass over parameters / gradients / moments for the update, the clip factor applied inside the update and
never leaving the device.  It is a torch.optim.Optimizer (param_groups, state_dict with step / exp_avg / exp_avg_sq per
parameter, LR schedulers), so the rest of the loop is unchanged.
"""
import numpy as np
import torch

from . import hmx

_DESC = np.dtype([("p", np.uint64), ("g", np.uint64), ("m", np.uint64), ("v", np.uint64), ("n", np.int64)])


class HmxAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 keep_clipped_grads=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("HmxAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.keep_clipped_grads = bool(keep_clipped_grads)     # also write the clipped gradients back into .grad
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
        tables = []
        for gi, group in enumerate(self.param_groups):
            self._init_state(group)
            # one launch per (group, step count): parameters that skipped steps (no gradient) keep their own count
            by_step = {}
            for p in group["params"]:
                if p.grad is not None:
                    by_step.setdefault(int(self.state[p]["step"]), []).append(p)
            for k, (st, active) in enumerate(sorted(by_step.items())):
                tables.append((group, active, self._table((gi, k), active), st + 1))
        if not tables:
            return loss
        scale_ptr = None
        if self.max_grad_norm is not None:
            # clip_grad_norm_ takes one norm over every gradient handed to it: all groups together
            if len(tables) == 1:
                allt = tables[0][2]
            else:
                allt = self._table(-1, [p for _, act, _, _ in tables for p in act])
            dev = tables[0][1][0].device
            ws = torch.empty(max(int(L.hmx_grad_norm_workspace_bytes(allt["chunks"])), 256), dtype=torch.uint8, device=dev)
            norm = torch.empty(2, dtype=torch.float32, device=dev)
            hmx.check(L.hmx_grad_norm_f32(allt["desc"].data_ptr(), allt["starts"].data_ptr(), len(allt["host"]),
                                          allt["chunks"], float(self.max_grad_norm), ws.data_ptr(), ws.numel(),
                                          norm.data_ptr(), stream), "hmx_grad_norm_f32")
            self.last_grad_norm = norm
            scale_ptr = norm.data_ptr() + 4
        for group, active, t, step in tables:
            torch._foreach_add_([self.state[p]["step"] for p in active], 1)
            b1, b2 = group["betas"]
            hmx.check(L.hmx_adamw_step_f32(t["desc"].data_ptr(), t["starts"].data_ptr(), len(active), t["chunks"],
                                           float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), step, scale_ptr,
                                           1 if (self.keep_clipped_grads and scale_ptr) else 0, stream),
                      "hmx_adamw_step_f32")
        return loss
