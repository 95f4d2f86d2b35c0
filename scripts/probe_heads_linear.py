"""Times hmx_heads_linear_f32 on the deltaBC shapes of the configs[1] train step (T = 102,400 tokens, two heads of
128 channels, 48 projection outputs): forward (K = 128 -> N = 48) and input gradient (K = 48 -> N = 128, with and
without accumulation)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_recommenders_amd import hmx  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
H, ED, P = 2, 128, 48
dev = "cuda"
L = hmx.lib()
s = torch.cuda.current_stream().cuda_stream
xc = torch.randn(T, H * ED, device=dev)
dbc = torch.randn(T, H * P, device=dev)
ws = [torch.randn(P, ED, device=dev) for _ in range(H)]
ptrs = (ctypes.c_void_p * H)(*[w.data_ptr() for w in ws])


def timed(name, fn, nbytes, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    print(f"{name}: {us:.1f} us, {nbytes / us / 1e6:.2f} TB/s")


timed("forward  K=128 -> N=48 ", lambda: hmx.check(L.hmx_heads_linear_f32(xc.data_ptr(), H * ED, ED, ptrs, ED, 1, T, P, ED,
      dbc.data_ptr(), H * P, P, H, 0, s), "f"), T * H * (ED + P) * 4)
for acc in (0, 1):
    timed(f"backward K=48 -> N=128 accumulate={acc}", lambda: hmx.check(L.hmx_heads_linear_f32(dbc.data_ptr(), H * P, P, ptrs, 1, ED,
          T, ED, P, xc.data_ptr(), H * ED, ED, H, acc, s), "b"), T * H * (P + ED * (1 + acc)) * 4)
