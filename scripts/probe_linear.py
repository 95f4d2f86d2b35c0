"""Timing probe of the projection kernel: python scripts/probe_linear.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd import enc_ops as E
T = 102400
for (K, N, res) in ((64, 64, False), (64, 64, True), (64, 256, False), (128, 48, False), (128, 32, False), (32, 256, False), (64, 128, False)):
    x = torch.randn(T, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.1
    r = torch.randn(T, N, device="cuda") if res else None
    for _ in range(3):
        y = E.linear(x, w, None, r1=r, c1=1.0 if res else 0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = E.linear(x, w, None, r1=r, c1=1.0 if res else 0.0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byts = T * (K + N + (N if res else 0)) * 4
    print(f"K={K:4d} N={N:4d} res={int(res)}  {us:7.1f} us  {byts / us / 1e6:6.2f} TB/s  {2.0 * T * N * K / us / 1e6:6.1f} TFLOP/s (x3 MFMA in split mode)")
