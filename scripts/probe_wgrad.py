"""Times hmx_linear_wgrad_batch_f32 on batches of identical products (run on the GPU box with PYTHONPATH=.)."""
import sys
import torch
from generative_recommenders_amd import hmx

DEV = "cuda"


def batch(T, N, K, n_jobs, bias=False):
    descs, keep = [], []
    for _ in range(n_jobs):
        dy = torch.randn(T, N, device=DEV)
        x = torch.randn(T, K, device=DEV)
        w = torch.zeros(N, K, device=DEV)
        b = torch.zeros(N, device=DEV)
        keep.append((dy, x, w, b))
        descs.append(hmx.WgradDesc(dy.data_ptr(), N, x.data_ptr(), K, None, T, N, K, w.data_ptr(), K,
                                   b.data_ptr() if bias else None, 1, 0, 0, 0, 0))
    return (hmx.WgradDesc * n_jobs)(*descs), keep


def run(T, N, K, n_jobs, bias=False, reps=20):
    lib = hmx.lib()
    arr, keep = batch(T, N, K, n_jobs, bias)
    ws = torch.empty(lib.hmx_linear_wgrad_batch_workspace_bytes(arr, n_jobs), dtype=torch.uint8, device=DEV)
    s = hmx.current_stream()
    for _ in range(3):
        hmx.check(lib.hmx_linear_wgrad_batch_f32(arr, n_jobs, ws.data_ptr(), ws.numel(), s), "wgrad")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.hmx_linear_wgrad_batch_f32(arr, n_jobs, ws.data_ptr(), ws.numel(), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    flop = 2.0 * T * N * K * n_jobs
    byts = 4.0 * T * (N + K) * n_jobs
    print(f"T={T:7d} N={N:4d} K={K:4d} jobs={n_jobs:2d} bias={int(bias)}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s  "
          f"{byts / us / 1e6:6.2f} TB/s (operands once)  ws={ws.numel() / 1e6:.1f} MB", flush=True)


if __name__ == "__main__":
    for T in (25600, 262144):
        for (N, K) in ((64, 64), (128, 128), (128, 32), (32, 128), (256, 64), (64, 256)):
            for nj in (1, 24):
                run(T, N, K, nj)
    run(25600, 64, 64, 24, bias=True)


def run_mixed(specs, label, reps=20):
    """specs: (T, N, K, ldy, ldx, bias, mask) per product."""
    lib = hmx.lib()
    descs, keep = [], []
    for (T, N, K, ldy, ldx, bias, mask) in specs:
        dy = torch.randn(T, ldy, device=DEV)
        x = torch.randn(T, ldx, device=DEV)
        w = torch.zeros(N, K, device=DEV)
        b = torch.zeros(N, device=DEV)
        m = torch.ones(T, device=DEV)
        keep.append((dy, x, w, b, m))
        descs.append(hmx.WgradDesc(dy.data_ptr(), ldy, x.data_ptr(), ldx, m.data_ptr() if mask else None, T, N, K,
                                   w.data_ptr(), K, b.data_ptr() if bias else None, 1, 0, 0, 0, 0))
    n = len(descs)
    arr = (hmx.WgradDesc * n)(*descs)
    ws = torch.empty(lib.hmx_linear_wgrad_batch_workspace_bytes(arr, n), dtype=torch.uint8, device=DEV)
    s = hmx.current_stream()
    for _ in range(3):
        hmx.check(lib.hmx_linear_wgrad_batch_f32(arr, n, ws.data_ptr(), ws.numel(), s), "wgrad")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.hmx_linear_wgrad_batch_f32(arr, n, ws.data_ptr(), ws.numel(), s)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) * 1000 / reps:8.1f} us", flush=True)


if __name__ == "__main__":
    big = (25600, 64, 64, 64, 64, False, False)
    run_mixed([big] * 3, "3 big")
    run_mixed([big] * 3 + [(1280, 64, 64, 64, 64, False, False)] * 21, "3 big + 21 small")
    run_mixed([(25600, 64, 64, 64, 64, False, True)] * 24, "24 big with mask")
    run_mixed([(25600, 64, 64, 64, 64, True, True)] * 24, "24 big with mask and bias")
    run_mixed([(25600, 128, 32, 256, 64, False, False)] * 24, "24 x (128, 32) in wider rows")
    run_mixed([(25600, 48, 128, 96, 256, False, False)] * 24, "24 x (48, 128) in wider rows")
    run_mixed([(25600, 32, 128, 64, 256, False, False)] * 24, "24 x (32, 128) in wider rows")
