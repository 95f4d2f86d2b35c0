"""Streaming ceilings of the box for the roofline discussion: device-to-device copy (read + write) and fill (write only)
of a 2 GiB fp32 buffer, event-timed."""
import torch

n = 1 << 29
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t = timed(lambda: b.copy_(a))
print(f"copy 2 GiB -> 2 GiB: {t * 1e3:.3f} ms, {2 * n * 4 / t / 1e12:.2f} TB/s (read + write)")
t = timed(lambda: b.zero_())
print(f"fill 2 GiB: {t * 1e3:.3f} ms, {n * 4 / t / 1e12:.2f} TB/s (write)")
t = timed(lambda: a.sum())
print(f"sum 2 GiB: {t * 1e3:.3f} ms, {n * 4 / t / 1e12:.2f} TB/s (read)")
