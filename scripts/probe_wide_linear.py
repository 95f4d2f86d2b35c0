"""Times the first projection of a raw multimodal vector (K = 1024 / 4096 columns into 2d = 128): the chunked HIP
projection with the row normalisation as its input scale, forward and forward + backward (weight gradient), against
torch's normalize + F.linear (library GEMM) on the same tensors."""
import sys
import time

import torch

sys.path.insert(0, ".")
from generative_recommenders_amd import enc_ops as E          # noqa: E402
from generative_recommenders_amd import feature_ops as FO     # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = "cuda"
    T, N = 102400, 128
    for K in (1024, 4096):
        x = torch.randn(T, K, device=dev)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).requires_grad_()
        b = torch.zeros(N, device=dev, requires_grad=True)
        g = torch.randn(T, N, device=dev)

        def hip(bwd):
            y = FO.linear_any(x, W, b, in_scale=FO.row_inv_norm(x))
            if bwd:
                W.grad = b.grad = None
                y.backward(g)

        def ref(bwd):
            y = torch.nn.functional.linear(torch.nn.functional.normalize(x, p=2, dim=-1), W, b)
            if bwd:
                W.grad = b.grad = None
                y.backward(g)

        print(f"K={K}: x is {T * K * 4 / 1e9:.2f} GB")
        for name, fn in (("hip chunked", hip), ("torch normalize + F.linear", ref)):
            with torch.no_grad():
                f = timed(lambda: fn(False))
            fb = timed(lambda: fn(True))
            print(f"  {name:28s} forward {f:7.3f} ms   forward + backward {fb:7.3f} ms")
        E.set_direct_param_grad(True)
        fb = timed(lambda: hip(True))
        E.set_direct_param_grad(False)
        print(f"  {'hip chunked, direct grads':28s} {'':18s} forward + backward {fb:7.3f} ms")
        del x


if __name__ == "__main__":
    main()
