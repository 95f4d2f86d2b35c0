"""The sharded retrieval of N ranks (one item range each) run through ShardedRetriever.search on ONE GPU, ranks as
threads that take turns (tests/test_gpu_rank_grid.py's stand-in for the process group).  The GPU does the work of all N
ranks one after the other, so (time of the whole emulated search) / N is what one rank's GPU spends per search --
sampling, seeded pass, packing, merges of the N lists, flag handling -- with the collectives themselves costing
nothing (a hand-over of device tensors).  Prints that per-rank time beside the one-GPU search."""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from generative_recommenders_amd import ops                      # noqa: E402
from test_gpu_rank_grid import _run_grid                          # noqa: E402


def main():
    rows, n_q, k, d = 4_000_001, 1_000_000, 10, 64
    g = torch.Generator().manual_seed(0)
    table = torch.randn(rows, d, generator=g).cuda()
    q = torch.randn(n_q, d, generator=g).cuda()
    prep = ops.PreparedTable(table)
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        want = ops.score_topk(q, prep, k)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) * 1e3
    print(f"one GPU, whole table: {one:.1f} ms per 1 M-query search")
    for world in ([int(a) for a in sys.argv[1:]] or [2, 4, 8]):
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = _run_grid(table, q, k, world, world, 262_144)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            best = ms if best is None else min(best, ms)
        ok = all(torch.equal(r[0], want[0]) and torch.equal(r[1], want[1]) for r in res)
        print(f"{world} ranges: all ranks' work on one GPU {best:.1f} ms -> {best / world:.1f} ms per rank "
              f"(x{one / (best / world):.2f} of one GPU before communication), uncertain {res[0][2]['uncertain']}, "
              f"equal to the one-GPU result: {ok}")


if __name__ == "__main__":
    main()
