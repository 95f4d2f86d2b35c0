"""GradientSync's exchange at the train step's real sizes (item table 4 M x 64, user table 1 M x 64, ~1.5 M dense
parameters, ~300 K touched item rows and ~1 K touched user rows per rank) for 2 / 4 / 8 ranks run as threads of one
process on one GPU (tests/test_gpu_rank_grid.py's stand-in for the process group): what one rank's GPU spends per
step on the exchange besides the collectives themselves (touched-row search, row gather, re-zero, scatter-add of
all ranks' rows, bucket copies)."""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from generative_recommenders_amd import parallel                  # noqa: E402
from test_gpu_rank_grid import _FakeDist, _Grid                    # noqa: E402


def run(world, sparse, reps=3):
    grid = _Grid(world)
    fake = _FakeDist(grid)
    parallel.dist = fake
    g = torch.Generator().manual_seed(world)
    d = 64
    def params():
        return (torch.nn.Parameter(torch.zeros(4_000_001, d, device="cuda")),
                torch.nn.Parameter(torch.zeros(1_000_001, d, device="cuda")),
                [torch.nn.Parameter(torch.zeros(n, device="cuda")) for n in (64 * 64,) * 190])
    ids_i = [torch.randint(1, 4_000_001, (300_000,), generator=g).cuda() for _ in range(world)]
    ids_u = [torch.randint(1, 1_000_001, (1024,), generator=g).cuda() for _ in range(world)]
    times, errors = [], []

    def rank_main(rank):
        grid.turn.acquire()
        fake.local.rank = rank
        try:
            item, user, dense = params()                    # every rank its own replica
            sync = parallel.GradientSync([item, user] + dense, sparse_tables=sparse)
            for rep in range(reps):
                gi = torch.zeros_like(item)
                gi[ids_i[rank]] = 1.0
                gu = torch.zeros_like(user)
                gu[ids_u[rank]] = 1.0
                item.grad, user.grad = gi, gu
                for p in dense:
                    p.grad = torch.ones_like(p)
                grid._meet()                                 # every rank has queued its set-up work
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sync()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                grid._meet()                                 # nobody starts the next set-up inside another's window
                if rep == reps - 1:
                    times.append((t1 - t0) * 1e3)
        except BaseException as e:       # noqa: BLE001
            errors.append((rank, repr(e)))
            grid.barrier.abort()
        finally:
            grid.turn.release()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    return times


if __name__ == "__main__":
    import torch.distributed as real_dist
    for world in ([int(a) for a in sys.argv[1:]] or [2, 4, 8]):
        for sparse in (True,):
            ts = run(world, sparse)
            # every rank's wall clock spans the whole lock-step exchange of all ranks: the GPU did `world` ranks' work
            print(f"{world} ranks, {'sparse rows' if sparse else 'dense all-reduce'}: {max(ts):7.2f} ms for all ranks' "
                  f"work -> {max(ts) / world:6.2f} ms per rank and step (collectives themselves not included)")
    parallel.dist = real_dist
