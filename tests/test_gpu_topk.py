"""GPU parity: MFMA scoring + fused top-K through the C ABI vs the C oracle (ids and scores bit-exact),
plus the reference-captured golden (ids from the reference's torch.mm + torch.topk)."""
import numpy as np
import pytest
import torch

import _golden as G
from generative_recommenders_amd import ops
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _data(nq, rows, d, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(rows, d, generator=g) * scale
    t[0] = 0
    q = torch.randn(nq, d, generator=g)
    return q, t


def _check(ids, sc, q, t, k, exclude=None, id_offset=0):
    want_ids, want_sc = C.score_topk(q.numpy(), t.numpy(), k, None if exclude is None else exclude.numpy(), id_offset)
    assert np.array_equal(ids.cpu().numpy(), want_ids)
    assert np.array_equal(sc.cpu().numpy(), want_sc)          # same fma chain => same bits


def test_golden_reference_ids_exact_and_bf16():
    s, p = G.load("g9_topk"), "G9_topk"
    t, q = G.t(s, p + "/table"), G.t(s, p + "/users")
    excl = torch.cat([G.t(s, p + "/pos"), G.t(s, p + "/hist")], dim=1)
    for fn in (lambda *a, **k: ops.score_topk_exact(a[0], a[1], *a[2:], **k),
               lambda *a, **k: ops.score_topk(a[0], ops.PreparedTable(a[1]), *a[2:], **k)):
        ids10, sc10 = fn(q.to(DEV), t.to(DEV), 10)
        assert torch.equal(ids10.cpu(), G.t(s, p + "/ids10"))
        torch.testing.assert_close(sc10.cpu(), G.t(s, p + "/val10"), rtol=1e-5, atol=1e-5)
        ids50, _ = fn(q.to(DEV), t.to(DEV), 50, exclude=excl.to(DEV))
        assert torch.equal(ids50.cpu(), G.t(s, p + "/ids50"))


@pytest.mark.parametrize("nq,rows,d,k", [(5, 3000, 16, 10), (70, 20000, 64, 10), (33, 5001, 60, 50), (1, 64, 64, 10),
                                         (64, 100000, 64, 10), (9, 40000, 128, 50), (40, 777, 6, 3)])
def test_exact_kernel_matches_oracle(nq, rows, d, k):
    q, t = _data(nq, rows, d, 100 + nq)
    ids, sc = ops.score_topk_exact(q.to(DEV), t.to(DEV), k)
    _check(ids, sc, q, t, k)


@pytest.mark.parametrize("nq,rows,d,k", [(5, 3000, 16, 10), (70, 20000, 64, 10), (33, 5001, 60, 50), (1, 64, 64, 10),
                                         (300, 100000, 64, 10), (9, 40000, 128, 50), (5000, 30000, 64, 10),
                                         (17000, 9000, 32, 10), (16500, 40001, 64, 10), (16385, 20011, 64, 50)])
def test_bf16_path_matches_oracle(nq, rows, d, k):
    q, t = _data(nq, rows, d, 200 + nq)
    ids, sc, st = ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), k, return_stats=True)
    _check(ids, sc, q, t, k)
    assert st["uncertain"] <= nq // 10 + 1


def test_exclusions_and_shard_offset():
    q, t = _data(40, 30000, 64, 7)
    g = torch.Generator().manual_seed(8)
    excl = torch.randint(0, 30000, (40, 30), generator=g) + 1000
    top = torch.from_numpy(C.score_topk(q.numpy(), t.numpy(), 8, None, 1000)[0])
    excl[:, :8] = top                                          # knock out the true top-8
    for fn in (lambda: ops.score_topk_exact(q.to(DEV), t.to(DEV), 50, excl.to(DEV), 1000),
               lambda: ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), 50, excl.to(DEV), 1000)):
        ids, sc = fn()
        _check(ids, sc, q, t, 50, excl, 1000)
        got = ids.cpu().numpy()
        assert not any(np.isin(got[i], top[i].numpy()).any() for i in range(got.shape[0]))


def test_ties_and_short_tables():
    t = torch.zeros(40, 16)
    t[:, 0] = torch.tensor([1, 3, 3, 2, 3, 0] * 6 + [3, 3, 1, 0])
    q = torch.zeros(3, 16)
    q[:, 0] = torch.tensor([1.0, 2.0, -1.0])
    for fn in (lambda k: ops.score_topk_exact(q.to(DEV), t.to(DEV), k),
               lambda k: ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), k)):
        for k in (3, 10, 50):                                  # k = 50 > 40 rows: tail is (-1, -inf)
            ids, sc = fn(k)
            _check(ids, sc, q, t, k)


def test_adversarial_near_duplicates_fall_back_to_exact():
    """Rows that differ below bf16 resolution: the coarse pass cannot rank them, so the proof must fail and
    the exact kernel must supply the answer."""
    g = torch.Generator().manual_seed(11)
    base = torch.randn(1, 64, generator=g)
    t = base + 1e-4 * torch.randn(4096, 64, generator=g)
    q = base.repeat(4, 1) + 0.01 * torch.randn(4, 64, generator=g)
    ids, sc, st = ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), 10, return_stats=True)
    _check(ids, sc, q, t, 10)
    assert st["uncertain"] > 0
    # the same repair without reading the count back: the exact kernels run on the flagged queries, in place
    ids2, sc2 = ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), 10)
    assert torch.equal(ids2, ids) and torch.equal(sc2, sc)


@pytest.mark.parametrize("d", [64, 128])
def test_many_query_launch_with_planted_rows_and_near_duplicates(d):
    """The many-query launch shape (shared item stream, sampled start thresholds) on a table built to stress the
    start thresholds: every query's best rows are planted scaled copies placed outside and inside the sampled
    tiles, and a block of near-duplicate rows, larger than the two candidate lists of a query, tops query 0's scores."""
    g = torch.Generator().manual_seed(17)
    nq, rows = 16400, 70_003
    t = torch.randn(rows, d, generator=g)
    q = torch.randn(nq, d, generator=g)
    spots = torch.randint(1, rows, (nq,), generator=g)
    q[:8000] = t[spots[:8000]] * 2.5                        # planted winners at random rows
    dup = q[0] / q[0].norm() * 9.0
    t[50_000:50_300] = dup + 1e-4 * torch.randn(300, d, generator=g)  # 300 rows the bf16 pass cannot rank (> 2 lists)
    ids, sc, st = ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), 10, return_stats=True)
    _check(ids, sc, q, t, 10)
    assert st["uncertain"] >= 1                               # query 0 at least needs the exact pass
    assert st["uncertain"] < nq // 4                           # ... and most need none: the flag-driven repair below
    ids2, sc2 = ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), 10)      # leaves unflagged queries alone
    assert torch.equal(ids2, ids) and torch.equal(sc2, sc)


def test_shared_stream_give_up_ends_early_and_is_rerun_exactly():
    """A lost hand-off inside the shared-stream kernel (armed through the library's test hook: workgroup 0 never
    publishes stage 7) must cost one bounded wait for that workgroup, not one per remaining stage: the launch comes
    back quickly, every query of it is flagged, and the exact re-run still yields the oracle's ids and scores."""
    import time
    from generative_recommenders_amd import hmx
    nq, rows = 16500, 70_003
    q, t = _data(nq, rows, 64, 41)
    qd, prep = q.to(DEV), ops.PreparedTable(t.to(DEV))
    ids0, sc0, st0 = ops.score_topk(qd, prep, 10, return_stats=True)      # healthy run (also warms everything up)
    assert st0["uncertain"] <= nq // 10
    hmx.check(hmx.lib().hmx_debug_ring_skip_publish(7), "hmx_debug_ring_skip_publish")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids, sc, st = ops.score_topk(qd, prep, 10, return_stats=True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert st["uncertain"] == nq                                           # the whole launch was flagged ...
    assert elapsed < 2.0, elapsed                                          # ... after ONE bounded wait (~0.1 s)
    assert torch.equal(ids, ids0) and torch.equal(sc, sc0)
    _check(ids[:512], sc[:512], q[:512], t, 10)
    ids2, sc2, st2 = ops.score_topk(qd, prep, 10, return_stats=True)      # the hook is one-shot
    assert st2["uncertain"] == st0["uncertain"] and torch.equal(ids2, ids0)


@pytest.mark.parametrize("nq,rows,k,with_excl,d", [(1024, 300_017, 50, True, 64), (640, 262_144, 10, False, 64),
                                                   (4100, 270_001, 10, False, 64), (4096, 262_200, 50, True, 64),
                                                   (1024, 300_017, 50, True, 128), (600, 262_208, 10, False, 128),
                                                   (4100, 100_001, 10, False, 128), (8200, 70_070, 50, True, 100)])
def test_few_query_shared_stream_shapes(nq, rows, k, with_excl, d):
    """Few queries over a long table (the hard-negative miner's shape): workgroups of eight waves with one or two
    query groups per wave share the item stream, after the seeding passes over the table prefixes.  Rows of 128
    bf16 (d = 128, and d = 100 padded to it) take the two-tile stages of the same kernel."""
    q, t = _data(nq, rows, d, 300 + nq)
    excl = None
    if with_excl:
        g = torch.Generator().manual_seed(31)
        excl = torch.randint(0, rows, (nq, 20), generator=g)
        top = torch.from_numpy(C.score_topk(q[:64].numpy(), t.numpy(), 5, None, 0)[0])
        excl[:64, :5] = top                                     # knock out true winners of the first queries
    ids, sc, st = ops.score_topk(q.to(DEV), ops.PreparedTable(t.to(DEV)), k,
                                 None if excl is None else excl.to(DEV), return_stats=True)
    _check(ids, sc, q, t, k, excl)
    assert st["uncertain"] <= nq // 10 + 1


def test_merge_matches_unsharded():
    q, t = _data(50, 30001, 64, 12)
    full_ids, full_sc = ops.score_topk_exact(q.to(DEV), t.to(DEV), 10)
    bounds = [0, 7000, 15000, 22222, 30001]
    parts = [ops.score_topk(q.to(DEV), ops.PreparedTable(t[a:b].to(DEV)), 10, id_offset=a)
             for a, b in zip(bounds[:-1], bounds[1:])]
    ids, sc = ops.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(ids, full_ids) and torch.equal(sc, full_sc)
    want_ids, want_sc = C.topk_merge(torch.stack([p[0] for p in parts]).cpu().numpy(),
                                     torch.stack([p[1] for p in parts]).cpu().numpy())
    assert np.array_equal(ids.cpu().numpy(), want_ids) and np.array_equal(sc.cpu().numpy(), want_sc)


def test_full_size_properties():
    """4M x 64 table: (1) bf16 path == exact kernel on 256 queries; (2) a query equal to a scaled table row ranks
    that row first; (3) returned scores are sorted and re-computable."""
    rows, d = 4_000_001, 64
    g = torch.Generator(device=DEV).manual_seed(13)
    t = torch.randn(rows, d, device=DEV, generator=g)
    t[0] = 0
    prep = ops.PreparedTable(t)
    q = torch.randn(256, d, device=DEV, generator=g)
    planted = torch.randint(1, rows, (256,), device=DEV, generator=g)
    q[:128] = t[planted[:128]] * 3.0
    ids, sc, st = ops.score_topk(q, prep, 10, return_stats=True)
    ids_e, sc_e = ops.score_topk_exact(q, t, 10)
    assert torch.equal(ids, ids_e) and torch.equal(sc, sc_e)
    assert torch.equal(ids[:128, 0], planted[:128])
    assert torch.all(sc[:, :-1] >= sc[:, 1:])
    torch.testing.assert_close(sc, (q[:, None, :] * t[ids]).sum(-1), rtol=1e-4, atol=1e-4)
    assert st["uncertain"] == 0
    # (4) the many-query launch shape (workgroups of eight waves sharing the item stream) gives the same answer
    big = torch.cat([q, torch.randn(16500, d, device=DEV, generator=g)])
    ids_b, sc_b, st_b = ops.score_topk(big, prep, 10, return_stats=True)
    assert torch.equal(ids_b[:256], ids_e) and torch.equal(sc_b[:256], sc_e)
    assert torch.all(sc_b[:, :-1] >= sc_b[:, 1:]) and bool((ids_b >= 0).all())
    torch.testing.assert_close(sc_b, (big[:, None, :] * t[ids_b]).sum(-1), rtol=1e-4, atol=1e-4)
    assert st_b["uncertain"] == 0
    # (5) the hard-negative miner's shape: 1024 queries, top-50, positives + history excluded (seeding passes over the
    # table prefixes, then eight-wave workgroups with one query group per wave)
    qm = torch.randn(1024, d, device=DEV, generator=g)
    excl = torch.randint(1, rows, (1024, 200), device=DEV, generator=g)
    excl[:, :10] = ops.score_topk_exact(qm, t, 10)[0]           # the true top-10 are on every query's list
    ids_m, sc_m, st_m = ops.score_topk(qm, prep, 50, excl, return_stats=True)
    ids_x, sc_x = ops.score_topk_exact(qm, t, 50, excl)
    assert torch.equal(ids_m, ids_x) and torch.equal(sc_m, sc_x)
    assert not bool((ids_m[:, :, None] == excl[:, None, :10]).any())
    assert st_m["uncertain"] <= 8


def test_full_size_table_against_c_oracle():
    """16 queries x 4,000,001 rows: ids and scores of both HIP paths against the C oracle's exhaustive scan (the
    same fma-chain score definition), bit for bit -- the full-size check that does not compare HIP with HIP."""
    rows, d = 4_000_001, 64
    g = torch.Generator().manual_seed(77)
    t = torch.randn(rows, d, generator=g)
    t[0] = 0
    q = torch.randn(16, d, generator=g)
    want_ids, want_sc = C.score_topk(q.numpy(), t.numpy(), 10)
    td = t.to(DEV)
    for ids, sc in (ops.score_topk(q.to(DEV), ops.PreparedTable(td), 10), ops.score_topk_exact(q.to(DEV), td, 10)):
        assert np.array_equal(ids.cpu().numpy(), want_ids)
        assert np.array_equal(sc.cpu().numpy(), want_sc)


def test_launch_shape_toggles_do_not_change_results(tmp_path):
    """The shared-stream kernel and the sampled start thresholds are speed features: with either switched off
    (fresh processes, the switches are read once) ids and scores must be bit-identical to the default path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from generative_recommenders_amd import ops\n"
        "g = torch.Generator().manual_seed(23)\n"
        "t = torch.randn(300_001, 64, generator=g).cuda(); q = torch.randn(16500, 64, generator=g).cuda()\n"
        "ids, sc = ops.score_topk(q, ops.PreparedTable(t), 10)\n"
        "np.save(sys.argv[1] + '_ids.npy', ids.cpu().numpy()); np.save(sys.argv[1] + '_sc.npy', sc.cpu().numpy())\n")
    outs = {}
    for name, env in (("default", {}), ("noring", {"HMX_TOPK_RING": "0"}), ("nowarm", {"HMX_TOPK_WARM": "0"})):
        e = dict(os.environ)
        e.update(env)
        prefix = str(tmp_path / name)
        subprocess.run([sys.executable, "-c", code, prefix], check=True, env=e, timeout=240)
        outs[name] = (np.load(prefix + "_ids.npy"), np.load(prefix + "_sc.npy"))
    for name in ("noring", "nowarm"):
        assert np.array_equal(outs[name][0], outs["default"][0]), name
        assert np.array_equal(outs[name][1], outs["default"][1]), name


@pytest.mark.parametrize("stress,d", [(False, 64), (True, 64), (False, 128)])
def test_threshold_exchange_between_item_ranges(stress, d):
    """Three item ranges emulated in one process: thresholds sampled per range, the k-th largest of all sampled
    scores taken as every range's start, seeded passes, merge, completeness against the ranges' drop bounds, exact
    re-run of what fails it.  The result must equal the unsharded kernel's for all 262,144 queries and the C
    oracle's on a slice.  stress: planted winners for half the queries (inside and outside the sampled tiles) and a
    block of near-duplicate rows that straddles two ranges and tops query 0's scores -- the bf16 pass cannot rank
    those, so the merged proof must fail for query 0 and the exact re-run must supply the answer."""
    from oracle import c_oracle as C
    g = torch.Generator().manual_seed(77)
    rows, k, S, nq = 60_003, 10, 3, 262_144
    table = torch.randn(rows, d, generator=g)
    q = torch.randn(nq, d, generator=g)
    per = -(-rows // S)
    if stress:
        spots = torch.randint(1, rows, (nq // 2,), generator=g)
        q[:nq // 2] = table[spots] * 2.5
        dup = q[0] / q[0].norm() * 9.0
        table[per - 150:per + 150] = dup + 1e-4 * torch.randn(300, d, generator=g)
    tq, qq = table.cuda(), q.cuda()
    preps = [ops.PreparedTable(tq[s * per:min((s + 1) * per, rows)]) for s in range(S)]
    thr = torch.stack([ops.sample_thresholds(qq, p, k, S) for p in preps])
    assert torch.isfinite(thr).all()                       # this launch shape samples
    vals = torch.cat([ops.sample_thresholds(qq, p, k, S, 4) for p in preps], dim=1)       # (nq, S * 4) sampled scores
    assert torch.isfinite(vals).all()
    start = vals.topk(k, dim=1).values[:, k - 1].contiguous()
    assert torch.equal(start, ops.kth_largest(vals.view(nq, S, 4).permute(1, 0, 2).contiguous(), k))
    assert bool((start >= thr.amin(0)).all())              # k-th of the union is no looser than the loosest range
    lists = [ops.score_topk_seeded(qq, preps[s], k, start, id_offset=s * per) for s in range(S)]
    ids, sc = ops.topk_merge(torch.stack([l[0] for l in lists]), torch.stack([l[1] for l in lists]))
    bound = torch.stack([l[2] for l in lists]).amax(0)
    failed = torch.stack([l[3] for l in lists]).amax(0) > 0
    assert not bool(failed.any())
    flagged = ~(sc[:, k - 1] > bound)
    assert int(flagged.sum()) < nq // 100                  # the shared start leaves almost every query provable
    if stress:
        assert bool(flagged[0])
    # some range's own list is short for many queries: that is the point of the shared start
    assert int(sum((l[0][:, -1] < 0).sum() for l in lists)) > 0
    redo = flagged.nonzero().flatten()
    if redo.numel():
        ex = [ops.score_topk_exact(qq[redo], preps[s].table, k, id_offset=s * per) for s in range(S)]
        ids[redo], sc[redo] = ops.topk_merge(torch.stack([e[0] for e in ex]), torch.stack([e[1] for e in ex]))
    want_ids, want_sc = ops.score_topk(qq, ops.PreparedTable(tq), k)
    assert torch.equal(ids, want_ids) and torch.equal(sc, want_sc)
    for sl in (slice(0, 32), slice(1000, 1032)):
        o_ids, o_sc = C.score_topk(q[sl].numpy(), table.numpy(), k)
        assert np.array_equal(ids[sl].cpu().numpy(), o_ids) and np.array_equal(sc[sl].cpu().numpy(), o_sc)


@pytest.mark.parametrize("nq,rows,d,k", [(40_000, 70_001, 64, 10), (100_000, 70_001, 64, 10), (131_072, 30_001, 64, 16),
                                         (150_000, 5000, 64, 16), (70_000, 5000, 128, 16), (100_000, 70_001, 128, 50)])
def test_launches_between_the_plan_switch_points(nq, rows, d, k):
    """Launches of 16 K to 262 K queries: six, two and one item splits (seeded from table prefixes, sampled start for
    two splits, sampled start for one), short tables, d = 128, with a block of near-duplicate rows that sends some
    queries to the exact re-run.  All queries equal the exact fp32 kernel, a slice equals the C oracle."""
    g = torch.Generator().manual_seed(nq + rows)
    t = torch.randn(rows, d, generator=g)
    t[100:140] = t[100] + 1e-4 * torch.randn(40, d, generator=g)
    q = torch.randn(nq, d, generator=g)
    q[:7] = t[100] * 3
    tq, qq = t.to(DEV), q.to(DEV)
    ids, sc = ops.score_topk(qq, ops.PreparedTable(tq), k)
    want_ids, want_sc = ops.score_topk_exact(qq, tq, k)
    assert torch.equal(ids, want_ids) and torch.equal(sc, want_sc)
    _check(ids[:24], sc[:24], q[:24], t, k)


@pytest.mark.parametrize("nq,sampled", [(213_568, True), (262_144, True), (100_000, True), (65_536, True), (40_000, False),
                                        (4096, False)])
def test_which_launch_shapes_sample_their_start(nq, sampled):
    """Launches of one to four item splits (512 query tiles or more) have sampled start thresholds to exchange between
    item ranges -- the ragged last launch of a 1 M-query search (213,568 queries) among them; launches of more splits
    report -inf (they are seeded from table prefixes instead)."""
    g = torch.Generator().manual_seed(nq)
    t = torch.randn(40_000, 64, generator=g).to(DEV)
    q = torch.randn(nq, 64, generator=g).to(DEV)
    prep = ops.PreparedTable(t)
    thr = ops.sample_thresholds(q, prep, 10, n_ranges=2)
    vals = ops.sample_thresholds(q, prep, 10, n_ranges=2, n_values=4)
    if sampled:
        assert bool(torch.isfinite(thr).all()) and bool(torch.isfinite(vals).all())
        exact = ops.score_topk_exact(q[:256], t, 10)[1]
        assert bool((thr[:256] <= exact[:, 9]).all())          # ten rows of the table do score that much
    else:
        assert bool(torch.isinf(thr).all()) and bool((thr < 0).all()) and bool(torch.isinf(vals).all())
