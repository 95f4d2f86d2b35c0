"""HmxAdamW (C ABI hmx_grad_norm_f32 / hmx_adamw_step_f32) against torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW on the same parameters and gradients.  Tolerance: 2e-6 of the tensor's largest magnitude on
parameters, moments and clipped gradients after several steps (same formulas in fp32, a few ulps apart through
fused multiply-adds and the summation order of the norm); the norm itself 2e-6 relative."""
import pytest
import torch

from generative_recommenders_amd.optim import HmxAdamW

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, want, rel=2e-6):
    scale = float(want.detach().abs().max()) + 1e-30
    err = float((got.detach() - want.detach()).abs().max())
    assert err <= rel * scale, (err, scale)
SHAPES = [(70_001, 64), (64, 64), (64,), (3, 5, 7), (1,), (40_000,), (129, 33)]     # > 1 chunk, odd sizes, scalars


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).to(DEV)) for s in SHAPES]


def _grads(ps, seed, scale, skip=()):
    g = torch.Generator().manual_seed(seed)
    for i, p in enumerate(ps):
        gr = (torch.randn(*p.shape, generator=g) * scale).to(DEV)
        p.grad = None if i in skip else gr


@pytest.mark.parametrize("max_norm", [1.0, None])
def test_matches_torch_adamw_with_clip(max_norm):
    a, b = _params(1), _params(1)
    ref = torch.optim.AdamW(a, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    opt = HmxAdamW(b, lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1, max_grad_norm=max_norm,
                   keep_clipped_grads=True)
    for step, scale in enumerate((3.0, 1e-4, 0.5, 2.0)):       # clipped, untouched, clipped ...
        skip = (5,) if step == 1 else ()                        # one parameter without a gradient for one step
        _grads(a, 10 + step, scale, skip)
        _grads(b, 10 + step, scale, skip)
        if max_norm is not None:
            want_norm = torch.nn.utils.clip_grad_norm_([p for p in a if p.grad is not None], max_norm)
        ref.step()
        opt.step()
        if max_norm is not None:
            torch.testing.assert_close(opt.last_grad_norm[0], want_norm, rtol=2e-6, atol=0)
    for p, q in zip(a, b):
        _close(q, p)
        _close(opt.state[q]["exp_avg"], ref.state[p]["exp_avg"])
        _close(opt.state[q]["exp_avg_sq"], ref.state[p]["exp_avg_sq"])
        assert int(opt.state[q]["step"]) == int(ref.state[p]["step"])
        if max_norm is not None:
            _close(q.grad, p.grad)     # clipped values left in .grad


def test_state_dict_round_trip_and_scheduler():
    a = _params(2)
    opt = HmxAdamW(a, lr=1e-2, betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 0.5 ** s)
    _grads(a, 3, 1.0)
    opt.step()
    sched.step()
    assert abs(opt.param_groups[0]["lr"] - 5e-3) < 1e-12
    import copy
    sd = copy.deepcopy(opt.state_dict())     # as if read back from a checkpoint file (no aliasing of the live moments)
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    opt2 = HmxAdamW(b, lr=1e-2, betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0)
    opt2.load_state_dict(sd)
    _grads(a, 4, 1.0)
    _grads(b, 4, 1.0)
    opt.step()
    opt2.step()
    for p, q in zip(a, b):
        assert torch.equal(p, q)


def test_two_groups_share_one_norm():
    a, b = _params(5), _params(5)
    ref = torch.optim.AdamW([{"params": a[:3], "lr": 1e-2}, {"params": a[3:], "lr": 1e-3, "weight_decay": 0.0}],
                            betas=(0.9, 0.98), weight_decay=0.1)
    opt = HmxAdamW([{"params": b[:3], "lr": 1e-2}, {"params": b[3:], "lr": 1e-3, "weight_decay": 0.0}],
                   betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0)
    for step in range(2):
        _grads(a, 20 + step, 2.0)
        _grads(b, 20 + step, 2.0)
        torch.nn.utils.clip_grad_norm_(a, 1.0)
        ref.step()
        opt.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("d", [64, 60])
def test_row_wise_table_update_with_flags_and_scoring_copy(d):
    """Embedding tables in fused-table-gradient mode: the scatter-adds flag the rows they write, the optimizer reads
    only those rows' gradients, zeroes them, clears the flags and rewrites the table's bf16 scoring copy.  Against the
    dense form of the same optimizer (sparse_table_grads=False) on the same gathers: parameters and moments
    bit-identical after three steps; the gradient buffer is zero after a step and is handed out again for the next
    one; the scoring copy equals a fresh conversion of the updated table."""
    from generative_recommenders_amd import ops
    rows, n = 5003, 700
    g = torch.Generator().manual_seed(d)
    w0 = torch.randn(rows, d, generator=g) * 0.1
    small0 = torch.randn(17, 9, generator=g)

    def run(sparse):
        w = torch.nn.Parameter(w0.clone().to(DEV))
        small = torch.nn.Parameter(small0.clone().to(DEV))
        opt = HmxAdamW([w, small], lr=1e-2, betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0,
                       sparse_table_grads=sparse)
        prep = ops.PreparedTable(w.detach())
        bufs, norms = [], []
        for step in range(3):
            gg = torch.Generator().manual_seed(100 + step)
            ids = torch.randint(0, rows, (n,), generator=gg).to(DEV)
            ids2 = torch.randint(0, rows, (n // 2,), generator=gg).to(DEV)
            coef = torch.randn(n, d, generator=gg).to(DEV)
            opt.zero_grad(set_to_none=True)
            loss = (ops.gather_rows(w, ids, 0) * coef).sum() + ops.gather_rows(w, ids2, 0).square().sum() + small.sum()
            loss.backward()
            if sparse:
                st = ops.table_grad_state(w)
                touched = w.grad.abs().sum(1) > 0
                assert st is not None and bool((st.flags[touched] == 1).all())      # every written row is flagged
                assert int(st.flags.sum()) <= n + n // 2
                bufs.append(w.grad.data_ptr())
            opt.step()
            norms.append(opt.last_grad_norm.clone())
            if sparse:
                assert float(w.grad.abs().sum()) == 0.0 and int(st.flags.sum()) == 0 and st.clean
                assert prep.synced_version == w._version
                fresh = ops.PreparedTable(w.detach().clone())
                assert torch.equal(prep.bf16[:, :d], fresh.bf16[:, :d])
                assert abs(float(prep.max_row_norm) - float(fresh.max_row_norm)) <= 1e-5 * float(fresh.max_row_norm)
        return w, small, opt, bufs, norms

    ops.set_fused_table_grad(True)
    try:
        wa, sa, oa, bufs, na = run(True)
        wb, sb, ob, _, nb = run(False)
    finally:
        ops.set_fused_table_grad(False)
    assert len(set(bufs)) == 1                       # one gradient buffer, never re-filled
    for x, y in zip(na, nb):
        torch.testing.assert_close(x, y, rtol=2e-6, atol=0)
    if d == 64:      # 256 rows of 64 are one 16,384-element chunk of the flat form: same partial sums, same bits
        assert torch.equal(wa, wb) and torch.equal(sa, sb)
        assert torch.equal(oa.state[wa]["exp_avg"], ob.state[wb]["exp_avg"])
        assert torch.equal(oa.state[wa]["exp_avg_sq"], ob.state[wb]["exp_avg_sq"])
    else:            # the norm's partial sums are cut elsewhere: the clip factor differs in its last bits
        _close(wa, wb)
        _close(sa, sb)
        _close(oa.state[wa]["exp_avg"], ob.state[wb]["exp_avg"])
        _close(oa.state[wa]["exp_avg_sq"], ob.state[wb]["exp_avg_sq"])
