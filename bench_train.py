"""Train-step workload of bench.py: hydra3 forward + backward (+ clip + AdamW) on synthetic ids,
batch 1024 x seq 100, d 64, 1 M users, 4 M items (BASELINE.json configs[1]).

The step reproduces the call sequence of main.py:684-713 with the drop-in modules: zero_grad, Hydra.forward
(encoder, miner top-50 over the whole item table, InfoNCE with the 153-way hard-negative gather), backward
(scatter-add into dense table gradients), clip_grad_norm_(1.0), AdamW(lr 1e-3, betas (0.9, 0.98), wd 0.1).
Differences from the reference's loop, stated here: compute is fp32 with bf16 MFMA inside the projection
kernels instead of fp16 autocast + GradScaler; the loss is one fused kernel over all masked positions
(the reference's query chunking only bounds its memory; the fused loss kernel needs none).
"""
import json
import os
import time

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0


def _sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def cpu_baseline_train(args, seconds_cap=40.0):
    """The CPU oracle (fixture-verified restatement of the reference) on this box's host cores: one forward +
    backward of the same model (1 M users, 4 M items, d 64, L 100) on a bounded batch sized to about 15 s of CPU work."""
    from oracle import hydra_oracle as O
    from generative_recommenders_amd import synth
    threads = torch.get_num_threads()
    model = O.build_model(args.users, args.items, d=args.dim, L=args.seq_len, dropout=0.2, seed=1).train()

    def one(B, seed):
        bt = synth.train_batch(B, args.seq_len, args.users, args.items, "cpu", seed=seed)
        feat = [[] for _ in range(B)]
        t0 = time.time()
        loss = model(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat, bt["ts"],
                     bt["act"], bt["dwell"])
        loss.backward()
        return time.time() - t0

    # a 4-sequence pass sizes the sample so that it is about 15 s of CPU work (at most 256 sequences)
    t4 = one(4, 2)
    B = int(max(4, min(256, 15.0 / max(t4 / 4, 1e-3))))
    model.zero_grad(set_to_none=True)
    dt = one(B, 3)
    # BASELINE.json configs[0]: encoder forward only (log2feats, eval mode), batch 128, on the same host cores
    model.eval()
    Be = max(4, min(128, int(128 * (100 * 64 * 64) / (args.seq_len * args.dim * args.dim))))   # 128 at configs[0]
    be = synth.train_batch(Be, args.seq_len, args.users, args.items, "cpu", seed=3)
    t1 = time.time()
    with torch.no_grad():
        model.log2feats(be["seq"], [[] for _ in range(Be)], be["tok"], be["ts"], be["act"], be["dwell"])
    de = time.time() - t1
    return {"value": B / dt, "unit": "user seqs/s", "cores": threads, "kind": "port",
            "sample": f"one fwd+bwd of {B} sequences x {args.seq_len}, same model and tables, PyTorch CPU oracle on "
                      f"{threads} threads ({dt:.1f} s)",
            "encoder_fwd": {"value": Be / de, "unit": "user seqs/s",
                            "sample": f"log2feats forward of {Be} sequences x {args.seq_len}, eval mode ({de:.1f} s)"
                                      + (" = BASELINE.json configs[0]" if (Be, args.seq_len, args.dim) == (128, 100, 64) else "")}}


def run_train(args, world, rank, with_cpu_baseline=True):
    from generative_recommenders_amd import hmx, synth
    from generative_recommenders_amd import hydra as H
    from generative_recommenders_amd import ops as ops_mod
    dev = torch.device("cuda", torch.cuda.current_device())
    B, L, d = args.batch, args.seq_len, args.dim
    model = H.build_model(args.users, args.items, d=d, L=L, dropout=0.2, device=dev, seed=1234)
    model.train()
    ref_terms = bool(getattr(args, "reference_terms", False))
    model.enchanced_loss.evaluate_all_terms = ref_terms
    ops_mod.set_fused_table_grad(True)     # table gradients accumulate straight into .grad (same sums, fewer passes)
    from generative_recommenders_amd import enc_ops
    enc_ops.set_direct_param_grad(True)    # weight gradients of the projections: batched launches adding into .grad
    # main.py:640 builds torch.optim.AdamW(lr, betas=(0.9, 0.98), weight_decay=0.1) and main.py:709-713 clips the gradient
    # norm to 1.0 before its step.  Default here: the same update and the same clip on the HIP kernels (one pass for
    # the norm, one for the update with the clip factor applied inside; optim.HmxAdamW, tested against the torch pair).
    # --torch-adamw: torch's fused single-kernel AdamW after torch's clip_grad_norm_; --foreach-adamw: torch's default
    # multi-tensor form.
    which_opt = "foreach" if getattr(args, "foreach_adamw", False) else ("torch" if getattr(args, "torch_adamw", False) else "hmx")
    if which_opt == "hmx":
        from generative_recommenders_amd.optim import HmxAdamW
        opt = HmxAdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.1, max_grad_norm=1.0)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.1, fused=which_opt == "torch")
    params = list(model.parameters())
    bt = synth.train_batch(B, L, args.users, args.items, dev, seed=99 + rank)
    feat = [[] for _ in range(B)]
    sync_grads = None
    if world > 1:
        from generative_recommenders_amd.parallel import GradientSync
        sync_grads = GradientSync(model)

    def step(timers=None):
        opt.zero_grad(set_to_none=True)
        loss = model(bt["seq"], bt["pos"], bt["neg"], bt["tok"], bt["next_tok"], None, feat, feat, feat, bt["ts"],
                     bt["act"], bt["dwell"])
        loss.backward()
        if timers is not None:
            timers.append(torch.cuda.Event(enable_timing=True))
            timers[-1].record()
        if sync_grads is not None:
            sync_grads()
        if which_opt != "hmx":
            torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()
        return loss

    for _ in range(max(args.warmup, 1)):
        step()
    TAGS = ("gather_rows", "scatter_segment_sum", "topk_coarse_bf16", "proj", "scan_fwd", "scan_bwd", "nce_fwd", "nce_bwd",
            "optimizer", "input_assembly")
    # Per-kernel durations of the other tagged kernels: a few extra steps with every bracket on, outside the timed
    # region (each bracket is two event records on the stream; some eighty of them per step are not free).  The timed
    # region itself brackets the roofline kernel (the backward scan) alone.
    hmx.profile_enable(True)
    for tag in TAGS:
        hmx.profile_collect(tag)
    n_side = min(args.steps, 5)
    for _ in range(n_side):
        step()
    _sync(world)
    side_spans = {tag: hmx.profile_collect(tag) for tag in TAGS}
    hmx.profile_enable(True, only="scan_bwd")
    _sync(world)
    starts, mids = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        starts.append(torch.cuda.Event(enable_timing=True))
        starts[-1].record()
        loss = step(mids)
    _sync(world)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # everything but the backward scan scaled from the side steps to the timed step count
    spans = {tag: (ms * args.steps / n_side, n * args.steps // n_side) for tag, (ms, n) in side_spans.items()}
    spans["scan_bwd"] = hmx.profile_collect("scan_bwd")
    hmx.profile_enable(False)
    fwd_bwd_ms = sum(s.elapsed_time(m) for s, m in zip(starts, mids)) / args.steps
    step_s = elapsed / args.steps

    # Row gathers of one step: the pos_seqs item rows (B*L ids; 520 B per id at d = 64: 8 B index + row read + row
    # write).  The sequence's own rows (routed item / user gather, position, time, action, dwell) are read inside the
    # input-assembly kernel and the hard negatives inside the fused loss kernel.
    n_ids = B * L
    # Input assembly, forward launch: per token five 8-byte indices, one ds-float table row (the position and small-table
    # rows repeat and come from cache), the d-float output and, in training, the ds-float row sum kept for the weight
    # gradient.  Backward launch: reads dout, writes g, gs (d floats each) and de (ds floats).
    ds_w = model.shared_embeddings.item_emb.embedding_dim
    ia_ms, ia_n = spans["input_assembly"]
    ia_bytes = B * L * ((40 + 4 * (2 * ds_w + d)) + 4 * (3 * d + ds_w))
    g_ms, g_n = spans["gather_rows"]
    c_ms, c_n = spans["topk_coarse_bf16"]
    miner_tf = 2.0 * B * (args.items + 1) * d * args.steps / (c_ms * 1e-3) / 1e12 if c_ms > 0 else 0.0
    s_ms, s_n = spans["scatter_segment_sum"]

    # Encoder projections (everything that runs on the matrix cores: the fused block halves, the stand-alone
    # projections and the weight gradients).  FLOPs per token and block, forward: 5 d x d products + per head in_proj
    # (hd -> 2 ED), deltaBC (ED -> R + 2 N), out_proj (ED -> hd); the dt projection (R -> ED) runs inside the scan
    # kernel and is left out of both sides.  Backward = 2 x forward.  Blocks: first + 2 + long-term at L, mid-term
    # at min(25, L), short-term at min(10, L) (hydra3.py:460-548, 618-681).
    H, R, N = 2, 16, 16
    hd = d // H
    ED = 4 * hd
    proj_flops_tok = 5 * 2 * d * d + H * 2 * (hd * 2 * ED + ED * (R + 2 * N) + ED * hd)
    block_tokens = B * (4 * L + min(25, L) + min(10, L))
    proj_flops_step = 3.0 * proj_flops_tok * block_tokens
    pr_ms, pr_n = spans["proj"]
    proj_tf = proj_flops_step * args.steps / (pr_ms * 1e-3) / 1e12 if pr_ms > 0 else 0.0

    # The step's longest kernel: the backward selective scan (recomputes the states from checkpoints every 4 steps).
    # Algorithmic bytes of a launch over (B, L') with C = H ED channels: reads xc, z, dyg (C floats per token each),
    # dbc (H (R + 2 N)), the state checkpoints; writes dxc, dz, the per-group dbc gradient and the per-sequence
    # parameter partials.
    C = H * ED
    gph = int(hmx.lib().hmx_scan_groups_per_head(ED))
    scan_bytes = 0
    for Lb in (L, L, L, L, min(25, L), min(10, L)):
        Tb = B * Lb
        scan_bytes += 4 * (3 * Tb * C + Tb * H * (R + 2 * N) + int(hmx.lib().hmx_scan_state_floats(B, Lb, C)) +
                           2 * Tb * C + Tb * H * gph * (R + 2 * N) + B * C * 34)
    sb_ms, sb_n = spans["scan_bwd"]
    scan_gbs = scan_bytes * args.steps / (sb_ms * 1e-3) / 1e9 if sb_ms > 0 else 0.0
    scan_traffic = None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_scan_bwd_pmc.json")
    if (B, L, d) == (1024, 100, 64) and os.path.exists(pmc):
        with open(pmc) as f:
            scan_traffic = json.load(f).get("hbm_bytes_per_launch")

    # HBM roofline of the gather kernel: the step's own gathers are 102,400 ids per launch (26 MB, launch-latency
    # sized), so the kernel is also timed on 8,388,608 uniform random ids over the same item table -- with fp32 output
    # (520 B per id at d = 64: index + row read + row write) and with bf16 output (392 B per id moved).
    from generative_recommenders_amd import ops
    n_probe = 1 << 23
    table = model.shared_embeddings.item_emb.weight.detach()
    pid = torch.randint(1, args.items + 1, (n_probe,), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    probes = {}
    for tag, fn in (("gather_rows", ops.gather_rows), ("gather_rows_bf16", ops.gather_rows_bf16)):
        fn(table, pid)
        hmx.profile_enable(True)
        hmx.profile_collect(tag)
        for _ in range(5):
            fn(table, pid)
        torch.cuda.synchronize()
        probes[tag] = hmx.profile_collect(tag)
        hmx.profile_enable(False)
    p_ms, p_n = probes["gather_rows"]
    b_ms, b_n = probes["gather_rows_bf16"]
    probe_bytes = n_probe * (8 + 2 * d * 4)
    probe_bytes_bf16 = n_probe * (8 + d * 4 + d * 2)
    traffic = None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_gather_pmc.json")
    if d == 64 and args.items == 4_000_000 and os.path.exists(pmc):    # counters were collected on this exact launch
        with open(pmc) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    gather_gbs = probe_bytes * p_n / (p_ms * 1e-3) / 1e9 if p_ms > 0 else 0.0
    gather_bf16_gbs = probe_bytes_bf16 * b_n / (b_ms * 1e-3) / 1e9 if b_ms > 0 else 0.0

    # The same probes on skewed ids: Zipf(1.05) over the item ranks (continuous inverse-CDF draw), ranks spread over
    # the table by a fixed multiplicative hash so that popular rows are not neighbours.  Repeated rows come from
    # cache in the gather and collapse into long runs in the scatter-add, which is what the run sums' dedup is for.
    gz = torch.Generator(device=dev).manual_seed(6)
    u = torch.rand(n_probe, device=dev, generator=gz, dtype=torch.float64)
    s_exp, n_items = 1.05, float(args.items)
    ranks = ((n_items ** (1.0 - s_exp) - 1.0) * u + 1.0) ** (1.0 / (1.0 - s_exp))
    zid = (ranks.long().clamp_(1, args.items) * 2654435761 % args.items + 1)
    del u, ranks
    ops.gather_rows(table, zid)
    hmx.profile_enable(True)
    hmx.profile_collect("gather_rows")
    for _ in range(5):
        ops.gather_rows(table, zid)
    torch.cuda.synchronize()
    z_ms, z_n = hmx.profile_collect("gather_rows")
    hmx.profile_enable(False)
    gather_zipf_gbs = probe_bytes * z_n / (z_ms * 1e-3) / 1e9 if z_ms > 0 else 0.0

    # Scatter-add of n_probe gradient rows into a dense (items + 1, d) gradient: whole operation (sort + run sums +
    # joins), event-timed on the launch stream.  Algorithmic bytes: n (8 + 4 d) + 2 u 4 d, u = distinct rows.
    scatter = {}
    grows = torch.randn(n_probe, d, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    dense = torch.zeros(args.items + 1, d, device=dev)
    for name, idv in (("uniform ids", pid), ("Zipf(1.05) ids", zid)):
        uniq = int(torch.unique(idv).numel())
        ops.scatter_add_rows(grows, idv, args.items + 1, 0, out=dense)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.scatter_add_rows(grows, idv, args.items + 1, 0, out=dense)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        nbytes = n_probe * (8 + 4 * d) + 2 * uniq * 4 * d
        scatter[name] = {"ms_per_call": ms, "distinct_rows": uniq, "algorithmic_bytes": nbytes,
                         "achieved_GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_8000": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del pid, zid, grows, dense
    out = {
        "metric": "user seqs/s hydra3 fwd+bwd", "value": B * world / step_s, "unit": "user seqs/s",
        "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": step_s * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"hydra3 train step fwd+bwd with embedding scatter-add, batch {B} x seq {L}, d {d}" +
                               (" (configs[1])" if (B, L, d) == (1024, 100, 64) else
                                " (configs[4], per GPU)" if (L, d) == (512, 128) else ""),
                   "batch_per_gpu": B, "seq_len": L, "d": d, "users": args.users, "items": args.items,
                   "includes": "zero_grad, forward (encoder + miner top-50 + fused InfoNCE), backward, "
                               "clip_grad_norm + AdamW (" + {"hmx": "optim.HmxAdamW: HIP norm and update kernels, clip factor applied inside the update",
                                                          "torch": "torch clip_grad_norm_ + torch.optim.AdamW(fused=True)",
                                                          "foreach": "torch clip_grad_norm_ + torch.optim.AdamW"}[which_opt] +
                               "); table and projection-weight gradients "
                               "accumulated directly into .grad. " +
                               ("Evaluated like the reference: the neg_seqs gather + projection, the ranking term and the "
                                "diversity term, multiplied by their default weights 0.0 (hydra3.py:694-696,711,1139-1149) "
                                "and back-propagated (--reference-terms)" if ref_terms else
                                "Not evaluated (run with --reference-terms to include them): "
                                "the neg_seqs gather + projection and the diversity term, which the reference computes "
                                "and multiplies by its default weights 0.0 (hydra3.py:694-696,711,1146-1149): two row "
                                "gathers per step here (seq, pos), three in the reference"),
                   "parallelism": f"dp{world}" if world > 1 else "single GPU"},
        "fwd_bwd_ms": fwd_bwd_ms, "fwd_bwd_seqs_per_s": B * world / (fwd_bwd_ms * 1e-3), "loss": float(loss.item()),
        "roofline": {"bound": "hbm", "achieved": scan_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": scan_gbs / HBM_PEAK_GBS, "traffic": scan_traffic,
                     "traffic_unit": "HBM bytes per launch, mean over the step's six launches (FETCH_SIZE x 2 + WRITE_SIZE, "
                                     "profiles/r03_scan_bwd_pmc.json)",
                     "kernel": "scan_bwd_kernel (the step's longest kernel; its pace is set by the exp / fma work of the "
                               "recomputed recurrence, not by memory)",
                     "launches": sb_n, "avg_launch_ms": sb_ms / max(sb_n, 1),
                     "algorithmic_bytes_per_launch": scan_bytes / 6.0},
        "encoder": {"projection_flops_per_step": proj_flops_step,
                    "projection_kernels_ms_per_step": pr_ms / args.steps, "projection_launches_per_step": pr_n / args.steps,
                    "mfma_util_projections": {"achieved_tflops": proj_tf, "frac_of_2500": proj_tf / MFMA_BF16_PEAK_TFLOPS,
                                              "note": "useful FLOPs / time of the kernels that hold the projections (fused "
                                                      "block halves incl. their norm / gate / residual work, linear_kernel, "
                                                      "wgrad_kernel); every product is three bf16 MFMA passes (hi*hi, "
                                                      "lo*hi, hi*lo) or fp32 MFMA in the weight gradients, so the matrix "
                                                      "pipes do 3x these FLOPs"},
                    "scan_fwd_ms_per_step": spans["scan_fwd"][0] / args.steps,
                    "scan_bwd_ms_per_step": sb_ms / args.steps,
                    "loss_fwd_ms_per_step": spans["nce_fwd"][0] / args.steps,
                    "loss_bwd_ms_per_step": spans["nce_bwd"][0] / args.steps,
                    "optimizer_kernels_ms_per_step": spans["optimizer"][0] / args.steps},
        "other_kernels": {
            "gather_rows_kernel probe (hbm)": {
                "achieved_GBps": gather_gbs, "frac_of_8000": gather_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, profiles/r01_gather_pmc.json)",
                "launch": f"{n_probe} uniform random ids over the {args.items + 1} x {d} item table",
                "launches": p_n, "avg_launch_ms": p_ms / max(p_n, 1), "algorithmic_bytes_per_launch": probe_bytes},
            "gather_rows_bf16_kernel probe (hbm)": {
                "achieved_GBps_moved": gather_bf16_gbs,
                "avg_launch_ms": b_ms / max(b_n, 1), "bytes_moved_per_launch": probe_bytes_bf16,
                "note": "same ids; the output row is bf16, so a launch moves 392 B per id at d = 64 instead of 520"},
            "gather_rows_kernel probe, Zipf(1.05) ids (hbm + cache)": {
                "achieved_GBps_at_520_B_per_id": gather_zipf_gbs, "avg_launch_ms": z_ms / max(z_n, 1),
                "note": "same launch with skewed ids: repeated rows are served by L2 / Infinity Cache"},
            f"scatter_add_rows probe, {n_probe} rows of d {d} into the dense item gradient (sort + run sums, hbm)": scatter,
            "gather_rows_kernel inside the step": {"ids_per_step": n_ids, "launches": g_n,
                                                   "ms_per_step": g_ms / args.steps},
            "input_assembly kernels (forward + backward, hbm)": {
                "ms_per_step": ia_ms / args.steps, "launches": ia_n,
                "algorithmic_bytes_per_step": ia_bytes,
                "achieved_GBps": ia_bytes * args.steps / (ia_ms * 1e-3) / 1e9 if ia_ms > 0 else 0.0,
                "note": "routed item / user row + position row -> hydra_projection on the matrix cores -> + time / action / "
                        "dwell rows, mask; the gathered rows are not written (training keeps their sum for the weight "
                        "gradient)"},
            "coarse scoring kernels (miner, mfma)": {"achieved_tflops": miner_tf, "frac_of_2500": miner_tf / 2500.0,
                                                     "ms_per_step": c_ms / args.steps},
            "embedding scatter-add (sort + run sums, hbm)": {"ms_per_step": s_ms / args.steps, "launches": s_n}},
    }
    if with_cpu_baseline and rank == 0 and world == 1 and not args.no_cpu_baseline:
        del model, opt
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline_train(args)
    return out
