"""Summarise a rocprofv3 kernel_stats.csv of a train-step run: ms and launches per step by kernel group.
usage: kernel_groups.py <dir-or-csv> <steps-in-run> [top-n]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
steps = float(sys.argv[2])
topn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = path if path.endswith(".csv") else max(glob.glob(os.path.join(path, "*", "*kernel_stats.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
groups = {"hmx sort": ["radix_"], "scan": ["scan_"], "linear": ["linear_kernel"], "wgrad": ["wgrad"], "fused block": ["ffn_", "pre_", "post_", "block_"],
          "loss (nce + its run sums)": ["nce_", "WeightedRows"],
          "optimizer + clip": ["adamw_kernel", "grad_sumsq", "grad_norm_finish", "multi_tensor", "fused_adam", "FusedAdam"], "rocprim sort": ["rocprim", "ROCPRIM"],
          "top-k (miner)": ["coarse", "rescore", "seed_thr", "to_bf16", "warm"], "gather": ["gather_rows"], "conv": ["conv_silu"],
          "scatter (plain rows)": ["PlainRows"], "run joins (both)": ["segment_join_kernel"],
          "pointwise hmx": ["rmsnorm", "binary_", "combine_", "silu_rope", "colsum", "sum_groups", "seq_index", "attn1q"]}
used = set()
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"total {tot:.2f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for g, keys in groups.items():
    t = c = 0
    for i, r in enumerate(rows):
        if i not in used and any(k in r["Name"] for k in keys):
            t += float(r["TotalDurationNs"]); c += int(r["Calls"]); used.add(i)
    if c:
        print(f"  {g:32s} {t / steps / 1e6:7.2f} ms {c / steps:7.1f} launches")
t = sum(float(r["TotalDurationNs"]) for i, r in enumerate(rows) if i not in used)
c = sum(int(r["Calls"]) for i, r in enumerate(rows) if i not in used)
print(f"  {'other (torch)':32s} {t / steps / 1e6:7.2f} ms {c / steps:7.1f} launches")
print("top kernels (ms/step, launches/step, avg us):")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:topn]:
    print(f"  {float(r['TotalDurationNs']) / steps / 1e6:7.3f} {int(r['Calls']) / steps:6.1f} {float(r['AverageNs']) / 1e3:9.1f}  {r['Name'][:110]}")
