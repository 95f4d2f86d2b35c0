"""Summarise a rocprofv3 run of the train bench: ms and launches per step by kernel group.
With a kernel_trace.csv next to the stats, only whole steps count (a step ends at its adamw_kernel launch; whatever
precedes the first one -- model set-up, data synthesis, the first step -- is left out, and so are the probes after the
last one).  With only a kernel_stats.csv everything in the run is divided by <steps-in-run>.
usage: kernel_groups.py <dir-or-csv> <steps-in-run> [top-n]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
steps = float(sys.argv[2])
topn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
groups = {"input assembly": ["input_assembly"], "hmx sort": ["radix_"], "scan": ["scan_"], "linear": ["linear_kernel"], "wgrad": ["wgrad"],
          "fused block": ["ffn_", "pre_", "post_", "block_"], "loss (nce + its run sums)": ["nce_", "WeightedRows", "inv_norm"],
          "optimizer + clip": ["adamw_kernel", "adamw_table", "table_sumsq", "grad_sumsq", "grad_norm_finish", "multi_tensor", "fused_adam", "FusedAdam"],
          "rocprim (nonzero, scans)": ["rocprim", "ROCPRIM"], "top-k (miner)": ["coarse", "rescore", "seed_thr", "to_bf16", "warm", "group_max", "group_threshold"],
          "gather": ["gather_rows"], "conv": ["conv_silu"], "scatter (plain rows)": ["PlainRows"],
          "run joins (both)": ["segment_join_kernel"],
          "pointwise hmx": ["rmsnorm", "binary_", "combine_", "silu_rope", "colsum", "sum_groups", "seq_index", "attn1q"]}

rows = None
if not path.endswith(".csv"):
    traces = glob.glob(os.path.join(path, "*", "*kernel_trace.csv"))
    if traces:
        tr = sorted(csv.DictReader(open(max(traces, key=os.path.getmtime))), key=lambda r: int(r["Start_Timestamp"]))
        ends = [i for i, r in enumerate(tr) if "adamw_kernel" in r["Kernel_Name"]]
        if len(ends) >= 2:
            steps = len(ends) - 1
            agg = defaultdict(lambda: [0.0, 0])
            for r in tr[ends[0] + 1:ends[-1] + 1]:
                a = agg[r["Kernel_Name"]]
                a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                a[1] += 1
            rows = [{"Name": k, "TotalDurationNs": v[0], "Calls": v[1], "AverageNs": v[0] / v[1]} for k, v in agg.items()]
            print(f"(from the kernel trace: {steps} whole steps after the first)")
if rows is None:
    f = path if path.endswith(".csv") else max(glob.glob(os.path.join(path, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
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
