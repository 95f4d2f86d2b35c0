"""Print per-kernel sums of rocprofv3 --pmc counter CSVs found under a directory."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if pat and pat not in name:
            continue
        acc[name[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:36s} {x:.6g}")
