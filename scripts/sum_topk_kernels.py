import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('rescore','coarse_bf16','seed_thr','to_bf16')):
        tot+=float(r['TotalDurationNs']); print(f"  {float(r['TotalDurationNs'])/4e3:8.1f} us/call-set  {int(r['Calls'])//4} x {n[:70]}")
print('total per top-k call us', tot/4e3)
