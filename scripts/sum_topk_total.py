import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
tot=0
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('rescore','coarse_bf16','seed_thr','to_bf16')): tot+=float(r['TotalDurationNs'])
print(sys.argv[2], 'total per top-k call us', round(tot/4e3,1))
