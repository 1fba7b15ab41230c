"""Sum of each counter over the fused_iter_kernel dispatches of a rocprofv3 counter_collection.csv (millions, dispatches)"""
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(float); n=collections.Counter()
for r in rows:
    if 'fused_iter' not in r['Kernel_Name']: continue
    acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print({k:(round(v/1e6,1), n[k]) for k,v in acc.items()})
