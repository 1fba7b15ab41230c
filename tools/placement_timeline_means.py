"""first-half -> second-half means per state of tools/ubench/placement_timeline output files"""
import json,sys,collections
for f in sys.argv[1:]:
    acc=collections.defaultdict(list); meta=[]
    for l in open(f):
        if not l.startswith('{'): continue
        r=json.loads(l)
        if 'round' in r:
            for k,v in r.items():
                if k not in('round','t'): acc[k].append(v)
        elif 'alloc_s' in r or 'create_s' in r: meta.append(round(r.get('alloc_s',r.get('create_s')),3))
    def st(v):
        h=len(v)//2
        return f"{sum(v[:h])/max(h,1):.3f}->{sum(v[h:])/max(len(v)-h,1):.3f}"
    print(f.split('/')[-1], 'alloc/create_s', meta, {k:st(v) for k,v in acc.items()})
