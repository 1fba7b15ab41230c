"""min / mean / max of the sweep time per (pool, selection) of tools/ubench/vmm_spread output files"""
import json,sys,collections
for f in sys.argv[1:]:
    rows=[json.loads(l) for l in open(f) if l.startswith('{')]
    print(f.split('/')[-1], [ {k:v for k,v in r.items() if k in('create_s','handles')} for r in rows if 'create_s' in r])
    acc=collections.OrderedDict()
    for r in rows:
        if 'set' in r: acc.setdefault((r['pool'],r['set']),[]).append(r['full_ms'])
    for (p,s),v in acc.items(): print('  pool %3d %-18s n %d  min %.3f mean %.3f max %.3f'%(p,s,len(v),min(v),sum(v)/len(v),max(v)))
