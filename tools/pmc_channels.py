"""Per-state summary of per-channel TCC counters (rocprofv3 --output-format json) of tools/ubench/placement_timeline: usage pmc_channels.py tl_results.json malloc0,malloc1,vmm1024"""
import json,collections,sys
import numpy as np
d=json.load(open(sys.argv[1]))
names=sys.argv[2].split(',')
r=d['rocprofiler-sdk-tool'][0]
cn={c['id']['handle']:c['name'] for c in r['counters']}
ks={k['kernel_id']:k.get('formatted_kernel_name',k.get('kernel_name')) for k in r['kernel_symbols']}
fused=[]
for x in r['callback_records']['counter_collection']:
    di=x['dispatch_data']['dispatch_info']
    if 'fused_iter' not in ks.get(di['kernel_id'],''): continue
    vals=collections.defaultdict(list)
    for q in x['records']: vals[cn[q['counter_id']['handle']]].append(q['value'])
    dur=(x['dispatch_data']['end_timestamp']-x['dispatch_data']['start_timestamp'])/1e6
    fused.append((dur,{k:np.array(v) for k,v in vals.items()}))
n=len(names)
print(len(fused),'fused dispatches')
agg=collections.defaultdict(list)
for j,(dur,v) in enumerate(fused):
    if j<n: continue
    agg[names[((j-n)//4)%n]].append((dur,v))
for s in names:
    a=agg[s]
    if not a: continue
    print(s,'n',len(a),'dur %.3f'%np.mean([x[0] for x in a]))
    for c in a[0][1]:
        m=np.mean([x[1][c] for x in a],axis=0)
        if len(m)==128:
            g=m.reshape(8,16)
            print('   %-36s sum %.4g  max/mean %.3f min/mean %.3f | per XCC/mean:'%(c,m.sum(),m.max()/m.mean(),m.min()/m.mean()), np.round(g.mean(axis=1)/m.mean(),3), '| per inst/mean:', np.round(g.mean(axis=0)/m.mean(),2))
        else: print('   ',c,m.sum())
    if 'TCC_EA0_RDREQ_LEVEL' in a[0][1]:
        lat=np.mean([x[1]['TCC_EA0_RDREQ_LEVEL']/np.maximum(x[1]['TCC_EA0_RDREQ'],1) for x in a],axis=0)
        print('   read latency (cycles): mean %.0f min %.0f max %.0f'%(lat.mean(),lat.min(),lat.max()))
