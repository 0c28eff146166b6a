import json, sys
rows = json.load(open(sys.argv[1]))
tot = sum(r['ms'] for r in rows)
print('total ms', tot)
P, BW = 157.3e12, 8e12
roof = sum(max(8*r['macs']/P, r['bytes']/BW) for r in rows)*1e3
print('roofline ms', roof)
rows.sort(key=lambda r:-r['ms'])
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 25]:
    fl = 8*r['macs']; rf = max(fl/P, r['bytes']/BW)*1e3
    print(r['step'], r['kernel'], 'R',r['R'],'K',r['K'],'N',r['N'], 'ms=%.2f'%r['ms'], 'TF=%.1f'%(fl/r['ms']/1e9), 'GB/s=%.0f'%(r['bytes']/r['ms']/1e6), '%.1f%%'%(100*r['ms']/tot), 'roof=%.2f eff=%.0f%%'%(rf, 100*rf/r['ms']))
print('valu total', sum(r['ms'] for r in rows if r['kernel']=='valu'), 'mfma total', sum(r['ms'] for r in rows if r['kernel']=='mfma'))
small = [r for r in rows if r['ms']<0.02]; print('n tiny', len(small), 'sum', sum(r['ms'] for r in small))
