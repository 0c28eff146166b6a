import re,collections,sys
L=open(sys.argv[1] if len(sys.argv)>1 else '/tmp/ctg_stem-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
labels=[i for i,l in enumerate(L) if re.match(r'\.LBB\d+_\d+:',l)]
def loops():
    out=[]
    for i in labels:
        name=L[i].split(':')[0]
        for j in range(i+1,len(L)):
            if re.search(r's_c?branch\w*\s+'+re.escape(name)+r'\b',L[j]):
                out.append((i,j)); break
            if L[j].startswith('.Lfunc_end'): break
    return out
lp=loops()
outer=[(i,j) for i,j in lp if sum(1 for k in range(i,j) if 's_barrier' in L[k])>=2]
def mix(i,j,skip=()):
    c=collections.Counter()
    for k in range(i,j+1):
        if any(a<=k<=b for a,b in skip): continue
        l=L[k].split(';')[0].strip()
        if not l or l.startswith('.'): continue
        op=l.split()[0]
        if op.startswith('v_mfma'): c['mfma']+=1
        elif op.startswith('v_'): c['valu']+=1; c['  '+op]+=1
        elif op.startswith('ds_'): c['lds']+=1; c['  '+op]+=1
        elif op.startswith('s_waitcnt'): c['wait']+=1
        elif op.startswith('s_nop'): c['nop']+=1
        elif op.startswith('s_barrier'): c['barrier']+=1
        elif op.startswith('s_'): c['salu']+=1
        elif op.startswith(('global_','buffer_','flat_','scratch_')): c['vmem']+=1; c['  '+op]+=1
    return c
for (i,j) in outer[:1]:
    inner=[(a,b) for a,b in lp if i<a and b<j]
    print('outer loop lines',i,j,'inner loops',inner)
    co=mix(i,j,inner)
    print('OUTER (excluding inner loops):'); 
    for k,v in sorted(co.items(), key=lambda kv:(kv[0].startswith(' '),-kv[1])): print('  ',v,k)
    for a,b in inner:
        ci=mix(a,b)
        print('INNER',a,b,{k:v for k,v in ci.items() if not k.startswith(' ')})
