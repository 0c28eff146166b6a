#!/usr/bin/env python
"""Price three-step tiles ("triples") BEFORE building them: for every chain of three
consecutive fusable stem steps of a tree, is there a tile that holds the row digits of the second AND third
contraction (k2r + k3r + X = 8 or 9 tile-row digits), do the intermediate (rewritten in place between steps 2 and
3) and the three small operands' bf16 limb planes fit 160 KB of LDS, are the item counts multiples of the eight
waves -- and what would the pair model, extended to three steps (matrix time of the three steps at the pairs'
bf16 x 3 rate, memory time of the big operand in and the LAST result out, the longer plus a fifth of the
shorter), make of it.  A dynamic programme over the chain then chooses among pairs and triples.

    python tools/price_triples.py sycamore_m20_fused [sycamore_m20_native ...]

Round 4, bf16 x 3 pricing: fused tree 33 feasible triples, 7 chosen, 21 ms less than pairs alone (of 217);
width-2^33 tree 3 chosen, -28 ms (of 379); headline tree 2 chosen, -8 ms (of 240) -- on trees that were refined
for PAIRS.  (Round 3 priced the same with fp32 products: -11.5 ms on the fused tree; with the cheaper products
the pairs are bound by their traffic, so taking one more round trip out is worth twice as much.)

MEASURED afterwards (the tiles were built: CTG_STEM_TRIPLES, profiles/r4_triples.txt): slower than pairs -- this pricing takes
one matrix rate for all shapes, and a 16-column stage runs at less than half of it (stem.TRIPLE_STAGE_RATE is what the planner
uses since).  The script is kept as the record of the estimate."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from cotengra_amd import plan as P, stem  # noqa: E402
from cotengra_amd.pathfind import MI355X_C64 as model  # noqa: E402


def price(name):
    tree=ca.tree_from_record(ca.load_network(os.path.join(ROOT,'tests','golden','trees','%s.json'%name)))
    sd=tree.size_dict
    plain=P.compile_tree(tree,"complex64",fuse=False)
    steps=plain.steps
    by_out={id(s.c):i for i,s in enumerate(steps) if s.kind==P.KIND_PAIR}
    def bits(inds): return stem._bits_of(inds, sd)
    def geometry3(s1,s2,s3):
        A,B1,B2,B3=s1.a,s1.b,s2.b,s3.b
        o1,o2,o3=set(s1.c.inds),set(s2.c.inds),set(s3.c.inds)
        a_bits=bits(A.inds); k1=bits([ix for ix in A.inds if ix not in o1]); n1=bits([ix for ix in B1.inds if ix in o1])
        k2=bits([ix for ix in s1.c.inds if ix not in o2]); n2=bits([ix for ix in B2.inds if ix in o2])
        k3=bits([ix for ix in s2.c.inds if ix not in o3]); n3=bits([ix for ix in B3.inds if ix in o3])
        if None in (a_bits,k1,n1,k2,n2,k3,n3): return None
        K1,N1,K2,N2,K3,N3=(1<<len(g) for g in (k1,n1,k2,n2,k3,n3))
        ok=(16,32,64,128)
        if any(x not in ok for x in (K1,N1,K2,N2,K3,N3)): return None
        aset=set(a_bits); k1s=set(k1); n1s=set(n1); n2s=set(n2)
        k2r=[b for b in k2 if b not in n1s]
        if any(b not in aset or b in k1s for b in k2r): return None
        k3a=[b for b in k3 if b in aset]
        if any(b in k1s or b in set(k2r) for b in k3a): return None
        if any((b not in aset) and (b not in n1s) and (b not in n2s) for b in k3): return None
        free=[b for b in a_bits if b not in k1s and b not in set(k2r) and b not in set(k3a)]
        free.sort(key=lambda b: stem._stride(A,b))
        cs1=max(1,N1//32)
        best=None
        for units in (8,16):
            nr1=5+int(math.log2(units//cs1)) if units>=cs1 else -1
            nx=nr1-len(k2r)-len(k3a)
            if nx<0 or nx>len(free): continue
            tm=nr1+len(n1); r2b=tm-len(k2)
            if r2b<5: continue
            t2=r2b+len(n2); r3b=t2-len(k3)
            if r3b<5: continue
            rows2,rows3=1<<r2b,1<<r3b
            mid=max(2*rows2*(K2+4)*4, 2*rows3*(K3+4)*4)
            # small operands: registers where they fit (96 floats), else LDS
            lds=mid+256+8*N3
            lds+=2*(3 if N1==16 else 2)*N1*((K1>>4)*48+8)
            for (K,N) in ((K2,N2),(K3,N3)):
                lds+=2*(3 if N==16 else 2)*N*((K>>3)*24+8)
            if lds>160*1024: continue
            it2=(rows2//32)*max(1,N2//32); it3=(rows3//32)*max(1,N3//32)
            if it2%8 or it3%8 or it2//8>2: continue
            cand=(units==16 and K1==16, -nr1, nr1, nx, it2, it3, lds)
            if best is None or cand>best: best=cand
        if best is None: return None
        _,_,nr1,nx,it2,it3,lds=best
        k1sorted=sorted(k1,key=lambda b: stem._stride(A,b))
        r1=sorted(k2r+k3a+free[:nx], key=lambda b: stem._stride(A,b))
        row_a=stem._table(r1,[stem._stride(A,b) for b in r1]); k_a=stem._table(k1sorted,[stem._stride(A,b) for b in k1sorted])
        task=np.sort((row_a[:32,None]+k_a[None,:16]).reshape(-1)); runs=np.flatnonzero(np.diff(task)!=1)
        run=8*int(runs[0]+1 if len(runs) else len(task))
        return dict(run=run,it2=it2,it3=it3,lds=lds,nr1=nr1)
    def tsec(macs, ea, ec, items_list, run):
        t_mfma=sum(8.0*m/(stem.FUSED_MFMA_RATE*1.6*min(1.0,it/8)) for m,it in zip(macs,items_list))
        t_mem=8.0*ea/stem.gather_rate(run)+8.0*ec/stem.FUSED_STORE_RATE
        return max(t_mfma,t_mem)+0.2*min(t_mfma,t_mem)
    big=[i for i,s in enumerate(steps) if s.kind==P.KIND_PAIR and s.a.size>=1<<24]
    unf={i:model.step_seconds(steps[i].macs,steps[i].elems_rw,steps[i].K,steps[i].N) for i in big}
    prev={i:by_out.get(id(steps[i].a)) for i in big}
    g2={}; g3={}
    for i in big:
        i1=prev[i]
        if i1 is None or i1 not in unf: continue
        s1,s2=steps[i1],steps[i]
        if stem._classify(s1,sd) is None or stem._classify(s2,sd) is None or s1.a.leaf>=0: continue
        geo=stem.geometry(sd,s1.a,s1.b,s2.b,s1.c.inds,s2.c.inds)
        if geo is not None:
            after=stem.pair_seconds(s1.macs,s2.macs,s1.a.size,s2.c.size,geo.items,geo.run_bytes)
            g2[i]=(i1,unf[i1]+unf[i]-after)
        i0=prev.get(i1)
        if i0 is None or i0 not in unf: continue
        s0=steps[i0]
        if stem._classify(s0,sd) is None or s0.a.leaf>=0: continue
        g=geometry3(s0,s1,s2)
        if g is not None:
            after=tsec((s0.macs,s1.macs,s2.macs), s0.a.size, s2.c.size, (8,g['it2'],g['it3']), g['run'])
            g3[i]=(i0,i1,unf[i0]+unf[i1]+unf[i]-after, after)
    # DP over the chain in step order
    order=sorted(big)
    best={}
    def get(i): return best.get(i,(0.0,()))
    last=None
    pos={i:n for n,i in enumerate(order)}
    tot_unf=sum(unf.values())
    # chain is linear through prev; do DP by index
    for i in order:
        cands=[get(prev[i]) if prev[i] in pos else (0.0,())]
        if i in g2:
            i1,g=g2[i]; b=get(prev[i1]) if prev.get(i1) in pos else (0.0,())
            cands.append((b[0]+g,b[1]+(('P',i1,i),)))
        if i in g3:
            i0,i1,g,_=g3[i]; b=get(prev[i0]) if prev.get(i0) in pos else (0.0,())
            cands.append((b[0]+g,b[1]+(('T',i0,i1,i),)))
        best[i]=max(cands,key=lambda c:c[0])
    end=max(best.values(), key=lambda c:c[0])
    # pairs only
    bestp={}
    for i in order:
        cands=[bestp.get(prev[i],(0.0,()))]
        if i in g2:
            i1,g=g2[i]; b=bestp.get(prev.get(i1),(0.0,()))
            cands.append((b[0]+g,b[1]+(('P',i1,i),)))
        bestp[i]=max(cands,key=lambda c:c[0])
    endp=max(bestp.values(), key=lambda c:c[0])
    print(name,'unfused big steps %.1f ms'%(tot_unf*1e3),'pairs-only gain %.1f ms'%(endp[0]*1e3),'with triples gain %.1f ms'%(end[0]*1e3), 'triples feasible',len(g3))
    print([c[0] for c in end[1]].count('T'),'triples',[c[0] for c in end[1]].count('P'),'pairs')
    for c in end[1]:
        if c[0]=='T':
            s=[steps[j] for j in c[1:]]
            print('  T',[(x.K,x.N) for x in s],'R=2^%.0f'%math.log2(s[0].R),'after %.1f ms'%(g3[c[3]][3]*1e3),'unf %.1f'%(sum(unf[j] for j in c[1:])*1e3))


for name in sys.argv[1:] or ['sycamore_m20_fused']:
    price(name)
