#!/usr/bin/env python
"""Polish a sliced m20 tree for the executor of round 3, which runs consecutive stem steps
fused (cotengra_amd/stem.py): ``pathfind.refine`` with the fused-pair machine model as the
first objective of the sweep (``pathfind.MI355X_C64_FUSED``; candidates are ranked by
``pathfind.modelled_seconds``, which prices the plan the executor really builds -- fused
pairs with the calibrated pair model).  Host tools of this package only.

    python tests/golden/gen/refine_fused.py tests/golden/trees/sycamore_m20_native.json OUT.json [log2 width] [arena GiB]

(defaults: width 2^32, arena 150 GiB; ``33 170`` gives sycamore_m20_w33_fused.json: 68 GB tensors,
a 161 GiB arena -- what 288 GB of HBM allow)
"""
import json, sys, time, math
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..'))
import cotengra_amd as ca
from cotengra_amd import pathfind as pf
src = sys.argv[1]; dst = sys.argv[2]
width = 2 ** int(sys.argv[3]) if len(sys.argv) > 3 else 2 ** 32
arena_gib = int(sys.argv[4]) if len(sys.argv) > 4 else 150
rec = ca.load_network(src)
tree = ca.tree_from_record(rec)
t0 = time.time()
base = pf.modelled_seconds(tree)[0]
print('start: %.1f ms/slice x 2^%.0f = %.3e s' % (base*1e3, math.log2(tree.nslices), base*tree.nslices), flush=True)
def prog(rnd, obj, sz, t, v):
    print('round', rnd, obj if isinstance(obj, str) else 'fused-model', sz, '2^%.0f slices' % math.log2(t.nslices), '10^%.3f MACs' % t.contraction_cost(log=10), '%.3e s' % v, '(%.0fs)' % (time.time()-t0), flush=True)
new = pf.refine(tree, objectives=(pf.MI355X_C64_FUSED, "time", "combo-64", "combo-32", "combo-128"), subtree_sizes=(8, 10, 12, 14), progress=prog,
                max_width=width, max_arena_bytes=arena_gib * 2**30)
secs, arena = pf.modelled_seconds(new)
print('final: %.1f ms/slice, 2^%.0f slices, %.3e s, arena %.0f GiB' % (secs*1e3, math.log2(new.nslices), secs*new.nslices, arena/2**30))
out = {k: rec[k] for k in ("source", "inputs", "output", "size_dict") if k in rec}
out["path"] = [list(p) for p in new.get_path()]
out["sliced_inds"] = list(new.sliced_inds)
out["search"] = {"optimizer": "pathfind.refine of %s under the fused-pair model (round 3): objectives MI355X_C64_FUSED, time, combo-64/32/128; subtree sizes 8-14; width <= 2^%d, arena <= %d GiB" % (src.split('/')[-1], int(math.log2(width)), arena_gib), "seconds": round(time.time()-t0)}
out["stats"] = {"nslices_log2": math.log2(new.nslices), "contraction_cost_log10": new.contraction_cost(log=10), "cost_per_slice": new.contraction_cost() // new.nslices, "max_size_log2": new.max_size(log=2), "model_ms_per_slice": secs*1e3, "arena_gib": arena/2**30}
json.dump(out, open(dst, 'w'), ensure_ascii=False)
