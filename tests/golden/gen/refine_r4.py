#!/usr/bin/env python
"""Round 4: polish a tree once more under the model of the executor as it is now -- fused pairs AND
single stem steps in the bf16 x 3 arithmetic (the default; ``stem.pair_seconds`` / ``single_seconds``
with the pairs' matrix rate x ``stem.BF16X3_SPEEDUP``), everything else as refine_bf3.py.  Host tools
of this package only; 20-30 minutes.

    python tests/golden/gen/refine_r4.py SRC.json OUT.json [log2 max width = 32] [max arena GiB = 90]
"""
import json, math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..'))
import cotengra_amd as ca
os.environ.pop("CTG_STEM_BF16X3", None)          # (the default arithmetic: bf16 x 3)
from cotengra_amd import pathfind as pf
src, dst = sys.argv[1], sys.argv[2]
width = int(sys.argv[3]) if len(sys.argv) > 3 else 32
arena_gib = int(sys.argv[4]) if len(sys.argv) > 4 else 90
rec = ca.load_network(src); tree = ca.tree_from_record(rec)
t0 = time.time()
base = pf.modelled_seconds(tree)[0]
print('start %.1f ms x 2^%.0f = %.3e s' % (base*1e3, math.log2(tree.nslices), base*tree.nslices), flush=True)
def prog(rnd, obj, sz, t, v):
    print('round', rnd, obj if isinstance(obj, str) else 'fused-model', sz, '2^%.0f' % math.log2(t.nslices), '%.3e s' % v, '(%.0fs)' % (time.time()-t0), flush=True)
new = pf.refine(tree, objectives=(pf.MI355X_C64_FUSED, "time", "combo-64", "combo-32", "combo-128"), subtree_sizes=(8, 10, 12, 14),
                progress=prog, max_width=2**width, max_arena_bytes=arena_gib * 2**30)
secs, arena = pf.modelled_seconds(new)
print('final %.1f ms x 2^%.0f = %.3e s arena %.0f GiB' % (secs*1e3, math.log2(new.nslices), secs*new.nslices, arena/2**30))
out = {k: rec[k] for k in ("source", "inputs", "output", "size_dict") if k in rec}
out["path"] = [list(p) for p in new.get_path()]; out["sliced_inds"] = list(new.sliced_inds)
out["search"] = {"optimizer": "pathfind.refine of %s under the round-4 executor model (fused pairs and single stem steps, bf16 x 3 arithmetic: matrix rate x 1.6): objectives MI355X_C64_FUSED, time, combo-64/32/128; subtree sizes 8-14; width <= 2^%d, arena <= %d GiB" % (src.split('/')[-1], width, arena_gib), "seconds": round(time.time()-t0)}
out["stats"] = {"nslices_log2": math.log2(new.nslices), "contraction_cost_log10": new.contraction_cost(log=10), "cost_per_slice": new.contraction_cost() // new.nslices, "max_size_log2": new.max_size(log=2), "model_ms_per_slice_bf16x3": secs*1e3, "arena_gib": arena/2**30}
json.dump(out, open(dst, 'w'), ensure_ascii=False)
