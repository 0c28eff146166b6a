#!/usr/bin/env python
"""Polish the width-2^33 tree once more under a pair model in which the stem pairs multiply on the
bf16 matrix cores (csrc/ctg_stem.hip: BF3): CTG_STEM_BF16X3 in the environment
makes ``stem.pair_seconds`` price the pairs' matrix work at ``stem.BF16X3_SPEEDUP`` x the fp32 rate,
everything else as tests/golden/gen/refine_fused.py.  The result is the better tree for BOTH
arithmetics (measured, round 3: 470 ms per slice with fp32 MFMAs against 487, 408 ms with bf16 x 3
against 438-449).  Host tools of this package only; about 25 minutes.

    python tests/golden/gen/refine_bf3.py tests/golden/trees/sycamore_m20_w33_fused.json OUT.json
"""
import json, math, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', '..'))
import cotengra_amd as ca
os.environ["CTG_STEM_BF16X3"] = "1"              # (read by stem.pair_seconds)
from cotengra_amd import pathfind as pf
src, dst = sys.argv[1], sys.argv[2]
rec = ca.load_network(src); tree = ca.tree_from_record(rec)
t0 = time.time()
base = pf.modelled_seconds(tree)[0]
print('start %.1f ms x 2^%.0f = %.3e s' % (base*1e3, math.log2(tree.nslices), base*tree.nslices), flush=True)
def prog(rnd, obj, sz, t, v):
    print('round', rnd, obj if isinstance(obj, str) else 'fused-model', sz, '2^%.0f' % math.log2(t.nslices), '%.3e s' % v, '(%.0fs)' % (time.time()-t0), flush=True)
new = pf.refine(tree, objectives=(pf.MI355X_C64_FUSED, "time", "combo-64", "combo-32", "combo-128"), subtree_sizes=(8, 10, 12, 14),
                progress=prog, max_width=2**33, max_arena_bytes=170 * 2**30)
secs, arena = pf.modelled_seconds(new)
print('final %.1f ms x 2^%.0f = %.3e s arena %.0f GiB' % (secs*1e3, math.log2(new.nslices), secs*new.nslices, arena/2**30))
out = {k: rec[k] for k in ("source", "inputs", "output", "size_dict") if k in rec}
out["path"] = [list(p) for p in new.get_path()]; out["sliced_inds"] = list(new.sliced_inds)
out["search"] = {"optimizer": "pathfind.refine of %s under the fused-pair model with the pairs' matrix rate x 1.6 (bf16 x 3, round 3): objectives MI355X_C64_FUSED, time, combo-64/32/128; subtree sizes 8-14; width <= 2^33, arena <= 170 GiB" % src.split('/')[-1], "seconds": round(time.time()-t0)}
out["stats"] = {"nslices_log2": math.log2(new.nslices), "contraction_cost_log10": new.contraction_cost(log=10), "cost_per_slice": new.contraction_cost() // new.nslices, "max_size_log2": new.max_size(log=2), "model_ms_per_slice_bf16x3": secs*1e3, "arena_gib": arena/2**30}
json.dump(out, open(dst, 'w'), ensure_ascii=False)
