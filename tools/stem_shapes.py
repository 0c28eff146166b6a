#!/usr/bin/env python
"""Which instantiations of the fused stem kernel (csrc/ctg_stem.hip: CTG_STEM_STATIC) the
tree fixtures need: (PACK1, PACK2, RT1, CS1, NCH, IT2, BR1, K2Q, VEC) with their share of the work."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cotengra_amd as ca  # noqa: E402
from cotengra_amd.plan import KIND_STEM2, compile_tree  # noqa: E402

need = {}
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20*.json"))):
    tree = ca.tree_from_record(json.load(open(f)))
    plan = compile_tree(tree, "complex64")
    for s in plan.steps:
        if s.kind != KIND_STEM2:
            continue
        st = s.stem
        cs1 = max(1, st["N1"] // 32)
        p1, p2 = st["N1"] == 16, st["N2"] == 16
        nch, it2 = st["K1"] // 16, (st["items"] // 8 if st["items"] % 8 == 0 else -st["items"])
        # which small operand's fragments live in registers (csrc/ctg_stem.hip: stem2_breg)
        r1 = st["K1"] if st["K1"] <= 64 else 0
        r2 = (st["K2"] if p2 else 2 * st["K2"]) if ((p2 and st["K2"] <= 64) or (not p2 and it2 == 1 and st["K2"] <= 32)) else 0
        if r1 and r2 and r1 + r2 > 96:
            r2 = 0
        key = (p1, p2, (1 << (st["nr1"] - 5)) * cs1 // 8, cs1, nch, it2, bool(r1), st["K2"] // 4 if r2 else 0, bool(st["vec"]))
        d = need.setdefault(key, [0, set()])
        d[0] += s.macs
        d[1].add((os.path.basename(f)[13:-5], st["K1"], st["N1"], st["K2"], st["N2"]))
for key, (macs, where) in sorted(need.items(), key=lambda kv: -kv[1][0]):
    print("X(%s, %s, %d, %d, %d, %d, %s, %d, %s)" % (str(key[0]).lower(), str(key[1]).lower(), *key[2:6], str(key[6]).lower(), key[7], str(key[8]).lower()),
          "%.2e" % macs, sorted(where)[:4])
