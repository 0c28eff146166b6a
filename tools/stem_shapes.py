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

SW, LDS = 8, 160 * 1024


def lds_ri2(st, b2_in_regs):
    b1 = (3 if st["N1"] == 16 else 2) * st["N1"] * (st["K1"] + 4)
    b2 = 2 * st["N2"] * (st["K2"] + 4)
    mid = 3 * st["rows2"] * st["ld2"]
    return 4 * (b1 + (max(mid, b2) if b2_in_regs else b2 + mid)) + 8 * st["N2"]


def shape(st):
    """csrc/ctg_stem.hip: stem2_shape, restated."""
    cs1 = max(1, st["N1"] // 32)
    p1, p2 = st["N1"] == 16, st["N2"] == 16
    rt1 = (1 << (st["nr1"] - 5)) * cs1 // SW
    nch, it2 = st["K1"] // 16, (st["items"] // SW if st["items"] % SW == 0 else 0)
    ng2 = st["ng2"]
    if not p2 and it2 > 0 and 1 <= ng2 <= SW and SW % ng2 == 0:
        fixed = rt1 * (16 if p1 else 32) + (64 if it2 > 1 else 32) + 32 + 40
        r1 = st["K1"] if st["K1"] <= 64 else 0
        r2 = st["K2"] if st["K2"] <= 64 else 0
        if fixed + r1 + r2 > 256:
            r2 = 0
        if fixed + r1 > 256:
            r1 = 0
        if lds_ri2(st, r2 != 0) <= LDS:
            return (p1, p2, rt1, cs1, nch, it2, bool(r1), st["K2"] // 4 if r2 else 0, bool(st["vec"]), True)
    r1 = st["K1"] if st["K1"] <= 64 else 0
    r2 = (st["K2"] if p2 else 2 * st["K2"]) if ((p2 and st["K2"] <= 64) or (not p2 and it2 == 1 and st["K2"] <= 32)) else 0
    if r1 and r2 and r1 + r2 > 96:
        r2 = 0
    return (p1, p2, rt1, cs1, nch, it2, bool(r1), st["K2"] // 4 if r2 else 0, bool(st["vec"]), False)


need, ones = {}, {}
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20*.json"))):
    tree = ca.tree_from_record(json.load(open(f)))
    # (the pairing is priced in the arithmetic the pairs will run in: the plans of the two modes differ)
    for mode in (True, False):
        plan = compile_tree(tree, "complex64", stem_bf16x3=mode)
        for s in plan.steps:
            if s.kind != KIND_STEM2:
                continue
            st = s.stem
            if st.get("one"):
                cs1 = st["N1"] // 32
                key = ((1 << (st["nr1"] - 5)) * cs1 // SW, cs1, st["K1"] // 16, bool(st["vec"]))
                d = ones.setdefault(key, [0, set()])
                d[0] += s.macs
                d[1].add((os.path.basename(f)[13:-5], st["K1"], st["N1"]))
                continue
            key = shape(st)
            if key[5] == 0:
                continue   # (item count not a multiple of the waves: the run-time-count variant)
            d = need.setdefault(key, [0, set()])
            d[0] += s.macs
            d[1].add((os.path.basename(f)[13:-5], st["K1"], st["N1"], st["K2"], st["N2"]))
lo = lambda b: str(b).lower()   # noqa: E731
print("// fp32 instantiations (P1, P2, RT1, CS1, NCH, IT2, BR1, K2Q, VEC, RI2), by share of the work")
for key, (macs, where) in sorted(need.items(), key=lambda kv: -kv[1][0]):
    print("X(%s, %s, %d, %d, %d, %d, %s, %d, %s, %s)" % (lo(key[0]), lo(key[1]), *key[2:6], lo(key[6]), key[7], lo(key[8]), lo(key[9])),
          "// %.2e" % macs, sorted(where)[:3])
geo = sorted({(k[0], k[1], k[2], k[3], k[4], k[5], k[8]) for k in need})
print("// geometries (P1, P2, RT1, CS1, NCH, IT2, VEC): the bf16 x 3 instantiations")
for g in geo:
    print("G(%s, %s, %d, %d, %d, %d, %s)" % (lo(g[0]), lo(g[1]), *g[2:6], lo(g[6])))
print("// single steps (RT1, CS1, NCH, VEC)")
for key, (macs, where) in sorted(ones.items(), key=lambda kv: -kv[1][0]):
    print("O(%d, %d, %d, %s)" % (*key[:3], lo(key[3])), "// %.2e" % macs, sorted(where)[:3])

# what csrc/ctg_stem.hip has: anything printed below runs the run-time-count variant (fp32) until it is added
import re  # noqa: E402

src = open(os.path.join(ROOT, "cotengra_amd", "csrc", "ctg_stem.hip")).read()


def have(macro, letter):
    body = src.split("#define %s(%s)" % (macro, letter))[-1].split("#endif")[0]   # (the full list: behind the development one)
    return {tuple(a.strip() for a in m.split(",")) for m in re.findall(r"%s\(([^)]*)\)" % letter, body)}


def fmt(key):
    return tuple(lo(v) if isinstance(v, bool) else str(v) for v in key)


missing = [("X", fmt(k)) for k in need if fmt(k) not in have("CTG_STEM_INST", "X")]
missing += [("G", fmt(g)) for g in geo if fmt(g) not in have("CTG_STEM_GEO", "G")]
missing += [("O", fmt(k)) for k in ones if fmt(k) not in have("CTG_STEM_ONE", "X")]
print("// not in csrc/ctg_stem.hip:", ", ".join("%s(%s)" % (l, ", ".join(k)) for l, k in missing) or "nothing")
