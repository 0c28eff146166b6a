#!/usr/bin/env python
"""The ACCURACY side of the two product-count levers VERDICT r5 names, emulated in numpy on the narrowed m20 trees
(CPU only; the kernels that would run them are not built -- see profiles/r6_product_levers.txt):

  fp32      every product exact in double, every step's result rounded to complex64 (what fp32 MFMA and the exact
            three-limb bf16 split deliver, up to the accumulation order)
  3m        the three-multiplication complex product (Gauss / ZGEMM3M): k1 = (Ar + Ai) Br, k2 = Ar (Bi - Br),
            k3 = Ai (Br + Bi), Re = k1 - k3, Im = k1 + k2 -- the sums and the three real GEMMs in float32
  fp16x2    every operand as two ROUNDED fp16 limbs under a per-tensor power-of-two scale (largest element at 2^14),
            the three products h1 h1', h1 h2', h2 h1' exact, the result rounded to complex64

Each against the complex128 oracle, next to numpy's own complex64 run, on the slices of
profiles/r5_single_precision_errors.txt:   python tools/exp_product_levers.py [log2_width]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

TREES = ["sycamore_m20_w32_c512.json", "sycamore_m20_native.json", "sycamore_m20_fused.json", "sycamore_m20_w33_bf3.json",
         "sycamore_m20_w32_r4.json", "sycamore_m20_w32_g.json"]
log2w = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def two_fp16_limbs(x):
    """x (complex128 / float64 array) -> (h1, h2, scale): x ~ scale * (h1 + h2), limbs rounded to fp16."""
    m = float(np.max(np.abs(np.concatenate([x.real.ravel(), x.imag.ravel()])))) if x.size else 0.0
    if m == 0.0:
        return x, np.zeros_like(x), 1.0
    scale = 2.0 ** (np.floor(np.log2(m)) - 14)

    def split(v):
        v = v / scale
        h1 = v.astype(np.float16).astype(np.float64)
        h2 = (v - h1).astype(np.float16).astype(np.float64)
        return h1, h2

    r1, r2 = split(x.real)
    i1, i2 = split(x.imag)
    return r1 + 1j * i1, r2 + 1j * i2, scale


def step(mode, tdot, arg, perm, l, r):
    def mul(a, b):
        p = orc.tensordot(a, b, arg) if tdot else orc.einsum(arg, a, b)
        return np.transpose(p, perm) if (tdot and perm) else p

    if mode == "fp32":
        return mul(l, r).astype(np.complex64).astype(np.complex128)
    if mode == "fp16x2":
        a1, a2, sa = two_fp16_limbs(l)
        b1, b2, sb = two_fp16_limbs(r)
        p = (mul(a1, b1) + mul(a1, b2) + mul(a2, b1)) * (sa * sb)
        return p.astype(np.complex64).astype(np.complex128)
    if mode == "3m":
        f = np.float32
        ar, ai, br, bi = (np.ascontiguousarray(v, dtype=f) for v in (l.real, l.imag, r.real, r.imag))
        k1 = mul((ar + ai).astype(f), br)
        k2 = mul(ar, (bi - br).astype(f))
        k3 = mul(ai, (br + bi).astype(f))
        return ((k1 - k3).astype(f) + 1j * (k1 + k2).astype(f)).astype(np.complex128)
    raise ValueError(mode)


def run(ops, arrays, mode):
    temps = dict(enumerate(arrays))
    p = None
    for pi, li, ri, tdot, arg, perm in ops:
        if ri is None:
            if li is None:
                temps[pi] = orc.einsum(arg, temps[pi])
                continue
            return orc.einsum(arg, temps[li])
        l, r = temps.pop(li), temps.pop(ri)
        p = step(mode, tdot, arg, perm, l, r)
        temps[pi] = p
    return p


print(f"relative error of one slice against the complex128 oracle, trees narrowed to width 2^{log2w}")
print("%-28s %22s %10s %10s %10s %10s" % ("tree", "slice", "numpy c64", "fp32", "3m", "fp16x2"))
worst = {"numpy c64": 0.0, "fp32": 0.0, "3m": 0.0, "fp16x2": 0.0}
for name in TREES:
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", name)))
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    small = tree.slice(target_size=2**log2w)
    ops = orc.extract_contractions(small)
    a128 = [a.astype("complex128") for a in arrays]
    for sid in (3, small.nslices - 1 if small.nslices < 2**62 else 12345):
        xs = orc.slice_arrays(small, a128, sid)
        ref = complex(np.asarray(orc.run_contractions(ops, xs)).reshape(-1)[0])
        row = {"numpy c64": complex(np.asarray(orc.run_contractions(ops, orc.slice_arrays(small, arrays, sid))).reshape(-1)[0])}
        for mode in ("fp32", "3m", "fp16x2"):
            row[mode] = complex(np.asarray(run(ops, xs, mode)).reshape(-1)[0])
        errs = {k: abs(v - ref) / abs(ref) for k, v in row.items()}
        for k, v in errs.items():
            worst[k] = max(worst[k], v)
        print("%-28s %22d %10.2e %10.2e %10.2e %10.2e" % (name, sid, errs["numpy c64"], errs["fp32"], errs["3m"], errs["fp16x2"]))
print("%-28s %22s %10.2e %10.2e %10.2e %10.2e" % ("worst", "", worst["numpy c64"], worst["fp32"], worst["3m"], worst["fp16x2"]))
