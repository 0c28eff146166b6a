#!/usr/bin/env python
"""Condition numbers of narrowed m20 slices (CPU only, numpy complex128 oracle): why one slice of
``sycamore_m20_w32_g.json`` at width 2^20 sits at 1.2e-5 in complex64 in EVERY arithmetic (numpy's own complex64
run: 6.9e-6) while its neighbours sit at 1e-6.

For every pairwise step of the slice, kappa_step = || |A| . |B| ||_max / || A . B ||_max (how much larger the sum of
magnitudes is than the largest result: the cancellation inside the step's sums), and for the closing dot product
kappa = sum |a_k b_k| / |sum a_k b_k|.  A single-precision result carries ~ eps x (accumulated cancellation):
the slice with the large kappa product is the one both numpy and the HIP path lose digits on.

  python tools/exp_condition_number.py [fixture [log2_width [slice ...]]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

fixture = sys.argv[1] if len(sys.argv) > 1 else "sycamore_m20_w32_g.json"
log2w = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", fixture)))
arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
small = tree.slice(target_size=2**log2w)
sids = [int(s) for s in sys.argv[3:]] or [3, small.nslices - 1 if small.nslices < 2**62 else 12345]
ops = orc.extract_contractions(small)
a128 = [a.astype("complex128") for a in arrays]
print(f"{fixture} narrowed to 2^{log2w}: {len(ops)} steps")
for sid in sids:
    temps = dict(enumerate(orc.slice_arrays(small, a128, sid)))
    temps32 = dict(enumerate(orc.slice_arrays(small, arrays, sid)))
    worst, log_sum, last = 1.0, 0.0, None
    for pi, li, ri, tdot, arg, perm in ops:
        if ri is None:
            continue
        l, r = temps.pop(li), temps.pop(ri)
        l32, r32 = temps32.pop(li), temps32.pop(ri)
        if tdot:
            p = orc.tensordot(l, r, arg)
            pa = orc.tensordot(np.abs(l), np.abs(r), arg)
            p32 = orc.tensordot(l32, r32, arg)
            if perm:
                p, pa, p32 = np.transpose(p, perm), np.transpose(pa, perm), np.transpose(p32, perm)
        else:
            p = orc.einsum(arg, l, r)
            pa = orc.einsum(arg, np.abs(l), np.abs(r))
            p32 = orc.einsum(arg, l32, r32)
        k = float(np.max(pa) / max(np.max(np.abs(p)), 1e-300))
        worst = max(worst, k)
        log_sum += np.log10(max(k, 1.0))
        last = (k, float(np.max(np.abs(p32 - p)) / np.max(np.abs(p))))
        temps[pi], temps32[pi] = p, p32
    val = complex(np.asarray(p).reshape(-1)[0])
    # the whole slice as ONE multilinear sum: sum of |terms| (the tree contracted on |tensors|) over |sum of terms|
    absval = float(np.asarray(orc.run_contractions(ops, [np.abs(x) for x in orc.slice_arrays(small, a128, sid)])).reshape(-1)[0])
    print(f" slice {sid}: value {val:.6e}  closing step kappa {last[0]:.1f}  largest step kappa {worst:.1f}  "
          f"numpy complex64 rel err {last[1]:.2e}  (eps_fp32 x closing kappa = {6e-8 * last[0]:.1e})")
    print(f"          sum of the steps' log10 kappa {log_sum:.1f};  whole slice: sum|terms| / |sum terms| = {absval / abs(val):.3e}")
