#!/usr/bin/env python
"""Phase timeline of the fused stem kernel (experiment build -DCTG_STEM_TIMELINE: CTG_LIB=.../libctg_tl.so):
one slice of a tree; the first launch whose shape is CTG_TL_SHAPE="K1,N1,K2,N2" records, for the 8 waves of
workgroup 0 and their first 256 tiles, the shader clock at: tile start | step 1 issued | past barrier 1 | scatter
done | past barrier 2 | step 2 issued.  Prints the mean cycles per phase (per wave and over all waves).

  CTG_LIB=cotengra_amd/lib/exp/libctg_tl.so CTG_TL_SHAPE=32,32,64,64 [CTG_STEM_FORM=1] python tools/exp_stem_timeline.py [tree]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from cotengra_amd import runtime  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

tree_file = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_native.json")
tree = ca.tree_from_record(ca.load_network(tree_file))
arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
dev = torch.device("cuda", 0)
fn = HipContractor(tree, handle_slicing=True)
st = fn.setup(*[torch.as_tensor(a, device=dev) for a in arrays])
ex = st["exec"]
ex.zero_result()
ex.run_slice_list([0])
ex.sync()
lib = runtime.load()
T = 256
buf = np.zeros((8, T, 6), dtype=np.uint64)
# (the fp16 x 2 object has its own copy of the hook: CTG_TL_H2=1 reads that one)
hook = lib.ctg_debug_stem_timeline_h2 if os.environ.get("CTG_TL_H2") else lib.ctg_debug_stem_timeline
hook.argtypes = [C.c_void_p, C.c_int]
rc = hook(C.c_void_p(buf.ctypes.data), 1)
assert rc == 0
fn.close()
ok = buf[:, :, 5] > 0
n = int(ok.all(axis=0).sum())
print("shape", os.environ.get("CTG_TL_SHAPE"), "form", os.environ.get("CTG_STEM_FORM", "default"), "tiles recorded", n)
if n < 8:
    raise SystemExit("nothing recorded (shape not in this tree?)")
b = buf[:, 4:n, :].astype(np.int64)          # (skip the first tiles: the pipeline fills)
names = ["step 1 (issue)", "wait barrier 1", "scatter", "wait barrier 2", "step 2 (issue)", "to next tile"]
d = [b[:, :, 1] - b[:, :, 0], b[:, :, 2] - b[:, :, 1], b[:, :, 3] - b[:, :, 2], b[:, :, 4] - b[:, :, 3],
     b[:, :, 5] - b[:, :, 4], b[:, 1:, 0] - b[:, :-1, 5]]
tile = b[:, 1:, 0] - b[:, :-1, 0]
print("cycles per tile: mean %.0f  (min %.0f  max %.0f over waves' means)" % (tile.mean(), tile.mean(axis=1).min(), tile.mean(axis=1).max()))
for nm, x in zip(names, d):
    print("  %-16s mean %7.0f   per wave: %s   p90 %7.0f" % (nm, x.mean(), " ".join("%6.0f" % v for v in x.mean(axis=1)), np.percentile(x, 90)))
