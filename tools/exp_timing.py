#!/usr/bin/env python
"""Where a tile of the fast MFMA kernel spends its time.

Needs an experiment build of the library with the in-kernel wall-clock stamps:

  cd cotengra_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCTG_TIMING \
      -o ../lib/libctg_hip_timing.so ctg_runtime.hip ctg_kernels_valu.hip ctg_pair_mfma.hip ctg_pair_mfma_f64.hip
  CTG_LIB=$PWD/cotengra_amd/lib/libctg_hip_timing.so python tools/exp_timing.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cotengra_amd as ca
from cotengra_amd import runtime
from cotengra_amd.contractor import HipContractor
lib = runtime.load()
lib.ctg_debug_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
for eq, sizes in [("ak,kb->ab", dict(a=16777216, k=64, b=256)), ("ak,kb->ab", dict(a=4194304, k=128, b=512)), ("ak,kb->ab", dict(a=2097152, k=512, b=512)), ("ak,kb->ab", dict(a=16777216, k=32, b=64))]:
    (ta, tb), out = ca.eq_to_inputs_output(eq)
    tree = ca.ContractionTree.from_path([ta, tb], out, sizes, path=[(0, 1)])
    rng = np.random.default_rng(0)
    arrays = [(rng.normal(size=[sizes[i] for i in t]) + 1j * rng.normal(size=[sizes[i] for i in t])).astype("complex64") for t in (ta, tb)]
    fn = HipContractor(tree)
    st = fn.setup(*arrays)
    ex = st["exec"]
    ex.run_slices(0, 1, 1); ex.sync()
    lib.ctg_debug_timing(buf, 1)
    ms = ex.profile_slice(0)
    lib.ctg_debug_timing(buf, 1)
    n = max(buf[5], 1)
    names = ["constants", "first gather+stage", "k loop", "epilogue issue", "store drain"]
    print(sizes, "kernel ms %.3f" % ms.max(), "blocks", buf[5], " ".join("%s %.2f us" % (nm, buf[i] / n / 100.0) for i, nm in enumerate(names)))
    fn.close()
