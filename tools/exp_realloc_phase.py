#!/usr/bin/env python
"""Does the two-speed alternation of consecutive processes (profiles/r6_process_alternation.txt) follow the
ALLOCATION of the arena?  One process: build the executor of the headline tree, time one slice group, close, again."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import cotengra_amd as ca  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402

rec = ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", sys.argv[1] if len(sys.argv) > 1 else "sycamore_m20_native.json"))
tree = ca.tree_from_record(rec)
arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
dev = [torch.as_tensor(a, device="cuda") for a in arrays]
import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
dummies = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else []   # executors before which a dummy stream is created
keep = []
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    if rep in dummies:
        st = ctypes.c_void_p()
        rc = hip.hipStreamCreate(ctypes.byref(st))
        # (used once: a stream is bound to a hardware queue at its first operation)
        buf = torch.zeros(1024, device="cuda")
        rc2 = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(4096), st)
        rc3 = hip.hipStreamSynchronize(st)
        print("  (dummy stream, used once ->", rc, rc2, rc3, ")", flush=True)
        keep.append(st)
    fn = HipContractor(tree)
    ex = fn.setup(*dev)["exec"]
    ex.zero_result()
    ex.run_slices(0, 4, 1)
    ex.sync()
    t0 = time.perf_counter()
    ex.run_slices(4, 8, 1)
    ex.sync()
    print("executor", rep, "ms/slice", round((time.perf_counter() - t0) / 8 * 1e3, 2), flush=True)
    fn.close()
    del ex, fn
    torch.cuda.empty_cache()
