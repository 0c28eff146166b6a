#!/usr/bin/env python
"""Run the fused stem kernel under its bounds-checked build (SURVEY section 5 "sanitizers";
VERDICT r3 missing item 4): every gather of the big operand and every store of the result is
tested on the device against the tensor's extent (csrc/ctg_stem.hip, -DCTG_STEM_BOUNDS;
`tools/build_variants.py bounds=-DCTG_STEM_BOUNDS`), violations are counted and skipped.

    CTG_LIB=cotengra_amd/lib/exp/libctg_bounds.so python tools/check_stem_bounds.py [log]

Cases: the 16 stem shapes of tests/golden_util.py (x 3 seeds x sliced / unsliced = 96 networks,
static and run-time-count instantiations, X / Y and row-interleaved step 2, and once more on the
bf16 x 3 kernels) with every result also checked against the numpy complex128 oracle -- a skipped
access would show there -- plus one full-width slice of the m20 headline tree (2^32-element
tensors: offsets beyond 2^31, 68 GB stores) when the device has the memory."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cotengra_amd as ca  # noqa: E402
from cotengra_amd import runtime, stem  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402
from cotengra_amd.plan import KIND_STEM2  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

import golden_util as G  # noqa: E402

lib = runtime.load()
if not hasattr(lib, "ctg_debug_stem_oob"):
    raise SystemExit("this library has no bounds checks: build tools/build_variants.py bounds=-DCTG_STEM_BOUNDS "
                     "and set CTG_LIB=cotengra_amd/lib/exp/libctg_bounds.so")
lib.ctg_debug_stem_oob.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout


def oob(reset=True):
    v = (C.c_ulonglong * 2)()
    assert lib.ctg_debug_stem_oob(v, int(reset)) == 0
    return int(v[0]), int(v[1])


def say(*a):
    print(*a, file=out, flush=True)


stem.gather_rate = lambda run_bytes: 5.4e12   # every pair the kernel can take
oob()
bad = launches = 0
for mode in ("fp32", "bf16x3"):
    os.environ["CTG_STEM_BF16X3"] = "1" if mode == "bf16x3" else "0"
    for ci, (nq, gates) in enumerate(G.STEM_CASES):
        for seed in (0, 1, 2):
            for sliced in (0, 2):
                tree = G.stem_network(nq, gates, 100 * ci + seed, sliced=sliced)
                arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=seed, dtype="complex64")
                ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
                fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
                plan = fn.get_plan("complex64")[0]
                n_stem = sum(1 for s in plan.steps if s.kind == KIND_STEM2)
                got = np.asarray(fn(*arrays))
                names = sorted({k for k in fn.setup(*arrays)["exec"].step_kernels() if k.startswith("stem2")})
                fn.close()
                g, s = oob()
                err = np.abs(got - ref).max() / np.abs(ref).max()
                ok = g == 0 and s == 0 and err <= 1e-4
                bad += not ok
                launches += n_stem * tree.nslices
                say(f"{mode} case {ci} seed {seed} sliced {sliced}: {n_stem} fused pairs x {tree.nslices} slices, "
                    f"out-of-bounds gathers {g} stores {s}, err vs oracle {err:.1e} {'ok' if ok else 'BAD'}  {names}")
    # single steps (the kernel's first half alone), every one the kernel can take
    keep_gain, stem.MIN_GAIN = stem.MIN_GAIN, -1e9
    for ci, (nq, gates) in enumerate(G.ONE_CASES):
        for sliced in (0, 2):
            tree = G.stem_network(nq, gates, 300 + ci, sliced=sliced)
            arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=ci, dtype="complex64")
            ref = np.asarray(orc.contract(tree, [a.astype("complex128") for a in arrays]))
            fn = HipContractor(tree, fuse=True, fuse_min_elems=1 << 10)
            n_one = sum(1 for s in fn.get_plan("complex64")[0].steps if s.kind == KIND_STEM2 and s.stem.get("one"))
            got = np.asarray(fn(*arrays))
            names = sorted({k for k in fn.setup(*arrays)["exec"].step_kernels() if k.startswith("stem2")})
            fn.close()
            g, s = oob()
            err = np.abs(got - ref).max() / np.abs(ref).max()
            ok = g == 0 and s == 0 and err <= 1e-4
            bad += not ok
            launches += n_one * tree.nslices
            say(f"{mode} single-step case {ci} sliced {sliced}: {n_one} single steps x {tree.nslices} slices, out-of-bounds "
                f"gathers {g} stores {s}, err vs oracle {err:.1e} {'ok' if ok else 'BAD'}  {names}")
    stem.MIN_GAIN = keep_gain
    os.environ.pop("CTG_STEM_BF16X3", None)
say(f"stem networks: {launches} stem launches checked, {bad} bad")

# one full-width slice of the headline tree: 2^32-element tensors
try:
    import torch

    free = torch.cuda.mem_get_info()[0]
except Exception:  # noqa: BLE001
    free = 0
if free > 90 * 2**30:
    tree = ca.tree_from_record(ca.load_network(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_native.json")))
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    for mode in ("fp32", "bf16x3"):
        os.environ["CTG_STEM_BF16X3"] = "1" if mode == "bf16x3" else "0"
        fn = HipContractor(tree, handle_slicing=True)
        plan = fn.get_plan("complex64")[0]
        n_stem = sum(1 for s in plan.steps if s.kind == KIND_STEM2)
        amp = complex(np.asarray(fn.contract_slice(arrays, 5)))
        fn.close()
        g, s = oob()
        bad += (g + s) != 0
        say(f"{mode} sycamore_m20_native.json slice 5 at full width (2^32): {n_stem} fused pairs, out-of-bounds gathers {g} "
            f"stores {s}, amplitude {amp:.6e} {'ok' if g + s == 0 else 'BAD'}")
        os.environ.pop("CTG_STEM_BF16X3", None)
else:
    say(f"(full-width slice skipped: {free / 2**30:.0f} GiB free)")
say("ALL OK" if not bad else f"FAILED {bad}")
sys.exit(1 if bad else 0)
