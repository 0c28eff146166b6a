#!/usr/bin/env python
"""GPU check of slice groups (CTG_SLICE_GROUPS, plan.choose_slice_group; round 4):
(1) m20 trees narrowed to width 2^20, group indices chosen as on the full trees: whole groups, partial
    groups and lone slices in one list through ctg_exec_run_slice_list against the complex128 oracle;
(2) full-width trees: the slices of one group with the shared steps computed once == the same slices
    with every step computed for every slice, bit for bit (same kernels, same order of additions), and
    what the group costs either way.

    python tools/check_groups.py [tree fixtures ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CTG_SLICE_GROUPS"] = "1"

import cotengra_amd as ca  # noqa: E402
from cotengra_amd import plan as P  # noqa: E402
from cotengra_amd.contractor import HipContractor  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402
from oracle.plan_interp import group_key, group_members  # noqa: E402

TREES = os.path.join(ROOT, "tests", "golden", "trees")
fixtures = sys.argv[1:] or ["sycamore_m20_w32_r4.json", "sycamore_m20_native.json", "sycamore_m20_w33_bf3.json"]
bad = 0
for fx in fixtures:
    tree = ca.tree_from_record(ca.load_network(os.path.join(TREES, fx)))
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    # ---- (1) narrowed, against the oracle
    keep = (P.GROUP_MIN_WIDTH, P.GROUP_MIN_SAVING)
    P.GROUP_MIN_WIDTH, P.GROUP_MIN_SAVING = 1, 0.0
    small = tree.slice(target_size=2**20)
    fn = HipContractor(small)
    plan = fn.get_plan("complex64")[0]
    ids = group_members(plan, 3) + group_members(plan, 77777)[:3] + [12345, 5]
    a128 = [a.astype("complex128") for a in arrays]
    ref = sum(complex(orc.contract_slice(small, a128, i)) for i in ids)
    st = fn.setup(*arrays)
    ex = st["exec"]
    ex.zero_result()
    ex.run_slice_list(ids[::-1])
    got = complex(ex.download_result())
    err = abs(got - ref) / abs(ref)
    n_shared = sum(s.group for s in plan.steps)
    # (the gate of the suite: what numpy's own single precision loses on the same sum sets the scale)
    np64 = sum(complex(orc.contract_slice(small, arrays, i)) for i in ids)
    ok = err <= max(1e-5, 8.0 * abs(np64 - ref) / abs(ref)) and n_shared > 0
    print(f"{fx} narrowed to 2^20: group {plan.group_inds}, {n_shared} shared steps, {len(ids)} slices in "
          f"{len({group_key(plan, i) for i in ids})} groups: err {err:.2e} {'ok' if ok else 'WRONG'}")
    bad += not ok
    fn.close()
    P.GROUP_MIN_WIDTH, P.GROUP_MIN_SAVING = keep
    # ---- (2) full width: one group with and without sharing
    res = {}
    for mode in ("1", "0"):
        os.environ["CTG_SLICE_GROUPS"] = mode
        fn = HipContractor(tree)
        plan = fn.get_plan("complex64")[0]
        if mode == "1":
            members = group_members(plan, 5)
            shared = sum(s.group for s in plan.steps)
        ex = fn.setup(*arrays)["exec"]
        ex.run_slice_list(members[:1])      # warm up (one-time launcher set-up)
        ex.sync()
        ex.zero_result()
        t0 = time.perf_counter()
        ex.run_slice_list(members)
        ex.sync()
        dt = time.perf_counter() - t0
        res[mode] = (complex(ex.download_result()), dt, plan.arena_elems * 8 / 2**30)
        fn.close()
    os.environ["CTG_SLICE_GROUPS"] = "1"
    same = res["1"][0] == res["0"][0]
    print(f"{fx} full width: group of {len(members)} slices, {shared} shared steps: {res['1'][1] * 1e3 / len(members):.1f} ms per "
          f"slice (arena {res['1'][2]:.0f} GiB) vs {res['0'][1] * 1e3 / len(members):.1f} ms without sharing (arena {res['0'][2]:.0f} GiB); "
          f"bit-identical {same} {'ok' if same else 'WRONG'}")
    bad += not same
# ---- (3) a small, batched configuration: the 200-tensor hyper network (C5), whole groups per launch
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G  # noqa: E402

case = next(c for c in G.cases("tree") if c["name"] == "C5_hyper200")
tree = G.tree_of(case)
arrays = [np.asarray(a).astype("complex64") for a in G.arrays_of(case, "complex128", tree)]
res = {}
for mode in ("1", "0"):
    os.environ["CTG_SLICE_GROUPS"] = mode
    fn = HipContractor(tree, handle_slicing=True)
    plan = fn.get_plan("complex64")[0]
    if mode == "1":
        gs = int(plan.group_size)
        ids = [i for g in range(22) for i in plan.group_ids(g)]
    ex = fn.setup(*arrays)["exec"]
    for _ in range(3):
        ex.run_slice_list(ids)
    ex.sync()
    ex.zero_result()
    t0 = time.perf_counter()
    for _ in range(10):
        ex.run_slice_list(ids)
    ex.sync()
    dt = (time.perf_counter() - t0) / 10
    res[mode] = (np.asarray(ex.download_result()).copy(), dt, ex.batch, sum(s.group for s in plan.steps))
    fn.close()
os.environ["CTG_SLICE_GROUPS"] = "1"
scale = np.abs(res["0"][0]).max()
err = np.abs(res["1"][0] - res["0"][0]).max() / scale
ok = err <= 1e-5 and gs > 1
print(f"C5 hyper network: {len(ids)} slices in groups of {gs} ({res['1'][3]} shared steps, {res['1'][2]} slices per launch): "
      f"{res['1'][1] * 1e3:.2f} ms vs {res['0'][1] * 1e3:.2f} ms without groups ({res['0'][2]} per launch); sums agree to {err:.1e} "
      f"{'ok' if ok else 'WRONG'}")
bad += not ok
print("FAILED" if bad else "ALL OK", bad)
