#!/usr/bin/env python
"""Refine a sliced contraction tree with the package's own host tools
(cotengra_amd.pathfind: native subtree reconfiguration, csrc/ctg_pathfind.cpp)
and write the result as a tree fixture.  No reference code is involved.

    python tests/golden/gen/refine_native.py SRC.json combo-512 OUT.json [flat|mi355x]
    python tests/golden/gen/refine_native.py SRC.json time,combo-64,combo-128 OUT.json mi355x 8,10,12,14

1. ``subtree_reconfigure(subtree_size=10, minimize=OBJ)`` on the sliced tree:
   fewer MACs per slice at a smaller width;
2. greedily take indices out of the slicing again while the largest
   intermediate stays <= 2^32 elements and the arena <= 160 GiB, each time the
   index that lowers the modelled time to the full result most, then
   reconfigure again (subtree size 10 / 12 alternating); six rounds.

The time model prices every step at max(flops / 125 TFLOP/s, bytes / 4.8 TB/s)
("flat", the default) or with the matrix-core rate measured for its contracted
extent K ("mi355x" = cotengra_amd.pathfind.MI355X_C64) -- the rates the kernels
reach on an MI355X.  ``OBJ`` = ``combo-F`` is ``flops + F *
size``: the larger F, the higher the arithmetic intensity of the steps (and the
FLOP/s), the smaller F, the less total work; ``OBJ`` = ``time`` makes the
reconfiguration itself minimise the modelled seconds.

With a fifth argument (subtree sizes) the script sweeps instead: every
objective of the comma-separated list x every subtree size, each followed by
the unslicing of step 2; a candidate is kept when the modelled time to the full
result drops; repeated until a whole sweep brings nothing (sycamore_m20_w32_time.json).
"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import cotengra_amd as ca  # noqa: E402
from cotengra_amd import pathfind  # noqa: E402
from cotengra_amd.plan import compile_tree  # noqa: E402

MAX_WIDTH = 2**32
MAX_ARENA_GIB = 160.0
P_FLOPS, BW = 125e12, 4.8e12


MODEL = "flat"


def model(tree):
    plan = compile_tree(tree, "complex64")
    rows = plan.describe_steps()
    if MODEL == "flat":
        t = sum(max(8 * r["macs"] / P_FLOPS, r["bytes"] / BW) for r in rows)
    else:
        m = pathfind.MI355X_C64
        t = sum(m.step_seconds(r["macs"], r["bytes"] / 8, r["K"], r["N"]) for r in rows if r["macs"])
    return t, plan.flops_per_slice() / t / 1e12, plan.arena_elems * 8 / 2**30


def stat(tree, tag):
    t, tf, arena = model(tree)
    print(
        f"{tag}: 2^{math.log2(tree.nslices):.0f} slices, 10^{tree.contraction_cost(log=10):.3f} MACs, "
        f"width 2^{tree.max_size(log=2):.0f}, model {t * 1e3:.1f} ms/slice {tf:.1f} TFLOP/s "
        f"{t * tree.nslices / 86400:.2f} days, arena {arena:.0f} GiB",
        flush=True,
    )
    return t * tree.nslices


def with_sliced(rec, tree, sliced):
    r = dict(rec)
    r["path"] = [list(p) for p in tree.get_path()]
    r["sliced_inds"] = list(sliced)
    return ca.tree_from_record(r)


def unslice(rec, tree):
    """Step 2: take indices out of the slicing while the model says it pays."""
    while True:
        cur = model(tree)[0] * tree.nslices
        best = None
        for ix in list(tree.sliced_inds):
            cand = with_sliced(rec, tree, [j for j in tree.sliced_inds if j != ix])
            if cand.max_size() > MAX_WIDTH:
                continue
            t, _, arena = model(cand)
            if arena > MAX_ARENA_GIB:
                continue
            if best is None or t * cand.nslices < best[0]:
                best = (t * cand.nslices, cand)
        if best is None or best[0] >= cur:
            return tree
        tree = best[1]


def sweep(rec, tree, objectives, sizes):
    """The library routine (cotengra_amd.pathfind.refine) with this script's limits."""
    assert MODEL == "mi355x", "the sweep prices steps with pathfind.MI355X_C64"
    stat(tree, "start")
    return pathfind.refine(
        tree, objectives=objectives, subtree_sizes=sizes, max_width=MAX_WIDTH,
        max_arena_bytes=MAX_ARENA_GIB * 2**30,
        progress=lambda rnd, obj, sz, t, v: stat(t, f"sweep {rnd} {obj} subtree {sz}"),
    )


def main():
    global MODEL
    src, obj, dst = sys.argv[1:4]
    MODEL = sys.argv[4] if len(sys.argv) > 4 else "flat"
    rec = ca.load_network(src)

    tree = ca.tree_from_record(rec)
    t0 = time.time()
    if len(sys.argv) > 5:
        tree = sweep(rec, tree, obj.split(","), [int(x) for x in sys.argv[5].split(",")])
        write(rec, tree, src, obj + " " + sys.argv[5], dst, t0)
        return
    stat(tree, "start")
    tree = pathfind.subtree_reconfigure(tree, subtree_size=10, minimize=obj)
    stat(tree, f"reconfigured ({obj})")
    for rnd in range(6):
        tree = unslice(rec, tree)
        again = pathfind.subtree_reconfigure(tree, subtree_size=10 + (rnd % 2) * 2, minimize=obj)
        if stat(again, f"round {rnd}") < model(tree)[0] * tree.nslices:
            tree = again
    write(rec, tree, src, obj, dst, t0)


def write(rec, tree, src, obj, dst, t0):
    out = dict(rec)
    out["path"] = [list(p) for p in tree.get_path()]
    out["sliced_inds"] = list(tree.sliced_inds)
    t, tf, arena = model(tree)
    out["search"] = {
        "optimizer": f"tests/golden/gen/refine_native.py {os.path.basename(src)} {obj} {MODEL} "
        "(cotengra_amd.pathfind: native subtree reconfiguration + model-guided unslicing)",
        "source_search": rec.get("search"),
        "seconds": round(time.time() - t0),
    }
    out["stats"] = {
        "nslices_log2": math.log2(tree.nslices),
        "contraction_cost_log10": tree.contraction_cost(log=10),
        "cost_per_slice": tree.contraction_cost() // tree.nslices,
        "write_per_slice": tree.total_write() // tree.nslices,
        "max_size_log2": tree.max_size(log=2),
        "model_ms_per_slice": t * 1e3,
        "arena_gib": arena,
    }
    with open(dst, "w") as f:
        json.dump(out, f, ensure_ascii=False)
    stat(tree, "final")


if __name__ == "__main__":
    main()
