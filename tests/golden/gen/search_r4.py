#!/usr/bin/env python
"""Round 4: a wider stand-alone search for a Sycamore-53 m20 tree under the model of the executor as it
is now (fused pairs and single stem steps in the bf16 x 3 arithmetic).  Stage 1 (this script): SEEDS
draws of ``pathfind.sample_sliced_tree`` on WORKERS processes, ranked by the modelled time to the
amplitude; the TOP best are written to OUTDIR/cand_<rank>_seed<seed>.json.  Stage 2: each candidate
is polished by tests/golden/gen/refine_r4.py (run them side by side).  Host tools of this package
only; no reference optimizer and no reference-found tree is involved.

    python tests/golden/gen/search_r4.py OUTDIR [first seed = 256] [n seeds = 768] [workers = 6] [log2 width = 32] [top = 6]
"""
import concurrent.futures as cf
import json
import math
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
os.environ.pop("CTG_STEM_BF16X3", None)          # (the default arithmetic: bf16 x 3)
import cotengra_amd as ca  # noqa: E402
from cotengra_amd import pathfind as pf  # noqa: E402
from cotengra_amd.tree import ContractionTree  # noqa: E402


def draw(job):
    inputs, output, size_dict, target, seed = job
    tree = pf.sample_sliced_tree(inputs, output, size_dict, target, seed, 64, "combo-64")
    secs, arena = pf.modelled_seconds(tree)
    return seed, [list(p) for p in tree.get_path()], list(tree.sliced_inds), secs * tree.nslices, arena


def main():
    outdir = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    width = int(sys.argv[5]) if len(sys.argv) > 5 else 32
    top = int(sys.argv[6]) if len(sys.argv) > 6 else 6
    os.makedirs(outdir, exist_ok=True)
    rec = ca.load_network(os.path.join(HERE, "..", "trees", "sycamore_m20_w32.json"))
    inputs = [tuple(t) for t in rec["inputs"]]
    output = tuple(rec["output"])
    jobs = [(inputs, output, dict(rec["size_dict"]), 2**width, first + i) for i in range(n)]
    t0 = time.time()
    ranked = []
    with cf.ProcessPoolExecutor(max_workers=workers) as pool:
        for k, r in enumerate(pool.map(draw, jobs)):
            ranked.append(r)
            if k % 32 == 31:
                b = min(ranked, key=lambda x: x[3])
                print("%d draws, best so far seed %d: %.3e s (%.0fs)" % (k + 1, b[0], b[3], time.time() - t0), flush=True)
    ranked.sort(key=lambda r: (r[3], r[0]))
    for rank, (seed, path, sliced, total, arena) in enumerate(ranked[:top]):
        tree = ContractionTree.from_path(inputs, output, rec["size_dict"], path=[tuple(p) for p in path]).apply_slicing_(sliced)
        out = {k: rec[k] for k in ("source", "inputs", "output", "size_dict") if k in rec}
        out["path"] = path
        out["sliced_inds"] = sliced
        out["search"] = {"optimizer": "tests/golden/gen/search_r4.py: draw %d of seeds %d..%d of pathfind.sample_sliced_tree "
                         "(target 2^%d), ranked by the round-4 executor model" % (seed, first, first + n - 1, width),
                         "seconds": round(time.time() - t0), "workers": workers}
        out["stats"] = {"nslices_log2": math.log2(tree.nslices), "contraction_cost_log10": tree.contraction_cost(log=10),
                        "max_size_log2": tree.max_size(log=2), "model_seconds_total": total, "arena_gib": arena / 2**30}
        dst = os.path.join(outdir, "cand_%d_seed%d.json" % (rank, seed))
        with open(dst, "w") as f:
            json.dump(out, f, ensure_ascii=False)
        print("rank %d seed %d: %.3e s modelled, 2^%.0f slices, arena %.0f GiB -> %s" % (
            rank, seed, total, math.log2(tree.nslices), arena / 2**30, dst), flush=True)


if __name__ == "__main__":
    main()
