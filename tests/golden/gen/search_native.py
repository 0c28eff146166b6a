#!/usr/bin/env python
"""Find a sliced contraction tree for the Sycamore-53 m20 network with the
package's own host tools only (cotengra_amd.pathfind.search: sampled greedy
trees + native subtree reconfiguration + slicing, best draws refined under the
MI355X machine model) and write it as a tree fixture.  No reference optimizer
and no reference-found tree is involved; the network structure comes from the
existing fixture (= the reference's examples/benchmarks JSON).

    python tests/golden/gen/search_native.py OUT.json [n_samples] [workers]
"""
import json
import math
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
import cotengra_amd as ca  # noqa: E402
from cotengra_amd import pathfind  # noqa: E402


def main():
    dst = sys.argv[1]
    n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rec = ca.load_network(os.path.join(HERE, "..", "trees", "sycamore_m20_w32.json"))
    inputs = [tuple(t) for t in rec["inputs"]]
    t0 = time.time()

    def progress(kind, seed, tree, total):
        print(f"{kind} seed {seed}: 2^{math.log2(tree.nslices):.0f} slices, 10^{tree.contraction_cost(log=10):.3f} MACs, "
              f"{total / 86400:.2f} days modelled ({time.time() - t0:.0f}s)", flush=True)

    tree = pathfind.search(inputs, tuple(rec["output"]), rec["size_dict"], target_size=2**32,
                           n_samples=n_samples, seed=0, workers=workers, refine_top=3, progress=progress)
    secs, arena = pathfind.modelled_seconds(tree)
    out = {k: rec[k] for k in ("source", "inputs", "output", "size_dict") if k in rec}
    out["path"] = [list(p) for p in tree.get_path()]
    out["sliced_inds"] = list(tree.sliced_inds)
    out["search"] = {
        "optimizer": f"tests/golden/gen/search_native.py (cotengra_amd.pathfind.search, n_samples={n_samples}, "
        "seed=0, refine_top=3, target 2^32, model MI355X_C64)",
        "seconds": round(time.time() - t0),
        "workers": workers,
    }
    out["stats"] = {
        "nslices_log2": math.log2(tree.nslices),
        "contraction_cost_log10": tree.contraction_cost(log=10),
        "cost_per_slice": tree.contraction_cost() // tree.nslices,
        "write_per_slice": tree.total_write() // tree.nslices,
        "max_size_log2": tree.max_size(log=2),
        "model_ms_per_slice": secs * 1e3,
        "arena_gib": arena / 2**30,
    }
    with open(dst, "w") as f:
        json.dump(out, f, ensure_ascii=False)
    print(f"final: {secs * 1e3:.1f} ms/slice, {secs * tree.nslices / 86400:.2f} days, arena {arena / 2**30:.0f} GiB")


if __name__ == "__main__":
    main()
