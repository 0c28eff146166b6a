"""Offline tree refinement with the REAL reference (build container only).

Takes a tree fixture `{inputs, output, size_dict, path, sliced_inds}` and
improves it with cotengra's own host-side tools, unchanged:

* ``ContractionTree.subtree_reconfigure`` (reference core.py:2283) under a
  memory-aware objective ``combo-<f>`` (flops + f * write, scoring.py): on an
  MI355X a complex64 step is HBM-bound below ~25 MACs per element moved, so the
  right factor is much larger than the CPU-oriented default 64;
* ``ContractionTree.slice_and_reconfigure`` (core.py:2723) to re-slice the
  unsliced tree to a target width that suits 288 GB of HBM.

The pathfinder is OUT OF SCOPE for the MI355X build (it "stays on the host CPU
unchanged"); only its *output* is committed as a fixture.

usage:
  PYTHONPATH=oracle/refshim:/root/reference python tests/golden/gen/reconf_tree.py \
      <in.json> <out.json> <minimize e.g. combo-512> <subtree_size> [--reslice LOG2_WIDTH] [--rounds N]
"""
import json
import math
import sys
import time


def main():
    import cotengra as ctg

    src, dst, minimize, sub = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    reslice = int(sys.argv[sys.argv.index("--reslice") + 1]) if "--reslice" in sys.argv else None
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2

    rec = json.load(open(src, encoding="utf-8"))
    inputs = [tuple(t) for t in rec["inputs"]]
    output = tuple(rec["output"])
    tree = ctg.ContractionTree.from_path(
        inputs, output, rec["size_dict"], path=[tuple(p) for p in rec["path"]]
    )

    def stats(t, tag):
        print(
            tag,
            "total flops 10^%.3f flops/slice %.4g write/slice %.4g max_size 2^%.1f nslices 2^%.0f"
            % (math.log10(t.contraction_cost()), t.contraction_cost() / t.nslices,
               t.total_write() / t.nslices, t.max_size(log=2), math.log2(t.nslices)),
            flush=True,
        )

    t0 = time.time()
    log = []
    if reslice is None:
        for ix in rec["sliced_inds"]:
            tree.remove_ind_(ix)
        stats(tree, "start")
    else:
        stats(tree, "unsliced")
        tree = tree.subtree_reconfigure(subtree_size=sub, minimize=minimize, maxiter=2000)
        stats(tree, "unsliced, reconfigured")
        tree = tree.slice_and_reconfigure(
            target_size=2**reslice, step_size=2, minimize=minimize, max_repeats=16,
            reconf_opts=dict(subtree_size=sub, maxiter=500),
        )
        stats(tree, "sliced")
        log.append(f"slice_and_reconfigure(target_size=2**{reslice}, minimize={minimize!r}, subtree_size={sub})")
    for r in range(rounds):
        tree = tree.subtree_reconfigure(subtree_size=sub, minimize=minimize, maxiter=3000, seed=r)
        stats(tree, f"round {r} ({time.time() - t0:.0f}s)")
    log.append(f"subtree_reconfigure(subtree_size={sub}, minimize={minimize!r}, maxiter=3000) x{rounds}")

    out = dict(rec)
    out["path"] = [list(map(int, p)) for p in tree.get_path()]
    out["sliced_inds"] = list(tree.sliced_inds)
    out["search"] = dict(rec.get("search", {}))
    out["search"]["refined"] = out["search"].get("refined", []) + log
    out["stats"] = {
        "nslices_log2": math.log2(tree.nslices),
        "contraction_cost_log10": tree.contraction_cost(log=10),
        "cost_per_slice": tree.contraction_cost() // tree.nslices,
        "write_per_slice": tree.total_write() // tree.nslices,
        "max_size_log2": tree.max_size(log=2),
        "peak_size_log2": tree.peak_size(log=2),
    }
    with open(dst, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)


if __name__ == "__main__":
    main()
