"""Offline tree search with the REAL reference (build container only).

Runs cotengra's own (unchanged, host-side, pure-Python) hyper-optimizer +
dynamic slicing on a benchmark network and stores the result in the
reference's own persistence format `{path, sliced_inds}`
(reference cotengra/hyperoptimizers/hyper.py:1075-1096) as a small JSON
fixture.  The pathfinder is OUT OF SCOPE for the MI355X build (it
"stays on the host CPU unchanged"); only its *output* (a tree) is an input to
our executor, so only the output is committed.

usage:
  PYTHONPATH=oracle/refshim:/root/reference python tests/golden/gen/search_tree.py \
      <network.json> <out.json> <log2 target_size> [max_repeats] [max_time_s] [minimize]
"""
import json
import sys
import time


def main():
    import cotengra as ctg

    src, dst, log2size = sys.argv[1], sys.argv[2], int(sys.argv[3])
    max_repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    max_time = float(sys.argv[5]) if len(sys.argv) > 5 else 600.0
    minimize = sys.argv[6] if len(sys.argv) > 6 else "combo"

    inputs, output, size_dict = ctg.utils.load_from_json(src)
    inputs = [tuple(t) for t in inputs]
    output = tuple(output)

    opt = ctg.HyperOptimizer(
        methods=["greedy", "labels"],
        minimize=minimize,
        max_repeats=max_repeats,
        max_time=max_time,
        parallel=8,
        optlib="sbplx",
        slicing_reconf_opts={"target_size": 2**log2size},
        reconf_opts={"subtree_size": 8},
        progbar=False,
    )
    t0 = time.time()
    tree = opt.search(inputs, output, size_dict)
    dt = time.time() - t0

    rec = {
        "source": src.split("/")[-1],
        "inputs": [list(t) for t in inputs],
        "output": list(output),
        "size_dict": size_dict,
        "path": [list(map(int, p)) for p in tree.get_path()],
        "sliced_inds": list(tree.sliced_inds),
        "search": {
            "optimizer": "HyperOptimizer(greedy+labels, sbplx)",
            "minimize": minimize,
            "target_size_log2": log2size,
            "max_repeats": max_repeats,
            "seconds": dt,
        },
        "stats": {
            "nslices_log2": __import__("math").log2(tree.nslices),
            "contraction_cost_log10": tree.contraction_cost(log=10),
            "cost_per_slice": tree.contraction_cost() // tree.nslices,
            "write_per_slice": tree.total_write() // tree.nslices,
            "max_size_log2": tree.max_size(log=2),
            "peak_size_log2": tree.peak_size(log=2),
        },
    }
    with open(dst, "w", encoding="utf-8") as f:
        json.dump(rec, f, ensure_ascii=False)
    print(json.dumps({k: rec[k] for k in ("search", "stats")}, indent=1))


if __name__ == "__main__":
    main()
