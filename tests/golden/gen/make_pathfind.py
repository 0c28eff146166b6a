"""Quality fixtures for the native path finder / slicer (build container only).

Runs the REAL reference's greedy optimizer (temperature 0, costmod 1: the
deterministic algorithm the native ``ctg_path_greedy`` restates) and its
``ContractionTree.slice`` on a few networks and freezes the resulting costs.
Only numbers are stored; the networks are regenerated from seeds with the
reference's own generators (lattice / random-regular / random hyper).

  PYTHONPATH=oracle/refshim:/root/reference python tests/golden/gen/make_pathfind.py
"""
import json
import math
import os

import cotengra as ctg

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def networks():
    yield "lattice8x8_d4", ctg.utils.lattice_equation([8, 8], d_min=4, d_max=4, seed=0)
    yield "lattice6x6x2_d2", ctg.utils.lattice_equation([6, 6, 2], d_min=2, d_max=2, seed=1)
    yield "randreg100_d3", ctg.utils.randreg_equation(100, 3, d_min=2, d_max=2, seed=2)
    yield "randreg200_d3_var", ctg.utils.randreg_equation(200, 3, d_min=2, d_max=3, seed=3)
    c = ctg.utils.rand_equation(60, 3, n_out=2, n_hyper_in=4, n_hyper_out=1, d_min=2, d_max=4, seed=4)
    yield "hyper60", (c.inputs, c.output, c.shapes, c.size_dict)
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m10.json"), encoding="utf-8"))
    yield "sycamore_m10", ([tuple(t) for t in rec["inputs"]], tuple(rec["output"]), None, rec["size_dict"])


def main():
    cases = []
    for name, (inputs, output, _, size_dict) in networks():
        inputs = [tuple(t) for t in inputs]
        output = tuple(output)
        path = ctg.pathfinders.path_basic.optimize_greedy(
            inputs, output, size_dict, costmod=1.0, temperature=0.0, simplify=False, use_ssa=True
        )
        tree = ctg.ContractionTree.from_path(inputs, output, size_dict, ssa_path=path)
        width = tree.max_size(log=2)
        target = max(2 ** int(width - 6), 2 ** 8)
        sliced = tree.slice(target_size=target, seed=0) if width > math.log2(target) else tree
        rf = tree.subtree_reconfigure(subtree_size=8, minimize="flops", seed=0)
        rc = tree.subtree_reconfigure(subtree_size=8, minimize="combo-256", seed=0)
        cases.append({
            "name": name,
            "inputs": [list(t) for t in inputs],
            "output": list(output),
            "size_dict": size_dict,
            # the deterministic greedy path itself (temperature 0): integer work, compared
            # index for index with the native finder
            "ref_greedy_ssa_path": [sorted(map(int, p)) for p in path],
            "ref_greedy_log10_flops": tree.contraction_cost(log=10),
            "ref_greedy_log2_width": width,
            "slice_target": target,
            "ref_sliced_log10_flops": sliced.contraction_cost(log=10),
            "ref_sliced_log2_nslices": math.log2(sliced.nslices),
            "ref_sliced_log2_width": sliced.max_size(log=2),
            "ref_reconf8_flops_log10_flops": rf.contraction_cost(log=10),
            "ref_reconf8_combo256_log10_cost": math.log10(rc.contraction_cost() + 256 * rc.total_write()),
        })
        print(name, {k: v for k, v in cases[-1].items() if k.startswith("ref") or k == "slice_target"})
    out = os.path.join(ROOT, "tests", "golden", "pathfind_cases.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump({"reference": "jcmgray/cotengra v0.8.2", "cases": cases}, f, ensure_ascii=False)
    print("->", out)


if __name__ == "__main__":
    main()
