"""Round-4 fixtures made with the REAL reference (build container only):

    PYTHONPATH=oracle/refshim:/root/reference python tests/golden/gen/make_golden_r4.py

``ContractionTree.sort_contraction_indices`` (reference core.py:3421-3506): for a set of
trees (lattices, random-regular and hyper networks, greedy and random-pair trees, some
sliced, the 8x8 lattice of config C2, Sycamore m10) and every combination of the method's
options that changes anything, the reference's index order of EVERY node afterwards
(``get_inds``, in ``traverse()`` order) and its linear IR ``extract_contractions(tree)``
(sha256 of its sorted repr strings, the format of ``golden_cases.json``), plus one two-call sequence with
``reset=False`` (the second call starts from what the first one left).  Only index lists
and strings are stored (``tests/golden/sorted_inds_cases.json``);
``tests/test_host_round4.py`` holds ``cotengra_amd.ContractionTree`` to them exactly, and
for a few small cases the contraction result after sorting is frozen too
(``sorted_inds_expected.npz``: the value does not depend on the order -- the GPU test
contracts the sorted tree through ``array_contract_expression(...,
sort_contraction_indices=True)`` and through the per-op plug-in, whose tensordot axes /
perms ARE the sorted ones).
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)

import cotengra as ctg  # noqa: E402  (the reference)
from cotengra.contract import extract_contractions as ref_extract  # noqa: E402

CONFIGS = [
    # (priority, make_output_contig, make_contracted_contig)
    ("flops", True, True),
    ("size", True, True),
    ("root", True, True),
    ("leaves", True, True),
    ("flops", True, False),
    ("flops", False, True),
    ("leaves", False, True),
    ("root", True, False),
]


def ir_repr(ops):
    """sha256 over the sorted repr strings of the IR's tuples (the strings themselves are in
    golden_cases.json's format; here the index orders are stored in full and the IR derived from
    them only has to be identical)."""
    import hashlib

    return hashlib.sha256("\n".join(sorted(repr(tuple(op)) for op in ops)).encode()).hexdigest()


def networks():
    for seed in range(3):
        yield f"lattice4x4_s{seed}", ctg.utils.lattice_equation([4, 4], d_min=2, d_max=3, seed=seed)
    for seed in range(2):
        yield f"lattice3x3x3_s{seed}", ctg.utils.lattice_equation([3, 3, 3], d_min=2, d_max=2, seed=seed)
    for seed in range(3):
        yield f"randreg30_s{seed}", ctg.utils.randreg_equation(30, 3, d_min=2, d_max=3, seed=seed)
    for seed in range(3):
        c = ctg.utils.rand_equation(24, 3, n_out=2, n_hyper_in=2, n_hyper_out=1, d_min=2, d_max=3, seed=seed)
        yield f"hyper24_s{seed}", (c.inputs, c.output, c.shapes, c.size_dict)
    yield "lattice8x8_d4", ctg.utils.lattice_equation([8, 8], d_min=4, d_max=4, seed=0)
    c = ctg.utils.rand_equation(10, 3, n_out=2, d_min=4, d_max=4, seed=0)
    yield "C1_rand10", (c.inputs, c.output, c.shapes, c.size_dict)
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m10.json"), encoding="utf-8"))
    yield "sycamore_m10", ([tuple(t) for t in rec["inputs"]], tuple(rec["output"]), None, rec["size_dict"])


def node_inds(tree):
    out = []
    for p, l, r in tree.traverse():
        out.append([list(tree.get_inds(p)), list(tree.get_inds(l)), list(tree.get_inds(r))])
    return out


def main():
    cases, expected = [], {}
    for k, (name, (inputs, output, _, size_dict)) in enumerate(networks()):
        inputs = [tuple(t) for t in inputs]
        output = tuple(output)
        if k % 2 == 0 or len(inputs) > 40:
            ssa = ctg.pathfinders.path_basic.optimize_greedy(
                inputs, output, size_dict, costmod=1.0, temperature=0.0, simplify=False, use_ssa=True
            )
        else:
            rng = random.Random(100 + k)
            live, ssa, nxt = list(range(len(inputs))), [], len(inputs)
            while len(live) > 1:
                i, j = sorted(rng.sample(range(len(live)), 2))
                ssa.append((live[i], live[j]))
                live.pop(j)
                live.pop(i)
                live.append(nxt)
                nxt += 1

        def fresh():
            t = ctg.ContractionTree.from_path(inputs, output, size_dict, ssa_path=ssa)
            if k % 3 == 0 and t.max_size() > 64:
                t.slice_(target_slices=4, seed=0)
            return t

        tree = fresh()
        sliced = [[si.ind, si.project] for si in tree.sliced_inds.values()]
        rec = {
            "name": name,
            "inputs": [list(t) for t in inputs],
            "output": list(output),
            "size_dict": size_dict,
            "ssa_path": [list(map(int, p)) for p in tree.get_ssa_path()],
            "sliced": sliced,
            "default": {"inds": node_inds(tree), "ir": ir_repr(ref_extract(tree))},
            "sorted": [],
        }
        changed = 0
        for priority, moc, mcc in CONFIGS:
            t = fresh()
            t.sort_contraction_indices(priority=priority, make_output_contig=moc, make_contracted_contig=mcc)
            got = {"priority": priority, "make_output_contig": moc, "make_contracted_contig": mcc,
                   "inds": node_inds(t), "ir": ir_repr(ref_extract(t))}
            changed += got["inds"] != rec["default"]["inds"]
            rec["sorted"].append(got)
        # a sequence: 'size' first, then 'leaves' WITHOUT reset (starts from the first one's orders)
        t = fresh()
        t.sort_contraction_indices(priority="size")
        t.get_einsum_eq(t.root) if not t.get_can_dot(t.root) else t.get_tensordot_axes(t.root)
        t.sort_contraction_indices(priority="leaves", make_output_contig=False, reset=False)
        rec["sequence"] = {"inds": node_inds(t)}
        # frozen results of small cases (complex128), contracted by the reference AFTER sorting
        if tree.contraction_cost() < 5e7:
            t = fresh()
            t.sort_contraction_indices()
            arrays = ctg.utils.make_arrays_from_inputs(inputs, size_dict, seed=7, dtype="complex128")
            x = np.asarray(t.contract(arrays))
            y = np.asarray(fresh().contract(arrays))
            assert np.allclose(x, y, rtol=1e-10, atol=1e-300), name
            expected[name] = x
            rec["seed"] = 7
        cases.append(rec)
        print(name, tree.N, sliced, "configs that changed an order:", changed, "of", len(CONFIGS))
    out = os.path.join(ROOT, "tests", "golden", "sorted_inds_cases.json")
    with open(out, "w", encoding="utf-8") as f:
        json.dump({"reference": "jcmgray/cotengra v0.8.2", "configs": CONFIGS, "cases": cases}, f, ensure_ascii=False)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sorted_inds_expected.npz"), **expected)
    print("->", out, len(cases), "trees;", len(expected), "results")


if __name__ == "__main__":
    main()
