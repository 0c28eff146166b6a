"""Build the Sycamore m10 amplitude network from the reference's qsim circuit
with OUR circuit front end, let the reference's hyper-optimizer find a tree
sliced into >= 64 slices, and freeze network + tensor values + tree + the
reference's results (build container only).

    PYTHONPATH=oracle/refshim:/root/reference python tests/golden/gen/make_m10.py

Outputs: tests/golden/trees/sycamore_m10.json  (inputs, output, size_dict, path, sliced_inds)
         tests/golden/sycamore_m10_arrays.npz  (the 170 gate tensors, complex128)
         tests/golden/sycamore_m10_expected.npz (full amplitude + slice partials by the reference)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, ROOT)
import cotengra as ctg  # noqa: E402  (the reference)

import cotengra_amd as ca  # noqa: E402
from cotengra_amd.circuits import circuit_to_network, parse_qsim  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

QSIM = "/root/reference/examples/circuit_n53_m10_s0_e0_pABCDCDAB.qsim"


def main():
    n, gates = parse_qsim(open(QSIM).read())
    inputs, output, size_dict, arrays = circuit_to_network(n, gates)
    print(len(inputs), "tensors")
    opt = ctg.HyperOptimizer(
        methods=["greedy", "labels"], minimize="combo", max_repeats=64, max_time=600, parallel=8,
        optlib="sbplx", slicing_opts={"target_slices": 64}, reconf_opts={"subtree_size": 8},
        progbar=False)
    rt = opt.search(inputs, output, size_dict)
    print(rt, "nslices", rt.nslices, "width", rt.max_size(log=2), "cost/slice", rt.contraction_cost() // rt.nslices)
    rec = {
        "source": "circuit_n53_m10_s0_e0_pABCDCDAB.qsim via cotengra_amd.circuits (bitstring 0...0)",
        "inputs": [list(t) for t in inputs], "output": [], "size_dict": size_dict,
        "path": [list(map(int, p)) for p in rt.get_path()], "sliced_inds": list(rt.sliced_inds),
        "stats": {"nslices": int(rt.nslices), "max_size_log2": rt.max_size(log=2),
                  "cost_per_slice": int(rt.contraction_cost() // rt.nslices)},
    }
    json.dump(rec, open(os.path.join(ROOT, "tests/golden/trees/sycamore_m10.json"), "w"), ensure_ascii=False)
    np.savez_compressed(os.path.join(ROOT, "tests/golden/sycamore_m10_arrays.npz"),
                        **{f"t{i}": a for i, a in enumerate(arrays)})
    # reference results: the full amplitude (all slices) and a few partials, complex128
    exp = {}
    ids = sorted({0, 1, rt.nslices // 2, rt.nslices - 1})
    for i in ids:
        exp[f"slice{i}"] = np.asarray(rt.contract_slice(arrays, i))
    exp["amplitude"] = np.asarray(rt.contract(arrays))
    mt = ca.tree_from_record(rec)
    assert abs(orc.contract_slice(mt, arrays, 1) - exp["slice1"]) <= 1e-12 * abs(exp["slice1"])
    np.savez_compressed(os.path.join(ROOT, "tests/golden/sycamore_m10_expected.npz"), **exp)
    print("amplitude", exp["amplitude"], "|a|^2 * 2^53 =", abs(exp["amplitude"]) ** 2 * 2.0**53)


if __name__ == "__main__":
    main()
