"""Sycamore m10 with 8 OPEN output qubits: one contraction = the batch of 256 amplitudes
<b1..b8 0...0| C |0...0> (open output qubits give the pairwise steps
a real N dimension).  Network by cotengra_amd.circuits from the reference's qsim file, tree by
this package's host search (cotengra_amd.pathfind), a few slice partials and the first
amplitudes pinned by the REAL reference (build container only):

    PYTHONPATH=oracle/refshim:/root/reference python tests/golden/gen/make_m10_open.py

Outputs: tests/golden/trees/sycamore_m10_open8.json, tests/golden/sycamore_m10_open8_arrays.npz,
         tests/golden/sycamore_m10_open8_expected.npz
"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, ROOT)
import cotengra as ctg  # noqa: E402  (the reference)

import cotengra_amd as ca  # noqa: E402
from cotengra_amd import pathfind  # noqa: E402
from cotengra_amd.circuits import circuit_to_network, parse_qsim  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

QSIM = "/root/reference/examples/circuit_n53_m10_s0_e0_pABCDCDAB.qsim"
N_OPEN = 8


def main():
    n, gates = parse_qsim(open(QSIM).read())
    bits = "?" * N_OPEN + "0" * (n - N_OPEN)
    inputs, output, size_dict, arrays = circuit_to_network(n, gates, bitstring=bits)
    print(len(inputs), "tensors,", len(output), "open indices")
    t0 = time.time()
    tree = pathfind.search(inputs, output, size_dict, target_size=2**24, n_samples=48, seed=0, workers=6,
                           refine_top=2)
    print("tree: 2^%.0f slices, width 2^%.1f, 10^%.2f MACs (%.0fs)" % (
        math.log2(tree.nslices), tree.max_size(log=2), tree.contraction_cost(log=10), time.time() - t0))
    rec = {
        "source": f"circuit_n53_m10_s0_e0_pABCDCDAB.qsim via cotengra_amd.circuits, first {N_OPEN} qubits open",
        "inputs": [list(t) for t in inputs], "output": list(output), "size_dict": size_dict,
        "path": [list(map(int, p)) for p in tree.get_path()], "sliced_inds": list(tree.sliced_inds),
        "stats": {"nslices": int(tree.nslices), "max_size_log2": tree.max_size(log=2),
                  "cost_per_slice": int(tree.contraction_cost() // tree.nslices)},
    }
    json.dump(rec, open(os.path.join(ROOT, "tests/golden/trees/sycamore_m10_open8.json"), "w"), ensure_ascii=False)
    np.savez_compressed(os.path.join(ROOT, "tests/golden/sycamore_m10_open8_arrays.npz"),
                        **{f"t{i}": a for i, a in enumerate(arrays)})
    # the reference on the same tree: slice partials (full 256-amplitude tensors of a slice)
    rt = ctg.ContractionTree.from_path(inputs, output, size_dict, path=rec["path"])
    for ix in rec["sliced_inds"]:
        rt.remove_ind_(ix)
    assert rt.nslices == tree.nslices
    exp = {}
    for i in sorted({0, 1, rt.nslices - 1}):
        exp[f"slice{i}"] = np.asarray(rt.contract_slice(arrays, i))
        mine = np.asarray(orc.contract_slice(tree, arrays, i))
        assert np.allclose(mine, exp[f"slice{i}"], rtol=1e-11, atol=1e-14)
    # amplitude of the all-zero bitstring = the committed m10 amplitude (another tree, another network)
    ref0 = np.load(os.path.join(ROOT, "tests/golden/sycamore_m10_expected.npz"))["amplitude"]
    t0 = time.time()
    full = np.asarray(rt.contract(arrays))
    print("reference contracted all slices in %.0fs" % (time.time() - t0))
    assert abs(full.reshape(-1)[0] - ref0) <= 1e-10 * abs(ref0), (full.reshape(-1)[0], ref0)
    exp["amplitudes"] = full
    np.savez_compressed(os.path.join(ROOT, "tests/golden/sycamore_m10_open8_expected.npz"), **exp)
    print("amplitudes", full.shape, "sum |a|^2 * 2^45 =", (abs(full) ** 2).sum() * 2.0**45)


if __name__ == "__main__":
    main()
