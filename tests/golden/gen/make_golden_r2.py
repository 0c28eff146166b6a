"""Round-2 additions to the golden fixtures, made with the REAL reference
(build container only):

    PYTHONPATH=oracle/refshim:/root/reference python tests/golden/gen/make_golden_r2.py

* trees whose OUTPUT index is projected onto one value (``remove_ind(ix,
  project=j)`` with ``ix`` in the output): the reference's ``contract`` keeps a
  size-1 axis there (``gather_slices`` stacks over ``sliced_range = [j]``,
  core.py:3866-3876);
* ``(mantissa, exponent)`` pairs of ``strip_exponent=True`` contractions, as
  the plain numbers ``mantissa * 10**exponent``;

into tests/golden/golden_r2_cases.json + tests/golden/golden_r2_expected.npz
(same record layout as make_golden.py: index lists, sizes, path, slicing, seed
-- data only).  The oracle is pinned on every case while generating.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)

import cotengra as ctg  # noqa: E402  (the reference)

import cotengra_amd as ca  # noqa: E402
from oracle import contract_ref as orc  # noqa: E402

CASES, EXPECTED = [], {}


def add(name, inputs, output, size_dict, path, sliced, seed):
    rt = ctg.ContractionTree.from_path(inputs, output, size_dict, path=path)
    mt = ca.ContractionTree.from_path(inputs, output, size_dict, path=path)
    for ix, project in sliced:
        rt.remove_ind_(ix, project=project)
        mt.remove_ind_(ix, project=project)
    arrays = ctg.utils.make_arrays_from_inputs(inputs, size_dict, seed=seed, dtype="complex128")
    mine = ca.make_arrays_from_inputs(inputs, size_dict, seed=seed, dtype="complex128")
    assert all(np.array_equal(a, b) for a, b in zip(arrays, mine))
    x = np.asarray(rt.contract(arrays))
    y = np.asarray(orc.contract(mt, arrays))
    assert x.shape == y.shape == mt.gathered_shape(), (name, x.shape, y.shape)
    assert np.allclose(x, y, rtol=1e-12, atol=1e-15)
    EXPECTED[f"{name}/complex128"] = x
    for i in range(min(rt.nslices, 3)):
        xi = np.asarray(rt.contract_slice(arrays, i))
        assert np.allclose(xi, np.asarray(orc.contract_slice(mt, arrays, i)), rtol=1e-12, atol=1e-15)
        EXPECTED[f"{name}/complex128/slice{i}"] = xi
    CASES.append({
        "name": name, "kind": "tree", "inputs": [list(t) for t in inputs], "output": list(output),
        "size_dict": dict(size_dict), "path": [list(map(int, p)) for p in rt.get_path()],
        "sliced": [[si.ind, si.project] for si in rt.sliced_inds.values()], "seed": seed,
        "rescale": False, "dtypes": ["complex128"], "slice_ids": list(range(min(rt.nslices, 3))),
        "stats": {"nslices": int(rt.nslices)},
    })
    print(name, x.shape, rt.nslices)


def main():
    # chain with a projected output index, with and without further slicing
    inputs = [("a", "b"), ("b", "c"), ("c", "d"), ("d", "e")]
    sd = dict(a=3, b=4, c=5, d=6, e=4)
    path = [(0, 1), (0, 1), (0, 1)]
    add("projout_chain", inputs, ("a", "e"), sd, path, [("e", 2)], 11)
    add("projout_chain_inner", inputs, ("a", "e"), sd, path, [("e", 1), ("c", None)], 12)
    add("projout_chain_both_outer", inputs, ("a", "e"), sd, path, [("e", 3), ("a", None), ("d", None)], 13)
    # a lattice with open legs, one projected, one sliced, one hyper-free inner slice
    li, lo, _, lsd = ctg.utils.lattice_equation([3, 3], d_min=2, d_max=3, seed=3)
    li = [list(t) for t in li]
    li[0].append("X"), li[8].append("Y"), li[4].append("Z")
    lsd = dict(lsd, X=3, Y=4, Z=2)
    lout = ("X", "Y", "Z")
    rt = ctg.ContractionTree.from_path(li, lout, lsd, path=ctg.array_contract_path(li, lout, lsd, optimize="greedy"))
    add("projout_lattice", [tuple(t) for t in li], lout, lsd, rt.get_path(),
        [("Y", 2), ("X", None), (li[4][0], None)], 14)
    json.dump({"cases": CASES}, open(os.path.join(HERE, "..", "golden_r2_cases.json"), "w"), indent=0)
    np.savez_compressed(os.path.join(HERE, "..", "golden_r2_expected.npz"), **EXPECTED)


if __name__ == "__main__":
    main()
