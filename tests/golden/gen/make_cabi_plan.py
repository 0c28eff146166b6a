"""A compiled plan + inputs + expected result as one flat binary file, for the plain-C
driver of the C ABI (tests/cabi_reduce.c):

    python tests/golden/gen/make_cabi_plan.py        ->  tests/golden/cabi_plan.bin

Workload: the golden case ``lattice4x4_sliced`` (reference tests/test_backends.py:105-119
shape; 4 slices) in complex128; the expected result is the reference-frozen golden value.
Layout (little-endian int64 words, then doubles):
  header[10] = magic 0x43544750, dtype, n_inputs, inputs_elems, arena_elems, result_elems,
               n_steps, n_table_words, n_sliced, nslices
  input_sizes[n_inputs] input_offsets[n_inputs] steps[n_steps * 48] tables[n_table_words]
  slice_sizes[n_sliced] slice_fixed[n_sliced] slice_strides[(n_inputs + 1) * n_sliced]
  inputs: sum(input_sizes) complex128 values (re, im doubles), input after input
  expected: result_elems complex128 values
tests/test_cabi.py checks that the committed file is what this script writes today.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_util as G  # noqa: E402
from cotengra_amd.plan import compile_tree  # noqa: E402

MAGIC = 0x43544750


def build():
    case = next(c for c in G.cases("tree") if c["name"] == "lattice4x4_sliced")
    tree = G.tree_of(case)
    arrays = G.arrays_of(case, "complex128", tree)
    plan = compile_tree(tree, "complex128", _group=())   # (the driver passes no slice_group array: no slice groups)
    ser = plan.serialise()
    n_in, n_sl = len(plan.input_sizes), len(plan.slice_sizes)
    head = np.array([MAGIC, ser["dtype"], n_in, plan.inputs_elems, ser["arena_elems"], ser["result_elems"],
                     ser["n_steps"], ser["tables"].size, n_sl, plan.nslices], dtype="<i8")
    words = [head, ser["input_sizes"], ser["input_offsets"], ser["steps"], ser["tables"], ser["slice_sizes"],
             ser["slice_fixed"], ser["slice_strides"]]
    blob = b"".join(np.ascontiguousarray(w, dtype="<i8").tobytes() for w in words)
    blob += b"".join(np.ascontiguousarray(a, dtype="<c16").tobytes() for a in arrays)
    expected = np.asarray(G.expected("lattice4x4_sliced/complex128"), dtype="<c16").reshape(-1)
    assert expected.size == ser["result_elems"]
    return blob + expected.tobytes()


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "cabi_plan.bin")
    with open(out, "wb") as f:
        f.write(build())
    print("->", out, os.path.getsize(out), "bytes")
