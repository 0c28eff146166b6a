"""Shared loader for the committed golden fixtures (tests/golden/)."""
import json
import os

import numpy as np

import cotengra_amd as ca

HERE = os.path.dirname(os.path.abspath(__file__))
_CASES = None
_EXPECTED = None


def load():
    global _CASES, _EXPECTED
    if _CASES is None:
        with open(os.path.join(HERE, "golden", "golden_cases.json"), encoding="utf-8") as f:
            _CASES = json.load(f)["cases"]
        _EXPECTED = np.load(os.path.join(HERE, "golden", "golden_expected.npz"))
    return _CASES, _EXPECTED


_R2 = None


def load_r2():
    """Round-2 fixtures (tests/golden/gen/make_golden_r2.py): projected output
    indices."""
    global _R2
    if _R2 is None:
        with open(os.path.join(HERE, "golden", "golden_r2_cases.json"), encoding="utf-8") as f:
            cs = json.load(f)["cases"]
        _R2 = (cs, np.load(os.path.join(HERE, "golden", "golden_r2_expected.npz")))
    return _R2


def cases(kind=None):
    cs, _ = load()
    return [c for c in cs if kind is None or c["kind"] == kind]


def expected(key):
    _, ex = load()
    return ex[key]


def tree_of(case):
    inputs = [tuple(t) for t in case["inputs"]]
    output = tuple(case["output"])
    if len(inputs) > 1:
        tree = ca.ContractionTree.from_path(inputs, output, case["size_dict"], path=case["path"])
    else:
        tree = ca.ContractionTree(inputs, output, case["size_dict"])
    for ind, project in case["sliced"]:
        tree.remove_ind_(ind, project=project)
    return tree


def arrays_of(case, dtype, tree=None):
    inputs = [tuple(t) for t in case["inputs"]] if tree is None else tree.inputs
    return ca.make_arrays_from_inputs(
        inputs, case["size_dict"], seed=case["seed"], dtype=dtype, rescale=case.get("rescale", False)
    )


def eq_tree_and_arrays(case, dtype):
    inputs, output = ca.eq_to_inputs_output(case["eq"])
    tree = ca.array_contract_tree(inputs, output, case["size_dict"],
                                  optimize=case["path"] if len(inputs) > 1 else "greedy")
    arrays = ca.make_arrays_from_inputs(inputs, case["size_dict"], seed=case["seed"], dtype=dtype)
    return tree, arrays


def relerr(x, ref):
    x, ref = np.asarray(x), np.asarray(ref)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    scale = max(float(np.abs(ref).max()) if ref.size else 0.0, 1e-300)
    return float(np.abs(x - ref).max() / scale) if ref.size else 0.0


NORTH_STAR = 1e-5


def single_gate(ref, numpy_single):
    """Tolerance (relative to max|ref|) of a single-precision device result:
    the north-star 1e-5, or -- where numpy itself, running the same contraction
    in the same single precision, is further than 1.25e-6 from the double
    precision reference -- eight times numpy's own error.  One rule for every
    single-precision comparison in the GPU suite (tree level since round 2,
    kernel-shape tests since round 3)."""
    if numpy_single is None:
        return NORTH_STAR
    return max(NORTH_STAR, 8.0 * relerr(np.asarray(numpy_single), ref))


def stem_flags(name):
    """Template arguments of a ``stem2_kernel<...>`` instantiation as the executor spells them
    (csrc/ctg_stem.hip: stem2_kernel_name): chunks and items known at compile time (0: the
    run-time-count variant), B2 in registers, bf16 x 3 products, row-interleaved step 2, single step,
    and -- three-step tiles only -- the middle stage's items per wave and whether it has 16 columns."""
    a = [x.strip() for x in name[name.index("<") + 1 : name.rindex(">")].split(",")]
    t = [x == "true" for x in a]
    return {"pack1": t[0], "pack2": t[1], "rt1": int(a[2]), "cs1": int(a[3]), "nch": int(a[4]), "it2": int(a[5]),
            "br1": t[6], "k2q": int(a[7]), "vec": t[8], "bf3": t[9], "ri2": t[10], "one": t[11],
            "itm": int(a[12]) if len(a) > 12 else 0, "packm": t[13] if len(t) > 13 else False,
            # round 5: two-accumulator real parts / the intermediate as bf16 limbs
            "xm": t[14] if len(t) > 14 else False, "lm": t[15] if len(t) > 15 else False,
            "ws": t[16] if len(t) > 16 else False}   # (specialised waves)


def stem_network(nq, gates, seed, sliced=0):
    """A small 'stem': one tensor of ``nq`` binary indices to which tensors are
    applied one after the other, gate ``(k, n)`` contracting ``k`` randomly chosen
    indices of the running tensor and adding ``n`` fresh ones (the structure of a
    sliced Sycamore contraction).  Returns ``(tree, inputs)``
    with the tree contracting the chain in order and ``sliced`` of the first
    tensor's indices sliced."""
    rng = np.random.default_rng(seed)
    cur = [f"a{i}" for i in range(nq)]
    inputs = [tuple(cur)]
    for g, (kin, nout) in enumerate(gates):
        sel = sorted(int(i) for i in rng.choice(len(cur), size=kin, replace=False))
        new = [f"g{g}_{j}" for j in range(nout)]
        inputs.append(tuple(cur[i] for i in rng.permutation(sel)) + tuple(new))
        cur = [c for i, c in enumerate(cur) if i not in sel] + new
    # the open indices in a scrambled order: the root's layout is the caller's
    output = tuple(cur[i] for i in rng.permutation(len(cur)))
    size_dict = {ix: 2 for t in inputs for ix in t}
    ssa, cur_id, nxt = [], 0, len(inputs)
    for i in range(1, len(inputs)):
        ssa.append((cur_id, i))
        cur_id, nxt = nxt, nxt + 1
    tree = ca.ContractionTree.from_path(inputs, output, size_dict, ssa_path=ssa)
    for ix in inputs[0][:sliced]:
        tree.remove_ind_(ix)
    return tree


# (nq, gates): single stem steps -- a large step no pair takes runs on the stem kernel's first half
ONE_CASES = [
    (17, [(3, 3), (5, 5)]),                       # k32 n32
    (17, [(3, 3), (7, 5)]),                       # k128 n32
    (17, [(3, 3), (6, 6)]),                       # k64 n64: two waves per row tile
    (18, [(3, 3), (7, 7)]),                       # k128 n128: four waves per row tile
    (17, [(3, 3), (4, 5)]),                       # k16 n32: 512-row tiles
    (17, [(3, 3), (5, 6)]),                       # k32 n64
    (17, [(3, 3), (5, 5), (6, 6), (5, 5)]),       # a chain of odd length: a pair and a step left over
]


def random_stem(seed):
    """A random stem for the fused-pair tests: 3 to 6 gates of 4 to 7 contracted and 4 to 7 new
    indices on a tensor of 15 to 18 binary indices, up to two indices sliced."""
    rng = np.random.default_rng(1000 + seed)
    nq = int(rng.integers(15, 19))
    gates, cur = [(3, 3)], nq
    for _ in range(int(rng.integers(3, 7))):
        kin = int(rng.integers(4, 8))
        nout = int(np.clip(kin + rng.integers(-2, 3), 4, 7))
        if kin > cur - 6 or cur - kin + nout > 19:
            continue
        gates.append((kin, nout))
        cur += nout - kin
    return stem_network(nq, gates, seed, sliced=int(rng.integers(0, 3)))


# (nq, gates): every instantiation of the fused kernel -- 16 / 32 columns on either step,
# 256 and 512 tile rows, K up to 128, N2 up to 128 -- and chains with leftovers
STEM_CASES = [
    (16, [(3, 3), (5, 5), (5, 5)]),                    # k32 n32 | k32 n32
    (16, [(3, 3), (4, 4), (5, 5)]),                    # k16 n16 | k32 n32: 16 columns first, 512 rows
    (16, [(3, 3), (5, 5), (4, 4)]),                    # k32 n32 | k16 n16: 16 columns last
    (16, [(3, 3), (4, 4), (4, 4)]),                    # k16 n16 | k16 n16
    (17, [(3, 3), (5, 5), (6, 6)]),                    # k32 n32 | k64 n64
    (17, [(3, 3), (7, 5), (5, 5)]),                    # k128 n32 | k32 n32
    (17, [(3, 3), (5, 5), (5, 6)]),                    # k32 n32 | k32 n64
    (17, [(3, 3), (6, 5), (6, 6)]),                    # k64 n32 | k64 n64
    (17, [(3, 3), (4, 5), (5, 7)]),                    # k16 n32 | k32 n128
    (18, [(3, 3), (6, 4), (5, 4)]),                    # k64 n16 | k32 n16
    (17, [(3, 3), (5, 5), (5, 5), (4, 4), (5, 5), (6, 6), (5, 5)]),   # a longer stem: pairs + leftovers
    (17, [(3, 3), (6, 6), (5, 5)]),                    # k64 n64 | k32 n32: two waves per row tile
    (17, [(3, 3), (6, 6), (4, 4)]),                    # k64 n64 | k16 n16
    (18, [(3, 3), (4, 6), (6, 6)]),                    # k16 n64 | k64 n64
    (17, [(3, 3), (5, 7), (5, 5)]),                    # k32 n128 | k32 n32: four waves per row tile
    (18, [(3, 3), (4, 7), (5, 4)]),                    # k16 n128 | k32 n16
]
