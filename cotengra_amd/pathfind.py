"""Host-side tree tools backed by the native library (no GPU needed):

* ``greedy_ssa_path`` / ``greedy_path`` -- the reference's ``optimize_greedy``
  (``cotengra/pathfinders/path_basic.py:1038-1106``; its inner loop is what the
  optional ``cotengrust`` accelerator replaces, ``:1351-1383``);
* ``random_greedy_tree`` -- repeated Boltzmann-sampled greedy runs, best tree
  kept (the idea of ``optimize_random_greedy_track_flops``, ``:1113-1240``);
* ``slice_tree`` -- greedy choice of sliced indices until the largest
  intermediate fits ``target_size`` (``SliceFinder``, ``cotengra/slicer.py``);
* ``subtree_reconfigure`` / ``slice_and_reconfigure`` -- exact re-ordering of
  subtrees by dynamic programming (``core.py:2316-2449, 2723-2808``), under the
  reference's objectives or under a machine model (``minimize="time"``);
* ``modelled_seconds`` / ``unslice`` / ``refine`` -- polishing of a sliced tree
  for the device, guided by the plan's own step list priced by the model;
* ``sample_sliced_tree`` / ``search`` -- a stand-alone search built from all of
  the above (many cheap, very unequal draws; the best ones refined).

The reference's hyper-optimizers (partition-based builders, Bayesian parameter
tuning) are not restated; this module is what the stand-alone front ends use
when they are not handed a tree.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import runtime
from .tree import ContractionTree


def _csr(inputs, output, size_dict):
    ids = {}
    for term in list(inputs) + [output]:
        for ix in term:
            if ix not in ids:
                ids[ix] = len(ids)
    offsets = np.zeros(len(inputs) + 1, dtype=np.int64)
    flat = []
    for t, term in enumerate(inputs):
        flat.extend(ids[ix] for ix in term)
        offsets[t + 1] = len(flat)
    flat = np.asarray(flat, dtype=np.int64) if flat else np.zeros(1, dtype=np.int64)
    out = np.asarray([ids[ix] for ix in output], dtype=np.int64) if len(output) else np.zeros(1, dtype=np.int64)
    sizes = np.empty(max(len(ids), 1), dtype=np.float64)
    for ix, i in ids.items():
        sizes[i] = float(size_dict[ix])
    return ids, offsets, flat, out, sizes


def _p(a, typ):
    return a.ctypes.data_as(C.POINTER(typ))


def greedy_ssa_path(inputs, output, size_dict, costmod=1.0, temperature=0.0, max_neighbors=16, seed=0):
    """SSA path ``[(i, j), ...]`` found by the native greedy algorithm."""
    n = len(inputs)
    if n < 2:
        return ()
    ids, offsets, flat, out, sizes = _csr(inputs, output, size_dict)
    path = np.empty(2 * (n - 1), dtype=np.int64)
    runtime._check(
        runtime.load().ctg_path_greedy(
            n, _p(offsets, C.c_int64), _p(flat, C.c_int64), len(output), _p(out, C.c_int64), len(ids),
            _p(sizes, C.c_double), float(costmod), float(temperature), int(max_neighbors or 0),
            int(seed) & (2**64 - 1), _p(path, C.c_int64),
        )
    )
    return tuple((int(path[2 * s]), int(path[2 * s + 1])) for s in range(n - 1))


def greedy_tree(inputs, output, size_dict, **kwargs):
    """``ContractionTree`` of the native greedy path."""
    inputs = [tuple(t) for t in inputs]
    if len(inputs) == 1:
        return ContractionTree(inputs, tuple(output), size_dict)
    return ContractionTree.from_path(
        inputs, tuple(output), size_dict, ssa_path=greedy_ssa_path(inputs, output, size_dict, **kwargs)
    )


def greedy_path(inputs, output, size_dict, **kwargs):
    """Linear (recycled-id) path, the ``opt_einsum`` convention."""
    return greedy_tree(inputs, output, size_dict, **kwargs).get_path()


def random_greedy_tree(inputs, output, size_dict, repeats=32, costmod=(0.1, 4.0), temperature=(0.001, 1.0),
                       minimize="flops", seed=0, max_neighbors=16):
    """Best of ``repeats`` sampled greedy trees.  ``minimize``: ``"flops"``,
    ``"size"`` or ``"combo-<f>"`` (flops + f * write, the reference's memory
    aware objective)."""
    rng = np.random.default_rng(seed)

    def cost(tree):
        if minimize == "flops":
            return tree.contraction_cost()
        if minimize == "size":
            return tree.max_size()
        if minimize.startswith("combo"):
            f = float(minimize.split("-")[1]) if "-" in minimize else 64.0
            return tree.contraction_cost() + f * tree.total_write()
        raise ValueError(f"unknown objective {minimize!r}")

    best = greedy_tree(inputs, output, size_dict, max_neighbors=max_neighbors)
    best_cost = cost(best)
    lo_t, hi_t = np.log(temperature[0]), np.log(temperature[1])
    for _ in range(repeats):
        tree = greedy_tree(
            inputs, output, size_dict, costmod=float(rng.uniform(*costmod)),
            temperature=float(np.exp(rng.uniform(lo_t, hi_t))), seed=int(rng.integers(0, 2**63)),
            max_neighbors=max_neighbors,
        )
        c = cost(tree)
        if c < best_cost:
            best, best_cost = tree, c
    return best


def find_sliced_inds(tree, target_size, allow_outer=True):
    """Indices to remove so that no intermediate of ``tree`` exceeds
    ``target_size`` elements (already sliced indices are taken into account)."""
    inputs = tree.get_inputs_sliced() if tree.sliced_inds else tree.inputs
    output = tree.get_output_sliced() if tree.sliced_inds else tree.output
    n = len(inputs)
    if n < 2:
        return ()
    ids, offsets, flat, out, sizes = _csr(inputs, output, tree.size_dict)
    ssa = np.asarray([x for pair in tree.get_ssa_path() for x in pair], dtype=np.int64)
    sliced = np.empty(max(len(ids), 1), dtype=np.int64)
    n_sliced = C.c_int64(0)
    runtime._check(
        runtime.load().ctg_slice_greedy(
            n, _p(offsets, C.c_int64), _p(flat, C.c_int64), len(output), _p(out, C.c_int64), len(ids),
            _p(sizes, C.c_double), _p(ssa, C.c_int64), float(np.log2(target_size)), int(bool(allow_outer)),
            len(ids), _p(sliced, C.c_int64), C.byref(n_sliced),
        )
    )
    names = {i: ix for ix, i in ids.items()}
    return tuple(names[int(sliced[q])] for q in range(n_sliced.value))


def slice_tree(tree, target_size, allow_outer=True, inplace=False):
    """``tree`` with indices removed until it fits ``target_size`` (reference
    ``ContractionTree.slice``, core.py:2632-2719, with a greedy finder)."""
    tree = tree if inplace else tree.copy()
    for ix in find_sliced_inds(tree, target_size, allow_outer=allow_outer):
        tree.remove_ind_(ix)
    return tree


def _objective(minimize):
    """write factor of an objective name: ``"flops"`` -> 0, ``"combo"`` -> 64,
    ``"combo-<f>"`` -> f (reference scoring.py)."""
    if minimize in (None, "flops"):
        return 0.0
    if minimize == "combo":
        return 64.0
    if isinstance(minimize, str) and minimize.startswith("combo-"):
        return float(minimize.split("-", 1)[1])
    raise ValueError(f"unknown objective {minimize!r} (flops, combo, combo-<f>)")


class MachineModel:
    """What a contraction step costs on the device, as the native reconfiguration
    prices it under ``minimize=<MachineModel>`` / ``minimize="time"``:
    ``max(MACs / mac_rate[floor(log2 K)], elements moved / elem_rate)`` seconds,
    ``K`` = contracted extent (csrc/ctg_pathfind.cpp: CostModel)."""

    def __init__(self, mac_rate_by_log2k, elem_rate):
        self.mac_rate_by_log2k = tuple(float(x) for x in mac_rate_by_log2k)
        self.elem_rate = float(elem_rate)

    def step_seconds(self, macs, elems, k, n=16):
        rate = self.mac_rate_by_log2k[min(max(int(math.floor(math.log2(max(k, 1)))), 0), len(self.mac_rate_by_log2k) - 1)]
        if n < 16:
            rate *= max(n, 1) / 16.0
        elif n < 64 and k >= 64:
            rate *= 0.8
        return max(macs / rate, elems / self.elem_rate)


# complex64 on one MI355X, from the per-step tables of this package's kernels
# (profiles/r1_m20_steps.txt): complex MACs/s (= TFLOP/s / 8) of pairwise steps
# by contracted extent K = 1, 2, 4, ... 512+, and 4.8 TB/s of 8-byte elements.
MI355X_C64 = MachineModel(
    [x * 1e12 / 8 for x in (4, 8, 16, 30, 50, 75, 100, 118, 127, 131)], 4.8e12 / 8
)


# The same machine once consecutive stem steps run fused (stem.py; round 3): a memory-bound
# step of a pair moves the big tensor once instead of twice, matrix work runs at ~0.73 of peak
# whatever K from 16 up.  Only an objective for the subtree search (the dynamic programme
# prices single steps and cannot know which will pair up); `modelled_seconds` prices the plan
# the executor really builds.
MI355X_C64_FUSED = MachineModel(
    [x * 1e12 / 8 for x in (4, 8, 16, 31, 95, 110, 115, 115, 127, 131)], 8.0e12 / 8
)


def subtree_reconfigure(tree, subtree_size=8, maxiter="auto", minimize="flops", inplace=False):
    """Locally optimal re-ordering of the subtrees of ``tree`` (reference
    ``ContractionTree.subtree_reconfigure``, core.py:2316-2449) by the native
    dynamic-programming routine.  Sliced indices are kept and count as size 1,
    so the per-slice cost is what is minimised.  ``minimize``: ``"flops"``,
    ``"combo"``, ``"combo-<f>"`` as in the reference, or ``"time"`` / a
    :class:`MachineModel` -- modelled seconds on the device."""
    if tree.N < 3:
        return tree if inplace else tree.copy()
    sliced = dict(tree.sliced_inds)
    size_dict = dict(tree.size_dict)
    eff = {ix: (1 if ix in tree.sliced_inds else d) for ix, d in size_dict.items()}
    ids, offsets, flat, out, sizes = _csr(tree.inputs, tree.output, eff)
    ssa_in = np.asarray([x for pair in tree.get_ssa_path() for x in pair], dtype=np.int64)
    ssa_out = np.empty_like(ssa_in)
    lib = runtime.load()
    head = (
        tree.N, _p(offsets, C.c_int64), _p(flat, C.c_int64), len(tree.output), _p(out, C.c_int64),
        len(ids), _p(sizes, C.c_double), _p(ssa_in, C.c_int64), int(subtree_size),
        0 if maxiter == "auto" else int(maxiter),
    )
    model = MI355X_C64 if minimize == "time" else minimize
    if isinstance(model, MachineModel):
        rates = np.asarray(model.mac_rate_by_log2k, dtype=np.float64)
        runtime._check(
            lib.ctg_subtree_reconfigure_timed(
                *head, _p(rates, C.c_double), len(rates), model.elem_rate, _p(ssa_out, C.c_int64)
            )
        )
    else:
        runtime._check(lib.ctg_subtree_reconfigure(*head, _objective(minimize), _p(ssa_out, C.c_int64)))
    new = ContractionTree.from_path(
        tree.inputs, tree.output, size_dict,
        ssa_path=[(int(ssa_out[2 * s]), int(ssa_out[2 * s + 1])) for s in range(tree.N - 1)],
    )
    new.apply_slicing_(sliced)  # (projections kept)
    if inplace:
        tree.__dict__.update(new.__dict__)
        return tree
    return new


def slice_and_reconfigure(tree, target_size, minimize="flops", subtree_size=8, allow_outer=True, step_bits=2.0):
    """Interleave slicing with subtree reconfiguration (reference
    ``ContractionTree.slice_and_reconfigure``, core.py:2723-2808): slice towards
    ``target_size`` ``step_bits`` at a time, re-optimising the subtrees for
    the sliced network after every step."""
    tree = subtree_reconfigure(tree, subtree_size=subtree_size, minimize=minimize)
    while tree.max_size() > target_size:
        step = max(float(target_size), tree.max_size() / 2.0**step_bits)
        tree = slice_tree(tree, step, allow_outer=allow_outer)
        tree = subtree_reconfigure(tree, subtree_size=subtree_size, minimize=minimize)
    return tree


def modelled_seconds(tree, model=None, dtype="complex64"):
    """``(seconds per slice, arena bytes)`` of ``tree`` as the executor would run
    it: the device plan's steps (their real K, N, MACs and bytes, slice-invariant
    steps included) priced by ``model`` (default :data:`MI355X_C64`)."""
    from .plan import KIND_STEM2, compile_tree
    from .stem import pair_seconds, single_seconds

    model = MI355X_C64 if model is None else model
    plan = compile_tree(tree, dtype)
    itemsize = plan.itemsize
    t = 0.0
    share = 1.0 / plan.group_size   # (a step shared by a group of slices costs a slice its share)
    for s in plan.steps:
        if not s.macs:
            continue
        t += step_seconds(s, model) * (share if s.group else 1.0)
    return t, plan.arena_elems * itemsize


def step_seconds(s, model=None):
    """Modelled time of one step of a device plan (what ``modelled_seconds`` adds up)."""
    from .plan import KIND_STEM2
    from .stem import pair_seconds, single_seconds

    model = MI355X_C64 if model is None else model
    t = 0.0
    if not s.macs:
        return t
    for _ in (0,):
        if s.kind == KIND_STEM2:
            # a fused stem pair (stem.py): its own model -- the big tensor moves once
            st = s.stem
            if st.get("one"):   # (a single step on the stem kernel's first half)
                t += single_seconds(s.macs, s.a.size, s.c.size, st["run_bytes"], bf3_fits=st.get("bf3_fits", True))
                continue
            if st.get("KM"):    # (a three-step tile: opt-in, stem.triples_enabled)
                from .stem import triple_seconds

                t += triple_seconds(st["macs3"], (st["N1"], st["NM"], st["N2"]), s.a.size, s.c.size, st["run_bytes"])
                continue
            macs1 = (s.a.size // st["K1"]) * st["K1"] * st["N1"]
            t += pair_seconds(macs1, s.macs - macs1, s.a.size, s.c.size, st["items"], st["run_bytes"],
                              bf3_fits=st.get("bf3_fits", True))
        else:
            t += model.step_seconds(s.macs, s.elems_rw, s.K, s.N)
    return t


def unslice(tree, model=None, max_width=2**32, max_arena_bytes=160 * 2**30):
    """Take indices out of the slicing again, each time the one that lowers the
    modelled time to the full result most, while the largest intermediate stays
    within ``max_width`` elements and the arena within ``max_arena_bytes``."""
    while True:
        cur = modelled_seconds(tree, model)[0] * tree.nslices
        best = None
        for ix in list(tree.sliced_inds):
            cand = tree.restore_ind(ix)
            if cand.max_size() > max_width:
                continue
            t, arena = modelled_seconds(cand, model)
            if arena > max_arena_bytes:
                continue
            if best is None or t * cand.nslices < best[0]:
                best = (t * cand.nslices, cand)
        if best is None or best[0] >= cur:
            return tree
        tree = best[1]


def refine(tree, objectives=("time", "combo-64", "combo-128"), subtree_sizes=(8, 10, 12, 14), model=None,
           max_width=2**32, max_arena_bytes=160 * 2**30, max_rounds=8, progress=None):
    """Polish a (sliced) tree for the device: sweep every objective x subtree size
    through :func:`subtree_reconfigure` followed by :func:`unslice`, keep a
    candidate whenever the modelled time to the full result drops, repeat until a
    whole sweep brings nothing.  On the Sycamore-53 m20 benchmark tree this takes
    five minutes and cuts the time to the amplitude fourfold
    (tests/golden/gen/refine_native.py)."""
    best = modelled_seconds(tree, model)[0] * tree.nslices
    for rnd in range(max_rounds):
        improved = False
        for obj in objectives:
            for sz in subtree_sizes:
                cand = subtree_reconfigure(tree, subtree_size=sz, minimize=obj)
                if cand.max_size() > max_width:   # the new order may be wider: slice it back
                    cand = slice_tree(cand, max_width)
                cand = unslice(cand, model, max_width, max_arena_bytes)
                secs, arena = modelled_seconds(cand, model)
                v = secs * cand.nslices
                if arena <= max_arena_bytes and v < best * (1 - 1e-6):
                    best, tree, improved = v, cand, True
                    if progress is not None:
                        progress(rnd, obj, sz, tree, v)
        if not improved:
            break
    return tree


def sample_sliced_tree(inputs, output, size_dict, target_size, seed=0, repeats=64, minimize="combo-64"):
    """One draw of the native search: the best of ``repeats`` sampled greedy
    trees, reconfigured (subtree sizes 10 and 12), then sliced to ``target_size``
    one bit at a time with reconfiguration in between.  About 13 s for the
    381-tensor Sycamore m20 network; the outcome varies over four decades of
    total work with the seed, which is what :func:`search` exploits."""
    tree = random_greedy_tree(inputs, output, size_dict, repeats=repeats, minimize="flops", seed=seed)
    for sz in (10, 12):
        tree = subtree_reconfigure(tree, subtree_size=sz, minimize=minimize)
    return slice_and_reconfigure(tree, target_size, minimize=minimize, subtree_size=10, step_bits=1.0)


def _search_worker(args):
    inputs, output, size_dict, target_size, seed, repeats, minimize = args
    tree = sample_sliced_tree(inputs, output, size_dict, target_size, seed, repeats, minimize)
    return seed, [tuple(p) for p in tree.get_path()], tree.slicing_record()


def search(inputs, output, size_dict, target_size=2**32, n_samples=64, seed=0, workers=1, refine_top=1,
           model=None, max_arena_bytes=160 * 2**30, repeats=64, minimize="combo-64", progress=None):
    """Stand-alone search for a sliced contraction tree (no reference optimizer
    involved): ``n_samples`` independent draws of :func:`sample_sliced_tree`
    (seeds ``seed .. seed + n_samples - 1``, on ``workers`` processes), ranked by
    the modelled time to the full result; the best ``refine_top`` are polished with
    :func:`refine` and the winner is returned.  Deterministic for given arguments."""
    inputs = [tuple(t) for t in inputs]
    output = tuple(output)
    jobs = [(inputs, output, dict(size_dict), target_size, seed + i, repeats, minimize) for i in range(n_samples)]
    if workers and workers > 1:
        import concurrent.futures as cf

        with cf.ProcessPoolExecutor(max_workers=workers) as pool:
            results = list(pool.map(_search_worker, jobs))
    else:
        results = [_search_worker(j) for j in jobs]
    ranked = []
    for sd, path, sliced in results:
        tree = ContractionTree.from_path(inputs, output, size_dict, path=path).apply_slicing_(sliced)
        secs, arena = modelled_seconds(tree, model)
        ranked.append((secs * tree.nslices, sd, tree))
        if progress is not None:
            progress("sample", sd, tree, secs * tree.nslices)
    ranked.sort(key=lambda r: (r[0], r[1]))
    best = None
    for total, sd, tree in ranked[: max(1, refine_top)]:
        tree = refine(tree, model=model, max_width=target_size, max_arena_bytes=max_arena_bytes)
        total = modelled_seconds(tree, model)[0] * tree.nslices
        if progress is not None:
            progress("refined", sd, tree, total)
        if best is None or total < best[0]:
            best = (total, tree)
    return best[1]
