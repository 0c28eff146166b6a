#!/usr/bin/env python
"""bench.py -- sliced Sycamore-53 m20 amplitude contraction on MI355X.

One "step" = one slice of the m20 contraction tree per GPU (SURVEY.md section 8d:
unit = one slice).  The headline tree is the one that reaches the full amplitude
FIRST on this machine (``sycamore_m20_native.json``: 2^20 slices, found by this
package's host-side search, tests/golden/gen/search_native.py) -- since round 3; before,
the line was quoted on ``sycamore_m20_w32_c512.json`` (the reference optimizer's tree,
refined), whose slices run at a higher FLOP/s but which needs 3.6x as long for the
amplitude: it stays in the line as ``peak_rate_tree``.  ``--tree`` takes any other
fixture; inputs are synthetic tensors of the named shapes
(reference ``make_arrays_from_inputs`` semantics, seed 42, complex64, rescaled
by size**0.25 so fp32 does not underflow) and are resident in HBM before the
timed region.

``--gpus N``: one process per GPU.  Started without a launcher (no RANK in the
environment) the script re-executes itself under ``torch.distributed.run`` with
N ranks on 127.0.0.1; started by a launcher it checks that WORLD_SIZE == N.
Every rank pins GPU LOCAL_RANK, the ranks verify that they own N distinct GPUs,
each contracts ITS SHARE of the slices (``ctg_exec_run_share``: whole slice groups ``rank, rank + world,
...`` -- the reference's round-robin, core.py:4070, with the group as the unit; no data-path traffic) and the
partial amplitudes are combined by ONE RCCL reduce on the executors' streams
(``ctg_exec_reduce`` of the C ABI) inside the timed region -- weak scaling.

Reports (rank 0): whole-node contracted FLOP/s, the dominant
kernel's roofline numbers measured live with HIP events, the numpy-oracle CPU
baseline on this node's host cores (rank 0, every N; the launcher's
OMP_NUM_THREADS=1 is lifted for it) and -- at N = 1 -- ``peak_rate_tree`` and the
other BASELINE.json configurations (``configs``: C2 8x8 lattice, C3 Sycamore m10,
C5 hyper network), each with its own mixed per-step roofline and CPU-oracle time;
at N > 1 ``configs.C3_amplitudes``: Sycamore m10 amplitudes of different
bitstrings per second, the unit of that configuration that shards, and
``configs.C3_strong``: the ONE 64-slice amplitude dealt over the N ranks + reduce, as
BASELINE.json words the configuration (a 3 ms job does not strong-scale; it is printed
anyway).  The LAST stdout line is a compact record (< 4 KB: the contract's keys, ``roofline``,
``cpu_baseline``, one number per extra leg); the full record is written to
``bench_full.json`` (``bench_full_n<N>.json`` at N > 1; also under gpurun_out/).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_MFMA_BF16_TFLOPS = 2500.0  # ... dense bf16 matrix peak (~2.5 PF)
# fp32 products as SIX bf16 products (three-way split operands, csrc/ctg_stem.hip BF3): the peak of
# that arithmetic in fp32-equivalent flops -- what the bf16 x 3 legs are priced against
PEAK_BF16X3_TFLOPS = PEAK_MFMA_BF16_TFLOPS / 6.0
# (round 6) two fp16 limbs, three products per fp32 product on v_mfma_f32_32x32x16_f16 (same dense peak as bf16)
PEAK_FP16X2_TFLOPS = PEAK_MFMA_BF16_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0
TREES = os.path.join(ROOT, "tests", "golden", "trees")
TREE = os.path.join(TREES, "sycamore_m20_native.json")        # reaches the amplitude first
PEAK_TREE = os.path.join(TREES, "sycamore_m20_w32_c512.json")  # highest FLOP/s per slice (r1 / r2 headline)
# the headline tree refined for an executor that fuses stem pairs (tests/golden/gen/refine_fused.py,
# round 3) and once more under the model of the executor as it is since round 4 (single stem steps,
# bf16 x 3 products: gen/refine_r4.py): a quarter less work in smaller, memory-bound steps -- fewer
# FLOP/s, but the amplitude 14 % sooner
TTS_TREE = os.path.join(TREES, "sycamore_m20_w32_r4.json")
# the same with one index less sliced: 2^19 slices of width 2^33 (68 GB tensors, a 161 GiB arena --
# what 288 GB of HBM are for); the amplitude another 6 % sooner
TTS33_TREE = os.path.join(TREES, "sycamore_m20_w33_bf3.json")   # (refined once more: gen/refine_bf3.py)
# `fused` refined under the model WITH slice groups (gen/refine_r4.py at the very end of round 4, after the
# round's GPU budget was spent): modelled 177.5 ms per slice against 189.0 for w32_r4 (which measures 185-192);
# reported as its own leg until it has been measured next to it
TTS_GROUPS_TREE = os.path.join(TREES, "sycamore_m20_w32_g.json")


# ---------------------------------------------------------------------- #
# helpers shared by all workloads
# ---------------------------------------------------------------------- #


def shrink_for_cpu(tree, log2_width):
    """Slice further indices (largest tensor first) until one slice fits a
    CPU-sized budget; the schedule is unchanged, so MACs/s stays comparable."""
    tree = tree.copy()
    while tree.max_size() > 2**log2_width:
        big = max((p for p, _, _ in tree.traverse()), key=tree.get_size)
        ix = next(iter(tree.get_legs(big)))
        tree.remove_ind_(ix)
    return tree


class host_threads:
    """Give numpy's BLAS the node's cores for a CPU leg: ``torch.distributed.run``
    exports OMP_NUM_THREADS=1 to every rank, which would throttle rank 0's oracle."""

    def __enter__(self):
        self.ctx = None
        try:
            from threadpoolctl import threadpool_limits

            self.ctx = threadpool_limits(limits=host_cores())
            self.ctx.__enter__()
        except Exception:  # noqa: BLE001  (no threadpoolctl: the leg runs with what it has)
            self.ctx = None
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def blas_threads():
    try:
        from threadpoolctl import threadpool_info

        return max([int(d.get("num_threads", 1)) for d in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        return None


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count()


def cpu_baseline(tree, arrays, budget_s=20.0, log2_width=24):
    """Time the numpy oracle (oracle/contract_ref.py, a port of the reference
    executor) on slices of the same tree narrowed to 2^log2_width."""
    from oracle import contract_ref as orc

    small = shrink_for_cpu(tree, log2_width)
    macs = small.contraction_cost() // small.nslices
    ops = orc.extract_contractions(small)
    with host_threads():
        threads = blas_threads()
        t0 = time.time()
        n = 0
        while True:
            orc.run_contractions(ops, orc.slice_arrays(small, arrays, n))
            n += 1
            if time.time() - t0 > budget_s or n >= 64:
                break
        dt = time.time() - t0
    return {
        "value": 8.0 * macs * n / dt,
        "unit": "FLOP/s",
        "cores": host_cores(),
        "blas_threads": threads,
        "kind": "port",
        "sample": (
            f"{n} slices of the same m20 tree narrowed to width 2^{log2_width} "
            f"({macs:.3e} complex MACs/slice), numpy complex64, {dt:.1f}s"
        ),
    }


def precision_check(tree, arrays, slice_id=3, log2_width=20):
    """One slice of ``tree`` narrowed to a CPU-sized width, HIP vs the numpy
    complex128 oracle.  The complex128 HIP path must reproduce the oracle to
    1e-10 (same schedule in double precision); the complex64 production path is
    gated at max(1e-5, 8 x the error numpy itself makes in complex64) -- 1e-5 is
    the north-star tolerance, and a heavily cancelling slice sum cannot be asked
    to beat single-precision arithmetic by more than its rounding noise."""
    from oracle import contract_ref as orc

    small = shrink_for_cpu(tree, log2_width)
    # (with the bf16 switch on, the narrowed tree must run fused pairs too -- they start at 2^24
    # elements by default, above anything a CPU-sized slice has)
    from cotengra_amd.stem import bf16x3_mode

    lowered = bf16x3_mode() and "CTG_FUSE_MIN_ELEMS" not in os.environ
    if lowered:
        os.environ["CTG_FUSE_MIN_ELEMS"] = str(1 << 12)
    # (fp16 x 2: on the full-width tree every pair of a stem but its first runs it; the narrowed tree's stems are one or
    # two pairs long, so the check asks for EVERY capable pair in fp16 x 2 -- CTG_STEM_H2_ALL: a max-abs pass scales a first pair)
    h2_all = stem_arithmetic() == "fp16x2" and "CTG_STEM_H2_ALL" not in os.environ
    if h2_all:
        os.environ["CTG_STEM_H2_ALL"] = "1"
    a128 = [a.astype("complex128") for a in arrays]
    with host_threads():
        ref = complex(orc.contract_slice(small, a128, slice_id))
        np64 = complex(orc.contract_slice(small, arrays, slice_id))
    got = complex(np.asarray(small.contract_slice(arrays, slice_id)))
    got128 = complex(np.asarray(small.contract_slice(a128, slice_id)))
    rel = abs(got - ref) / abs(ref)
    rel_np = abs(np64 - ref) / abs(ref)
    gate = max(1e-5, 8.0 * rel_np)
    out = {
        "check": f"slice {slice_id} of the tree narrowed to width 2^{log2_width}, vs numpy complex128",
        "rel_err": rel,
        "gate": gate,
        "meets_north_star_1e-5": bool(rel <= 1e-5),
        "numpy_complex64_rel_err": rel_np,
        "rel_err_complex128_path": abs(got128 - ref) / abs(ref),
        "gate_complex128_path": 1e-10,
    }
    if h2_all:
        del os.environ["CTG_STEM_H2_ALL"]
        out["check"] += "; every capable pair in fp16 x 2"
    if lowered:
        del os.environ["CTG_FUSE_MIN_ELEMS"]
        out["check"] += "; stem pairs fused from 2^12 elements, on the bf16 matrix cores"
        out["fused_pairs_in_check"] = sum(
            1 for c in small.contraction_cores.values() for pl, _ in getattr(c, "_plans", {}).values()
            if pl.dtype == "complex64" for s_ in pl.steps if s_.kind == 3)
    for c in list(small.contraction_cores.values()):
        c.close()
    if out["rel_err"] > gate or out["rel_err_complex128_path"] > 1e-10:
        raise SystemExit(f"precision check failed: {out}")
    return out


def precision_sum_check(tree, arrays, slice_id=3, log2_width=22, log2_slices=12):
    """The quantity ``north_star`` gates is a SUM of slices (core.py:3842-3844).  One slice of ``tree``
    narrowed to width 2^log2_width is -- exactly -- the sum of the 2^log2_slices slices of the same tree with
    that many more indices sliced: the complex64 HIP path sums those on the device (double-precision running
    sum, accum_kernel), the numpy complex128 oracle contracts the one wide slice.  No numpy-relative clause:
    the number is reported against 1e-5 as it is."""
    import itertools

    from cotengra_amd.contractor import HipContractor
    from cotengra_amd.stem import bf16x3_mode
    from oracle import contract_ref as orc

    coarse = shrink_for_cpu(tree, log2_width)
    fine = coarse.copy()
    extra = []
    while len(extra) < log2_slices:
        big = max((p for p, _, _ in fine.traverse()), key=fine.get_size)
        ix = next(iter(fine.get_legs(big)))
        fine.remove_ind_(ix)
        extra.append(ix)
    # ids (in fine's numbering) of the slices of ``fine`` that make up slice ``slice_id`` of ``coarse``
    import cotengra_amd as ca

    key = coarse.slice_key(slice_id)
    strides = dict(zip(fine.sliced_inds, ca.get_slice_strides(fine.sliced_inds)))
    ids = []
    for combo in itertools.product(*[range(fine.size_dict[ix]) for ix in extra]):
        k = dict(key)
        k.update(zip(extra, combo))
        ids.append(sum(k[ix] * strides[ix] for ix in fine.sliced_inds))
    ids.sort()
    lowered = bf16x3_mode() and "CTG_FUSE_MIN_ELEMS" not in os.environ
    if lowered:
        os.environ["CTG_FUSE_MIN_ELEMS"] = str(1 << 12)
    h2_all = stem_arithmetic() == "fp16x2" and "CTG_STEM_H2_ALL" not in os.environ
    if h2_all:   # (as in precision_check: every capable pair of the narrowed tree in fp16 x 2)
        os.environ["CTG_STEM_H2_ALL"] = "1"
    a128 = [a.astype("complex128") for a in arrays]
    with host_threads():
        t0 = time.perf_counter()
        ref = complex(orc.contract_slice(coarse, a128, slice_id))
        oracle_s = time.perf_counter() - t0
    fn = HipContractor(fine)
    try:
        st = fn.setup(*arrays)
        ex = st["exec"]
        ex.zero_result()
        ex.run_slice_list(ids)
        got = complex(np.asarray(ex.download_result()))
        wide = str(ex.state_dtype())
        fused = sum(1 for s_ in st["plan"].steps if s_.kind == 3)
    finally:
        fn.close()
        if lowered:
            del os.environ["CTG_FUSE_MIN_ELEMS"]
        if h2_all:
            del os.environ["CTG_STEM_H2_ALL"]
    rel = abs(got - ref) / abs(ref)
    return {
        "check": f"sum of {len(ids)} complex64 slices (width 2^{math.log2(fine.max_size()):.0f}, device running sum in "
                 f"{wide}) vs ONE numpy complex128 slice of the same tree at width 2^{log2_width}",
        "slices": len(ids), "rel_err": rel, "meets_north_star_1e-5": bool(rel <= 1e-5),
        "fused_pairs_in_check": fused, "oracle_seconds": oracle_s, "every_capable_pair_in_fp16x2": bool(h2_all),
    }


def step_table(ex, plan, slice_id=0):
    """Per-step rows of one slice with HIP-event durations (events on the exec's
    stream, ``ctg_exec_profile_slice``) and kernel names."""
    ms = ex.profile_slice(slice_id)
    rows = plan.describe_steps()
    share = 1.0 / plan.group_size
    for r, m, nm, st in zip(rows, ms, ex.step_kernels(), plan.steps):
        r["ms"] = float(m)
        r["kernel_name"] = nm
        if st.group and plan.group_size > 1:
            # a step the slices of a group share: a slice is charged its share of the time, the work
            # and the bytes (the profiled slice computed it in full: ms_full)
            r["shared"], r["ms_full"] = True, r["ms"]
            for key in ("ms", "macs", "bytes", "bytes_moved"):
                if key in r:
                    r[key] = r[key] * share
    return rows


def is_bf16x3_kernel(name):
    """stem2_kernel<..., BF3, RI2> / stem2h_kernel<...>: the tenth template argument says whether the instantiation
    multiplies on the 16-bit matrix cores (shapes without such an instantiation keep fp32 products)."""
    for prefix in ("stem2_kernel<", "stem2h_kernel<"):
        if name and name.startswith(prefix):
            args = name[len(prefix):].rstrip(">").split(",")
            return len(args) >= 10 and args[9].strip() == "true"
    return False


def step_peak_tflops(r, bf16x3=False):
    """The matrix peak a step is priced against: fp32 MFMA; a fused stem pair on the 16-bit matrix cores: the dense
    bf16 / fp16 peak over its products per fp32 product -- 6 with three bf16 limbs (stem2_kernel), 3 with two fp16
    limbs (stem2h_kernel, round 6) -- decided per launch by the kernel's name when the row carries one."""
    name = r.get("kernel_name") or r.get("kernel_symbol")
    if name is not None and name.startswith("pair_mfma_bf3_kernel"):   # (long tiled steps, round 5)
        return PEAK_BF16X3_TFLOPS
    if name is not None and name.startswith("pair_mfma_h2_kernel"):    # (... in the fp16 x 2 arithmetic, round 6)
        return PEAK_FP16X2_TFLOPS
    if name is not None and name.startswith("stem2h_kernel<"):
        return PEAK_FP16X2_TFLOPS if is_bf16x3_kernel(name) else PEAK_MFMA_F32_TFLOPS
    if name is not None and name.startswith("stem2_kernel<"):
        return PEAK_BF16X3_TFLOPS if is_bf16x3_kernel(name) else PEAK_MFMA_F32_TFLOPS
    if bf16x3 and r.get("kind") == "stem2":
        return PEAK_FP16X2_TFLOPS if bf16x3 == "fp16x2" else PEAK_BF16X3_TFLOPS
    return PEAK_MFMA_F32_TFLOPS


def mixed_roofline_ms(rows, flops_per_mac, moved=True, bf16x3=False):
    """Sum over the per-slice steps of max(F_i / matrix peak, B_i / HBM peak) (SURVEY section 8d
    "mixed, per step").  ``moved=True`` (THE BOUND): B_i = the bytes the plan really moves -- a
    fused stem pair reads its big operand and writes its result, the intermediate never exists,
    and a kernel cannot be asked to beat the traffic it actually has.  ``moved=False``: B_i = the
    algorithmic bytes of the UNFUSED reference steps (every operand read once, every
    result written once per reference step) -- the roofline of the reference's execution model,
    which a fused pair may finish below; reported next to the bound, never as one.
    Slice-invariant steps cost nothing per slice and are excluded (0 ms in the profile)."""
    t = 0.0
    for r in rows:
        if r["ms"] <= 0.0:
            continue
        b = r.get("bytes_moved", r["bytes"]) if moved else r["bytes"]
        peak = step_peak_tflops(r, bf16x3) if moved else PEAK_MFMA_F32_TFLOPS
        t += max(flops_per_mac * r["macs"] / (peak * 1e12), b / (PEAK_HBM_GBS * 1e9))
    return t * 1e3


def dominant_kernel(rows, flops_per_mac):
    by_name = {}
    for r in rows:
        d = by_name.setdefault(r["kernel_name"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "moved": 0.0, "n": 0,
                                                  "kind": r.get("kind"), "kernel_name": r["kernel_name"]})
        d["ms"] += r["ms"]
        d["flops"] += flops_per_mac * r["macs"]
        d["bytes"] += r["bytes"]
        d["moved"] += r.get("bytes_moved", r["bytes"])
        d["n"] += 1 if r["ms"] > 0 else 0
    name, dom = max(by_name.items(), key=lambda kv: kv[1]["ms"])
    return name, dom, by_name


# RI2, ONE, ITM, PACKM, XM, LM, WS of stem2_kernel (template arguments 11-17)
STEM2_TEMPLATE_DEFAULTS = ("false", "false", "0", "false", "false", "false", "false")


def norm_kernel_name(name):
    """One spelling per kernel instantiation: no blanks, and a ``stem2_kernel<...>`` name written
    with its trailing template arguments left at their defaults (the executor's step names,
    csrc/ctg_stem.hip: ctg_stem_kernel_name) padded to the full list rocprof prints."""
    if not name:
        return name
    name = name.replace(" ", "")
    for prefix in ("stem2_kernel<", "stem2h_kernel<"):   # (stem2h: the same instantiations in the fp16 x 2 arithmetic)
        if name.startswith(prefix) and name.endswith(">"):
            targs = name[len(prefix):-1].split(",")
            full = 10 + len(STEM2_TEMPLATE_DEFAULTS)
            if 10 <= len(targs) < full:
                targs += list(STEM2_TEMPLATE_DEFAULTS[len(targs) - 10:])
            name = prefix + ",".join(targs) + ">"
    return name


def pmc_traffic_for(tree_file, kernel):
    """HBM bytes per launch of ``kernel`` from the PMC passes taken on THIS tree
    (profiles/pmc_summary_<tree>.json, written by tools/pmc_traffic.py); None
    when no such pass exists.  Names are compared through ``norm_kernel_name``."""
    tag = os.path.splitext(os.path.basename(tree_file))[0]
    path = os.path.join(ROOT, "profiles", f"pmc_summary_{tag}.json")
    if not os.path.exists(path):
        return None, None
    try:
        pm = json.load(open(path))
        want = norm_kernel_name(kernel)
        kv = next((v for k, v in pm.get("kernels", {}).items()
                   if want in (norm_kernel_name(k), norm_kernel_name(v.get("rocprof_name")))), None)
        return (kv["hbm_bytes_per_launch"] if kv else None), pm.get("hbm_bytes_per_launch")
    except Exception:
        return None, None


ARITH_ENV = ("CTG_STEM_ARITH", "CTG_STEM_BF16X3", "CTG_STEM_H2")


class arithmetic:
    """``with arithmetic("fp32" | "bf16x3" | "fp16x2" | None)``: the fused stem pairs' arithmetic for the duration
    (CTG_STEM_ARITH for the executors created inside; CTG_STEM_BF16X3=0 as well for fp32, which the planner's pairing
    model reads); None = leave the environment alone."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ARITH_ENV}
        if self.mode is not None:
            for k in ARITH_ENV:
                os.environ.pop(k, None)
            os.environ["CTG_STEM_ARITH"] = self.mode
            if self.mode == "fp32":
                os.environ["CTG_STEM_BF16X3"] = "0"
        return self

    def __exit__(self, *exc):
        if self.mode is not None:
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        return False


def stem_arithmetic():
    """The arithmetic an executor created now multiplies its stem pairs with (csrc/ctg_runtime.hip reads the same
    variables): CTG_STEM_BF16X3=0 -> fp32; CTG_STEM_ARITH; fp16 x 2 when nothing is said (CTG_STEM_H2=0: bf16 x 3)."""
    v = os.environ.get("CTG_STEM_BF16X3")
    if v is not None and v in ("", "0"):
        return "fp32"
    a = {"0": "fp32", "1": "bf16x3", "2": "fp16x2"}.get(os.environ.get("CTG_STEM_ARITH", "fp16x2"),
                                                       os.environ.get("CTG_STEM_ARITH", "fp16x2"))
    if a == "fp16x2" and os.environ.get("CTG_STEM_H2") in ("", "0"):
        a = "bf16x3"
    return a


BF16X3_NOTE = ("fused stem pairs: fp32 operands split exactly into 3 bf16 limbs, 6 of the 9 cross terms on "
               "v_mfma_f32_32x32x16_bf16, fp32 accumulation (DESIGN 4b: error bound + adversarial tests); every "
               "other step: fp32 MFMA")
FP16X2_NOTE = ("fused stem pairs: fp32 operands as 2 rounded fp16 limbs under per-tensor power-of-two scales, 3 products on "
               "v_mfma_f32_32x32x16_f16, fp32 accumulation (DESIGN 4.5); long tiled steps likewise (DESIGN 4.3; k-split launches: bf16 x 3); "
               "every other step fp32 MFMA")
FP32_NOTE = "every step on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: an exact-fp32 multiply-add chain)"
ARITH_NOTE = {"fp32": FP32_NOTE, "bf16x3": BF16X3_NOTE, "fp16x2": FP16X2_NOTE}


def time_slices(ex, first, count, stride=1):
    ex.sync()
    t0 = time.perf_counter()
    ex.run_slices(first, count, stride)
    ex.sync()
    return time.perf_counter() - t0


def slice_ids_from_groups(plan, first_group, count, rank=0, world=1):
    """``count`` slice ids for ``rank`` out of ITS SHARE as the library deals it (``Plan.rank_slice_ids`` =
    ``ctg_exec_run_share``: the whole slice groups ``rank, rank + world, ...``; without groups in the plan a
    unit is a single slice), starting at the unit that holds group ``first_group + rank`` -- whole groups, so
    that what a group shares is computed once per group INSIDE the timed region, as it is over the whole job."""
    units, gs = plan.share_units(rank, world)
    if units == 0:
        # (more ranks than slice groups: this rank has nothing to time -- the callers' "every rank times the
        # same number of groups" cannot hold)
        raise ValueError(f"rank {rank} of {world} holds no slice group of this tree")
    u0 = (first_group // world) % max(units, 1)
    need = -(-count // gs)
    ids = []
    while len(ids) < count:
        n = min(need, units - u0)
        if n <= 0:   # (wrapped around a share smaller than the request: start over at its first unit)
            u0, need = 0, -(-(count - len(ids)) // gs)
            n = min(need, units)
        ids += plan.rank_slice_ids(rank, world, u0, n).tolist()
        need -= n
        u0 = 0
    return ids[:count]


def slice_groups_note(plan, ids):
    """What the line says about slice groups (cotengra_amd/plan.py: choose_slice_group)."""
    if plan.group_size <= 1:
        return None
    return {
        "group_indices": len(plan.group_inds),
        "slices_per_group": int(plan.group_size),
        "shared_steps": sum(1 for s in plan.steps if s.group),
        "shared_macs_fraction": plan.macs_shared_per_group / max(plan.macs_per_slice, 1),
        "timed": "%d slices in %d groups" % (len(ids), len({plan.group_of(i) for i in ids})),
        "note": "slices that differ only in the group indices share every step that depends on none of them: "
                "computed for the first slice of a group, kept for the others (CTG_SLICE_GROUPS=0: off)",
    }


def executed_flops(plan, ids, flops_per_mac=8.0):
    """Flops the executor really does for the slices ``ids`` in one call: every step for the first slice
    of each group among them, everything but the steps the group shares for the others."""
    groups = len({plan.group_of(i) for i in ids}) if plan.group_size > 1 else len(ids)
    shared = plan.macs_shared_per_group if plan.group_size > 1 else 0
    return flops_per_mac / 8.0 * (8.0 if plan.is_complex else 2.0) * (
        plan.macs_per_slice * len(ids) - shared * (len(ids) - groups))


# ---------------------------------------------------------------------- #
# the other workloads of the line (rank 0, N = 1)
# ---------------------------------------------------------------------- #


def tree_report(tree_file, dev, steps=5, warmup=1, mode=None):
    """ms/slice, FLOP/s, dominant-kernel and mixed rooflines of another m20 tree.  ``mode``:
    "fp32" / "bf16x3" = with the fused stem pairs in that arithmetic for the duration of the
    report; None = the default (bf16 x 3 since round 4)."""
    if mode is not None:
        with arithmetic(mode):
            return tree_report(tree_file, dev, steps, warmup)
    arith = stem_arithmetic()
    bf16x3 = arith if arith != "fp32" else False
    import torch

    import cotengra_amd as ca
    from cotengra_amd.contractor import HipContractor

    rec = ca.load_network(tree_file)
    tree = ca.tree_from_record(rec)
    arrays = ca.make_arrays_from_inputs(tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True)
    fn = HipContractor(tree, handle_slicing=True)
    st = fn.setup(*[torch.as_tensor(a, device=dev) for a in arrays])
    ex, plan = st["exec"], st["plan"]
    ex.zero_result()
    # (whole slice groups: warm up on the last groups, time the first ones)
    gs = plan.group_size
    steps = gs * max(1, -(-steps // gs))
    n_groups = max(1, plan.nslices // gs)
    ex.run_slice_list(slice_ids_from_groups(plan, n_groups - 1 - (warmup - 1) // gs, warmup))
    ids = slice_ids_from_groups(plan, 0, steps)
    ex.sync()
    t0 = time.perf_counter()
    ex.run_slice_list(ids)
    ex.sync()
    dt = (time.perf_counter() - t0) / steps
    rows = step_table(ex, plan)
    name, dom, _ = dominant_kernel(rows, 8.0)
    flops = executed_flops(plan, ids) / steps
    bound_ms = mixed_roofline_ms(rows, 8.0, moved=True, bf16x3=bf16x3)
    unfused_ms = mixed_roofline_ms(rows, 8.0, moved=False)
    mf = dom["flops"] / max(dom["ms"] * 1e-3, 1e-12) / 1e12
    bw = dom["moved"] / max(dom["ms"] * 1e-3, 1e-12) / 1e9
    dom_peak = step_peak_tflops(dom, bf16x3)
    traffic, _ = pmc_traffic_for(tree_file, name)
    out = {
        "tree": os.path.basename(tree_file),
        "nslices_log2": float(np.log2(tree.nslices)),
        "width_log2": float(np.log2(tree.max_size())),
        "macs_per_slice": int(plan.macs_per_slice),
        "slice_groups": slice_groups_note(plan, ids),
        "ms_per_slice": dt * 1e3,
        "tflops": flops / dt / 1e12,
        "mixed_bound_ms": bound_ms,
        "mixed_bound_frac": bound_ms / (dt * 1e3),
        "mixed_bound_definition": "sum over steps of max(flops_i / matrix peak, MOVED bytes_i / 8 TB/s)",
        "unfused_roofline_ms": unfused_ms,
        "est_time_total_s": dt * tree.nslices,
        "dominant_kernel": {
            "kernel": name,
            "share_of_slice_time": dom["ms"] / max(sum(r["ms"] for r in rows), 1e-9),
            "tflops": mf,
            "matrix_peak_tflops": dom_peak,
            "frac_of_matrix_peak": mf / dom_peak,
            "moved_gbs": bw,
            "frac_of_hbm_peak": bw / PEAK_HBM_GBS,
            "traffic": traffic,
        },
        "precision": precision_check(tree, arrays),
    }
    out["arithmetic"] = ARITH_NOTE[arith]
    if bf16x3:
        # fp32-equivalent flops; the fused pairs run on the bf16 pipe and are priced against bf16 peak /
        # 6 products (per step, in mixed_bound_ms); the ratio to the fp32 pipe's peak is a comparison
        # with the fp32 arithmetic, not a fraction of a bound
        out["tflops_are"] = ("fp32-equivalent (8 real flops per complex MAC; %s per fp32 product)"
                             % ("three fp16 products" if arith == "fp16x2" else "six bf16 products"))
        out["bf16x3_peak_tflops"] = PEAK_FP16X2_TFLOPS if arith == "fp16x2" else PEAK_BF16X3_TFLOPS
        out["ratio_to_fp32_mfma_peak"] = flops / dt / 1e12 / PEAK_MFMA_F32_TFLOPS
    else:
        out["frac_of_mfma_peak"] = flops / dt / 1e12 / PEAK_MFMA_F32_TFLOPS
    fn.close()
    return out


def small_config(name, tree, arrays, dev, slices, reps, cpu_slices, note, dtype="complex64"):
    """A launch-/latency-bound configuration: time per contraction (all
    ``slices`` slices), mixed roofline, the oracle on the host cores."""
    import torch

    from cotengra_amd.contractor import HipContractor
    from oracle import contract_ref as orc

    arrays = [np.asarray(a).astype(dtype) for a in arrays]
    fn = HipContractor(tree, handle_slicing=True)
    st = fn.setup(*[torch.as_tensor(a, device=dev) for a in arrays])
    ex, plan = st["exec"], st["plan"]
    ex.zero_result()
    # (slice groups: whole groups -- over the whole job every group is complete; a slice count that is not a
    # multiple of the group size is rounded down to one and says so in "slices_timed")
    ids = None
    if plan.group_size > 1 and slices < tree.nslices:
        slices = int(plan.group_size) * max(1, slices // int(plan.group_size))
        ids = slice_ids_from_groups(plan, 0, slices)

    def run_all():
        if ids is None:
            ex.run_slices(0, slices, 1)
        else:
            ex.run_slice_list(ids)

    for _ in range(2):   # warm-up touches every arena replica of the slice batch
        run_all()
    ex.sync()
    # repetitions in groups, the median group counts: the first passes after an
    # allocation of this size are occasionally several times slower (first touch)
    groups, per = (5, max(reps // 5, 1))
    times = []
    for _ in range(groups):
        t0 = time.perf_counter()
        for _ in range(per):
            run_all()
        ex.sync()
        times.append((time.perf_counter() - t0) / per)
    dt = sorted(times)[len(times) // 2]
    rows = step_table(ex, plan)
    batch = ex.batch
    launches = ex.launch_count()[1]   # independent small steps share launches
    roof_ms = mixed_roofline_ms(rows, 8.0) * slices   # (no fused pairs here: moved = algorithmic bytes)
    # (round 5: long tiled steps multiply on the bf16 pipe -- pair_mfma_bf3_kernel -- and are priced against bf16
    # peak / 6 above; the same sum with every step priced on the fp32 pipe is what rounds 1-4 reported)
    roof32_ms = mixed_roofline_ms(rows, 8.0, moved=False) * slices
    flops = (executed_flops(plan, ids) if ids is not None else
             executed_flops(plan, list(range(slices))) if plan.group_size > 1 else plan.flops_per_slice() * slices)
    # the oracle (numpy, the reference's executor restated) on this node's cores
    ops = orc.extract_contractions(tree)
    t0 = time.perf_counter()
    for i in range(cpu_slices):
        orc.run_contractions(ops, orc.slice_arrays(tree, arrays, i) if tree.sliced_inds else arrays)
    cpu = (time.perf_counter() - t0) / cpu_slices * slices
    fn.close()
    return {
        "workload": note,
        "steps_per_slice": len(plan.steps),
        "launches_per_slice": launches,
        "slices_timed": slices,
        "ms": dt * 1e3,
        "ms_slowest_group": max(times) * 1e3,
        "slices_per_sec": slices / dt,
        "tflops": flops / dt / 1e12,
        "mixed_roofline_ms": roof_ms,
        "mixed_roofline_frac": roof_ms / (dt * 1e3),
        "mixed_roofline_fp32_pipe_ms": roof32_ms,
        "mixed_roofline_fp32_pipe_frac": roof32_ms / (dt * 1e3),
        "bf16x3_steps": sum(1 for r in rows if (r.get("kernel_name") or "").startswith(("pair_mfma_bf3_kernel", "pair_mfma_h2_kernel"))),
        "cpu_oracle_ms": cpu * 1e3,
        "cpu_cores": host_cores(),
        "cpu_sample": f"{cpu_slices} slice(s) with numpy {dtype}, scaled to {slices}",
        "slices_per_launch": int(min(slices, batch)),
        "speedup_vs_cpu_oracle": cpu / dt,
        "slice_groups": slice_groups_note(plan, ids if ids is not None else list(range(slices))),
    }


def other_configs(dev):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util as G

    import cotengra_amd as ca

    out = {}
    by_name = {c["name"]: c for c in G.cases("tree")}
    c2 = by_name["C2_lattice8x8_d4"]
    t2 = G.tree_of(c2)
    out["C2"] = small_config(
        "C2", t2, G.arrays_of(c2, "complex128", t2), dev, slices=1, reps=200, cpu_slices=5,
        note="8x8 PEPS-style lattice, bond dim 4, unsliced, one contraction",
    )
    m10 = os.path.join(TREES, "sycamore_m10.json")
    arr = os.path.join(ROOT, "tests", "golden", "sycamore_m10_arrays.npz")
    if os.path.exists(m10) and os.path.exists(arr):
        t3 = ca.tree_from_record(ca.load_network(m10))
        z = np.load(arr)
        out["C3"] = small_config(
            "C3", t3, [z[f"t{i}"] for i in range(t3.N)], dev, slices=t3.nslices, reps=25, cpu_slices=2,
            note=f"Sycamore circuit_n53_m10 amplitude, all {t3.nslices} slices (the whole amplitude)",
        )
    # C3 with 8 open output qubits: one contraction = the batch of 256 amplitudes (the open qubits
    # give the pairwise steps a real N dimension -- SURVEY section 8f item 3)
    m10o = os.path.join(TREES, "sycamore_m10_open8.json")
    arro = os.path.join(ROOT, "tests", "golden", "sycamore_m10_open8_arrays.npz")
    if os.path.exists(m10o) and os.path.exists(arro):
        t3o = ca.tree_from_record(ca.load_network(m10o))
        zo = np.load(arro)
        cfg = small_config(
            "C3_batched", t3o, [zo[f"t{i}"] for i in range(t3o.N)], dev, slices=t3o.nslices, reps=25,
            cpu_slices=1,
            note=f"Sycamore circuit_n53_m10, 8 open output qubits: 256 amplitudes per contraction, all "
                 f"{t3o.nslices} slices",
        )
        cfg["amplitudes_per_contraction"] = 256
        cfg["amplitudes_per_sec"] = 256.0 / (cfg["ms"] * 1e-3)
        out["C3_batched"] = cfg
    c5 = by_name["C5_hyper200"]
    t5 = G.tree_of(c5)
    out["C5"] = small_config(
        "C5", t5, G.arrays_of(c5, "complex128", t5), dev, slices=64, reps=10, cpu_slices=2,
        note="random-regular hyper network, 200 tensors, batch + hyper indices, 64 of its slices",
    )
    return out


def m10_amplitudes(dev, seconds=2.0):
    """Sycamore m10 amplitudes per second on THIS rank: every call is a full amplitude
    (upload of the 170 input tensors -- what changes with the bitstring --, all 64
    slices, fetch).  Amplitudes of different bitstrings are independent: the unit of
    configuration C3 that shards over GPUs without any exchange."""
    import torch

    import cotengra_amd as ca
    from cotengra_amd.contractor import HipContractor

    m10 = os.path.join(TREES, "sycamore_m10.json")
    arr = os.path.join(ROOT, "tests", "golden", "sycamore_m10_arrays.npz")
    if not (os.path.exists(m10) and os.path.exists(arr)):
        return None
    tree = ca.tree_from_record(ca.load_network(m10))
    z = np.load(arr)
    xs = [torch.as_tensor(z[f"t{i}"].astype("complex64"), device=dev) for i in range(tree.N)]
    fn = HipContractor(tree, handle_slicing=True)
    for _ in range(3):
        amp = fn(*xs)
    torch.cuda.synchronize(dev)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            amp = fn(*xs)
        torch.cuda.synchronize(dev)
        n += 10
    dt = time.perf_counter() - t0
    fn.close()
    return {"amplitudes": n, "seconds": dt, "nslices": int(tree.nslices), "amplitude": complex(amp.item())}


def m10_strong_setup(dev):
    """Everything of ``m10_strong`` that can fail on one rank alone (files, executor, device memory) and
    involves no collective: ``(fn, st, tree)`` or None.  The caller lets the ranks agree before the first
    collective (all ranks or none)."""
    import torch

    import cotengra_amd as ca
    from cotengra_amd.contractor import HipContractor

    m10 = os.path.join(TREES, "sycamore_m10.json")
    arr = os.path.join(ROOT, "tests", "golden", "sycamore_m10_arrays.npz")
    if not (os.path.exists(m10) and os.path.exists(arr)):
        return None
    tree = ca.tree_from_record(ca.load_network(m10))
    z = np.load(arr)
    xs = [torch.as_tensor(z[f"t{i}"].astype("complex64"), device=dev) for i in range(tree.N)]
    fn = HipContractor(tree, handle_slicing=True)
    st = fn.setup(*xs)
    return fn, st, tree


def m10_strong(dev, rank, world, comm, dist, setup, reps=20):
    """BASELINE config 3 as worded: ONE Sycamore m10 amplitude, its 64 slices dealt over the N ranks
    (``ctg_exec_run_share``) + the RCCL reduce to rank 0, against the same amplitude on one rank.  Strong
    scaling of a 3 ms job: printed because the configuration names it, not because it scales.
    ``setup`` = ``m10_strong_setup(dev)``, which every rank has completed (the caller checked)."""
    import torch

    fn, st, tree = setup
    ex, result = st["exec"], st["result"]

    def sync_all():
        ex.sync()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(r, w, reduce_):
        for _ in range(3):
            ex.zero_result()
            ex.run_share(r, w)
            if reduce_:
                reduce_()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(reps):
            ex.zero_result()
            ex.run_share(r, w)
            if reduce_:
                reduce_()
        sync_all()
        return (time.perf_counter() - t0) / reps

    def reduce_():
        if comm is not None:
            ex.reduce(comm, 0)
        elif dist is not None:
            dist.reduce(torch.view_as_real(result), dst=0)

    one = timed(0, 1, None)                       # every rank computes the whole amplitude alone
    amp1 = complex(result.item())
    many = timed(rank, world, reduce_ if world > 1 else None)
    ampn = complex(result.item())
    if dist is not None:
        t = torch.tensor([one, many], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        one, many = float(t[0].item()), float(t[1].item())
    fn.close()
    return {
        "workload": "Sycamore circuit_n53_m10 amplitude, %d slices dealt over %d rank(s) + 1 RCCL reduce to rank 0"
                    % (tree.nslices, world),
        "ms": many * 1e3, "ms_one_rank": one * 1e3, "speedup_vs_one_rank": one / many, "nslices": int(tree.nslices),
        "amplitude_rel_diff": abs(ampn - amp1) / abs(amp1) if rank == 0 else None,
    }


# ---------------------------------------------------------------------- #
# what goes to stdout: ONE compact line (the driver parses the last stdout line and keeps an 8 KB
# tail); the full record -- every leg, every kernel -- goes to bench_full.json
# ---------------------------------------------------------------------- #

COMPACT_LIMIT = 4096


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _sig(x, digits=6):
    """Floats to ``digits`` significant digits (the compact line only)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_record(out, full_path="bench_full.json"):
    """The line the driver reads: the contract's keys, ``roofline`` and ``cpu_baseline`` of the
    headline, and ONE number per extra leg.  Everything else lives in ``full_path``."""
    roof = out.get("roofline") or {}
    cfg = out.get("config") or {}
    rec = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline")
    rec["dtype"] = out.get("dtype_short", "complex64")
    rec["data"] = out.get("data", "synthetic")
    rec.update(_pick(out, "slices_per_sec", "tflops", "est_time_total_s"))
    rec["config"] = _pick(cfg, "workload", "network", "tree", "width_log2", "arena_gib", "nslices_log2", "macs_per_slice", "flops_per_slice",
                          "bytes_moved_per_slice", "algorithmic_bytes_per_slice", "steps_per_slice", "parallelism",
                          "reduce_via", "arithmetic")
    sg = cfg.get("slice_groups")
    if sg:
        rec["config"]["slice_groups"] = _pick(sg, "slices_per_group", "shared_steps", "timed")
    rec["roofline"] = _pick(roof, "bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                            "launches_per_slice", "moved_bytes_per_launch", "algorithmic_bytes_per_launch",
                            "flops_per_launch", "traffic", "share_of_slice_time")
    if "matrix_side" in roof:
        rec["roofline"]["matrix_frac"] = roof["matrix_side"].get("frac")
    if "mixed_per_step" in roof:
        # (flat: the driver's parser keeps one level below "roofline")
        rec["roofline"]["mixed_frac"] = roof["mixed_per_step"].get("frac")
        rec["roofline"]["mixed_bound_ms"] = roof["mixed_per_step"].get("bound_ms")
    if out.get("precision"):
        rec["precision"] = _pick(out["precision"], "rel_err", "gate", "numpy_complex64_rel_err",
                                 "rel_err_complex128_path", "sum_rel_err", "sum_slices")
    if out.get("cpu_baseline"):
        rec["cpu_baseline"] = _pick(out["cpu_baseline"], "value", "unit", "cores", "kind", "sample")
    # one number per extra leg
    legs = {}
    for key, leg in (("tts_ms", "time_to_solution_tree"), ("w33_ms", "time_to_solution_tree_w33"),
                     ("tts_g_ms", "time_to_solution_tree_groups"), ("peak_tree_ms", "peak_rate_tree")):
        if isinstance(out.get(leg), dict):
            legs[key] = out[leg].get("ms_per_slice")
            legs[key.replace("_ms", "_total_s")] = out[leg].get("est_time_total_s")
    for name, leg in out.items():
        if name.endswith("_arithmetic") and isinstance(leg, dict):
            short = name[: -len("_arithmetic")]
            for tree_name, rep in leg.items():
                if tree_name == cfg.get("tree"):
                    legs[short + "_ms"] = rep.get("ms_per_slice")
                    legs[short + "_rel_err"] = (rep.get("precision") or {}).get("rel_err")
    for name, c in (out.get("configs") or {}).items():
        if "mixed_roofline_frac" in c:
            legs[name + "_ms"] = c.get("ms")
            legs[name + "_frac"] = c.get("mixed_roofline_frac")
            legs[name + "_launches"] = c.get("launches_per_slice")
            if c.get("bf16x3_steps"):   # (some steps on the bf16 pipe: the fraction with every step priced at 157.3 TF next to it)
                legs[name + "_frac_fp32_pipe"] = c.get("mixed_roofline_fp32_pipe_frac")
        elif "amplitudes_per_sec" in c:
            legs[name + "_per_sec"] = c.get("amplitudes_per_sec")
        elif "ms" in c:
            legs[name + "_ms"] = c.get("ms")
            if "speedup_vs_one_rank" in c:
                legs[name + "_speedup"] = c.get("speedup_vs_one_rank")
    if legs:
        rec["legs"] = legs
    if out.get("per_rank"):
        rec["distinct_gpus"] = out.get("distinct_gpus")
        rec["slowest_rank_ms"] = max(r["wall_ms"] for r in out["per_rank"])
        rec["reduce_wait_ms_max"] = max(r["reduce_wait_ms"] for r in out["per_rank"])
        # the spread over the ranks (an 8-GPU run: which rank lags, and whether in its slices or in the reduce)
        rec["ranks"] = {
            "slices_ms": [round(r["slices_ms"], 3) for r in out["per_rank"]],
            "reduce_wait_ms": [round(r["reduce_wait_ms"], 3) for r in out["per_rank"]],
            "share_units": [r.get("share_units") for r in out["per_rank"]],
            "slices_per_unit": out["per_rank"][0].get("slices_per_unit"),
        }
    rec["full_record"] = full_path
    rec = _sig(rec)
    # never over the limit: drop the least important keys first (none of them is part of the contract)
    for drop in ("legs", "precision", ("config", "slice_groups"), ("roofline", "mixed_bound_ms"),
                 ("config", "network")):
        if len(json.dumps(rec, separators=(",", ":"))) < COMPACT_LIMIT:
            break
        if isinstance(drop, tuple):
            rec.get(drop[0], {}).pop(drop[1], None)
        else:
            rec.pop(drop, None)
    return rec


def emit(out):
    """Write the full record next to the script (and under gpurun_out/ when that exists on this
    box), then print the compact record as the LAST stdout line."""
    name = "bench_full.json" if out.get("n_gpus", 1) == 1 else "bench_full_n%d.json" % out["n_gpus"]
    written = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, name), "w") as f:
                    json.dump(out, f)
                written = written or os.path.relpath(os.path.join(d, name), ROOT)
            except OSError:
                pass
    line = json.dumps(compact_record(out, written or name), separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT and "\n" not in line
    sys.stdout.flush()
    print(line, flush=True)


def respawn(args):
    """No launcher in sight: start N ranks of this script through
    torch.distributed.run on the loopback address and pass its exit code on."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
    ] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tree", default=TREE)
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip every CPU-oracle leg (baseline, precision) and the extra workloads")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip time_to_solution_tree and the C2/C3/C5 configs")
    ap.add_argument("--dump-steps", default=None, help="write per-step timings JSON here")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        respawn(args)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(
            f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible"
        )
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    devices = None
    if "RANK" in os.environ:
        # one process per GPU over RCCL
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"world size {dist.get_world_size()} != --gpus {args.gpus}")
        from cotengra_amd.distributed import assert_distinct_devices

        devices = assert_distinct_devices()  # raises unless N distinct GPUs are in use

    import cotengra_amd as ca
    from cotengra_amd import runtime
    from cotengra_amd.contractor import HipContractor

    rec = ca.load_network(args.tree)
    tree = ca.tree_from_record(rec)
    arrays = ca.make_arrays_from_inputs(
        tree.inputs, tree.size_dict, seed=42, dtype="complex64", rescale=True
    )
    fn = HipContractor(tree, handle_slicing=True)
    st = fn.setup(*[torch.as_tensor(a, device=dev) for a in arrays])
    ex, plan = st["exec"], st["plan"]
    ex.set_strip_exponent(False)
    result = st["result"]
    nsl = tree.nslices

    # the single collective: RCCL behind the C ABI on the exec's stream; the
    # torch process group only hands the unique id around
    comm, reduce_via = None, None
    if dist is not None:
        try:
            comm = runtime.Comm.from_torch_group(None, device=local_rank)
            reduce_via = "ctg_exec_reduce (RCCL, C ABI)"
        except runtime.CtgError as e:  # e.g. an RCCL the loader cannot find
            if rank == 0:
                print(f"ctg_comm_init failed ({e}); reducing through torch.distributed", file=sys.stderr)
            reduce_via = "torch.distributed.reduce (RCCL)"
        flag = torch.tensor([0 if comm is not None else 1], device=dev)
        dist.all_reduce(flag)
        if int(flag.item()) != 0 and comm is not None:  # all ranks or none
            comm.close()
            comm, reduce_via = None, "torch.distributed.reduce (RCCL)"

    def reduce_partials():
        if comm is not None:
            ex.reduce(comm, 0)
        elif dist is not None:
            dist.reduce(torch.view_as_real(result), dst=0)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warmup (untimed)
    ex.zero_result()
    n_groups = max(1, plan.nslices // plan.group_size)
    if args.warmup:   # (on the last slice groups; the timed slices are whole groups from the first on)
        ex.run_slice_list(slice_ids_from_groups(plan, n_groups - world * (1 + (args.warmup - 1) // plan.group_size),
                                                args.warmup, rank, world))
    timed_ids = slice_ids_from_groups(plan, 0, args.steps, rank, world)
    reduce_partials()
    barrier()

    # timed: K slices per rank + the single RCCL reduce of the partial amplitude
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ex.zero_result()
    barrier()
    t0 = time.perf_counter()
    ev[0].record()
    ex.run_slice_list(timed_ids)
    ev[1].record()
    reduce_partials()
    ev[2].record()
    barrier()
    dt_local = dt = time.perf_counter() - t0
    slices_ms, reduce_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([dt_local * 1e3, slices_ms, reduce_ms], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [
            {"rank": r, "device": devices[r] if devices else None, "wall_ms": float(v[0]),
             "slices_ms": float(v[1]), "reduce_wait_ms": float(v[2]),
             # (what the library deals this rank over the WHOLE job: ctg_plan_share_units)
             "share_units": int(plan.share_units(r, world)[0]), "slices_per_unit": int(plan.share_units(r, world)[1])}
            for r, v in enumerate(allr)
        ]

    extras = rank == 0 and world == 1 and not args.no_cpu_baseline
    precision = precision_check(tree, arrays) if (rank == 0 and not args.no_cpu_baseline) else None
    if precision is not None:
        # the quantity north_star gates: a sum of slices, without the numpy-relative clause
        psum = precision_sum_check(tree, arrays)
        precision["sum"] = psum
        precision["sum_rel_err"], precision["sum_slices"] = psum["rel_err"], psum["slices"]
    # C3 at N > 1: every rank computes m10 amplitudes (of different bitstrings) on its own
    c3_amp = None
    # (CTG_BENCH_C3_AMPLITUDES=1: also at N = 1 under a launcher -- how the one-GPU lease tests this leg)
    if (world > 1 or os.environ.get("CTG_BENCH_C3_AMPLITUDES")) and not args.headline_only:
        try:
            mine3 = m10_amplitudes(dev)
        except Exception as e:  # noqa: BLE001  (an extra leg must not cost the headline line)
            print(f"rank {rank}: C3_amplitudes leg failed: {e!r}", file=sys.stderr)
            mine3 = None
        # (all ranks or none: the gather below is a collective)
        if dist is not None:
            ok3 = torch.tensor([0 if mine3 is not None else 1], device=dev)
            dist.all_reduce(ok3)
            if int(ok3.item()) != 0:
                mine3 = None
        if mine3 is not None:
            t3 = torch.tensor([mine3["amplitudes"], mine3["seconds"]], dtype=torch.float64, device=dev)
            all3 = [torch.zeros_like(t3) for _ in range(world)]
            if dist is not None:
                dist.all_gather(all3, t3)
            else:
                all3 = [t3]
            c3_amp = {
                "workload": "Sycamore circuit_n53_m10 amplitudes (64 slices each; input upload, all slices, "
                            "fetch per amplitude), every rank its own bitstrings, no exchange",
                "amplitudes_per_sec": float(sum(float(v[0]) / float(v[1]) for v in all3)),
                "amplitudes_per_sec_per_rank": [float(v[0]) / float(v[1]) for v in all3],
                "slices_per_sec": float(sum(float(v[0]) / float(v[1]) for v in all3)) * mine3["nslices"],
                "note": "one amplitude takes ~3 ms on one GPU and does not strong-scale: the unit that "
                        "shards is the amplitude, not the slice",
            }

    # BASELINE config 3 as worded: the one 64-slice amplitude over the N ranks (strong scaling)
    c3_strong = None
    if (world > 1 or os.environ.get("CTG_BENCH_C3_AMPLITUDES")) and not args.headline_only:
        try:
            setup3 = m10_strong_setup(dev)
        except Exception as e:  # noqa: BLE001  (an extra leg must not cost the headline line)
            print(f"rank {rank}: C3_strong setup failed: {e!r}", file=sys.stderr)
            setup3 = None
        # (all ranks or none: the leg barriers and reduces -- a rank that failed above must not leave the
        # others waiting in a collective before the headline line is printed)
        if dist is not None:
            bad3 = torch.tensor([0 if setup3 is not None else 1], device=dev)
            dist.all_reduce(bad3)
            if int(bad3.item()) != 0:
                if setup3 is not None:
                    setup3[0].close()
                setup3 = None
        if setup3 is not None:
            try:
                c3_strong = m10_strong(dev, rank, world, comm, dist, setup3)
            except Exception as e:  # noqa: BLE001
                print(f"rank {rank}: C3_strong leg failed: {e!r}", file=sys.stderr)
                c3_strong = None

    # (flops really executed: the steps a slice group shares count once per group -- every rank times the
    # same number of whole groups, so rank 0's count x world is the job's)
    flops_slice = executed_flops(plan, timed_ids) / args.steps
    total_slices = args.steps * world
    value = flops_slice * total_slices / dt

    if rank == 0:
        rows = step_table(ex, plan)
        dom_name, dom, by_name = dominant_kernel(rows, 8.0)
        arith_run = stem_arithmetic()
        bf3_run = arith_run if arith_run != "fp32" else False
        dom_peak = step_peak_tflops(dom, bf3_run)
        achieved_tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        achieved_gbs = dom["moved"] / (dom["ms"] * 1e-3) / 1e9
        # which roof binds the dominant kernel: the longer of flops / matrix peak and moved bytes / HBM peak
        hbm_bound = dom["moved"] / (PEAK_HBM_GBS * 1e9) > dom["flops"] / (dom_peak * 1e12)
        mf = [r for r in rows if r["kernel"] == "mfma"]
        all_flops = sum(8.0 * r["macs"] for r in mf)
        all_ms = sum(r["ms"] for r in mf)
        slice_ms = sum(r["ms"] for r in rows)
        bound_ms = mixed_roofline_ms(rows, 8.0, moved=True, bf16x3=bf3_run)
        unfused_ms = mixed_roofline_ms(rows, 8.0, moved=False)
        traffic, traffic_all = pmc_traffic_for(args.tree, dom_name)
        step_ms = dt * 1e3 / args.steps
        roofline = {
            "bound": "hbm" if hbm_bound else "mfma",
            "kernel": dom_name,
            "achieved": achieved_gbs if hbm_bound else achieved_tf,
            "peak": PEAK_HBM_GBS if hbm_bound else dom_peak,
            "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": achieved_gbs / PEAK_HBM_GBS if hbm_bound else achieved_tf / dom_peak,
            "bound_is": "the longer of flops / matrix peak and MOVED bytes / 8 TB/s for this kernel's launches; "
                        "both sides follow",
            # the matrix side: an fp16 x 2 pair is priced against fp16 peak / 3 products, a bf16 x 3 pair against
            # bf16 peak / 6 (fp32-equivalent flops), an fp32 kernel against the fp32 matrix peak
            "matrix_side": {"achieved_tflops": achieved_tf, "peak_tflops": dom_peak, "frac": achieved_tf / dom_peak,
                            "pipe": ("fp16 MFMA, 3 products per fp32 product" if dom_peak == PEAK_FP16X2_TFLOPS
                                     else "bf16 MFMA, 6 products per fp32 product" if dom_peak == PEAK_BF16X3_TFLOPS
                                     else "fp32 MFMA")},
            "hbm_side": {"achieved_gbs": achieved_gbs, "peak_gbs": PEAK_HBM_GBS, "frac": achieved_gbs / PEAK_HBM_GBS},
            "launches_per_slice": dom["n"],
            "avg_launch_ms": dom["ms"] / dom["n"],
            "flops_per_launch": dom["flops"] / dom["n"],
            # SURVEY 8(d) bytes of the launch's reference steps (a fused pair: both steps, incl. the
            # intermediate it never writes) and the bytes the launch really moves; the HBM side of
            # the kernel is priced on the latter, cross-checked by `traffic` (PMC)
            "algorithmic_bytes_per_launch": dom["bytes"] / dom["n"],
            "moved_bytes_per_launch": dom["moved"] / dom["n"],
            "share_of_slice_time": dom["ms"] / max(slice_ms, 1e-9),
            "traffic": traffic,
            "traffic_source": "profiles/pmc_summary_%s.json (separate --pmc passes of this tree, committed; "
                              "not measured in this run)" % os.path.splitext(os.path.basename(args.tree))[0],
            "traffic_all_mfma_per_launch": traffic_all,
            "all_mfma_kernels": {
                "achieved_tflops": all_flops / (all_ms * 1e-3) / 1e12,
                "launches_per_slice": len(mf),
                "share_of_slice_time": all_ms / max(slice_ms, 1e-9),
            },
            "mixed_per_step": {
                "definition": "THE BOUND: sum over steps of max(flops_i / matrix peak_i, MOVED bytes_i / 8 TB/s) -- a "
                              "fused pair is priced on the bytes it moves (big operand in, result out) and against "
                              "its own pipe (fp16 x 2: fp16 peak / 3; bf16 x 3: bf16 peak / 6); every other step "
                              "against 157.3 TF",
                "bound_ms": bound_ms,
                "frac": bound_ms / step_ms,
                "unfused_roofline_ms": unfused_ms,
                "unfused_roofline_is": "the same sum with the SURVEY 8(d) bytes of the unfused reference steps: the "
                                       "reference's execution model, which fused pairs may finish below -- reported, "
                                       "not a bound of this executor",
            },
            "by_kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(by_name.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if args.dump_steps:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump_steps)), exist_ok=True)
            json.dump(rows, open(args.dump_steps, "w"))
        out = {
            "metric": "contracted FLOP/s (whole node), Sycamore n53 m20 sliced amplitude",
            "value": value,
            "unit": "FLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("complex64, 8 real flops per complex MAC -- " + ARITH_NOTE[arith_run]),
            "dtype_short": "complex64",
            "data": "synthetic",
            "slices_per_sec": total_slices / dt,
            "tflops": value / 1e12,
            # (fp32-equivalent flops over the fp32 matrix peak: a FRACTION of a bound only in the fp32
            # arithmetic -- with bf16 x 3 pairs it is the ratio to the pipe the reference arithmetic maps to)
            ("frac_of_mfma_peak_whole_job" if not bf3_run else "ratio_to_fp32_mfma_peak_whole_job"):
                value / 1e12 / (PEAK_MFMA_F32_TFLOPS * world),
            "cotengra_convention_gigaflops": 4.0 * plan.macs_per_slice * total_slices / dt / 1e9,
            "est_time_total_s": nsl / (total_slices / dt),
            "config": {
                # (<= 120 characters: the driver's parser cuts fields at 128; file, width and arena have keys of their own)
                "workload": "Sycamore n53 m20 amplitude, sliced tree, one slice per step per GPU",
                "network": "examples/benchmarks/sycamore_n53_m20_s0_e0_pABCDCDAB.json",
                "tree": os.path.basename(args.tree),
                "width_log2": int(round(np.log2(tree.max_size()))),
                "arena_gib": round(plan.arena_elems * plan.itemsize / 2**30, 1),
                "nslices_log2": float(np.log2(nsl)),
                "macs_per_slice": int(plan.macs_per_slice),
                "flops_per_slice": float(flops_slice),
                "flops_per_slice_is": "flops executed per slice in the timed region: the steps a slice group shares "
                                      "count once per group (nominal, every step for every slice: %.6g)" % plan.flops_per_slice(),
                "slice_groups": slice_groups_note(plan, timed_ids),
                "algorithmic_bytes_per_slice": float(plan.bytes_per_slice()),
                "bytes_moved_per_slice": float(plan.elems_moved_per_slice * plan.itemsize),
                "fused_stem_pairs": sum(1 for s_ in plan.steps if s_.kind == 3),
                "steps_per_slice": len(plan.steps),
                "parallelism": f"slice-parallel x{world}, 1 RCCL reduce",
                "arithmetic": {"fp32": "fp32 MFMA", "bf16x3": "bf16x3 stem pairs + fp32 MFMA",
                               "fp16x2": "fp16x2 stem pairs and long tiled steps + fp32 MFMA"}[arith_run],
                "reduce_via": reduce_via,
                "partial_amplitude": [float(result.real.item()), float(result.imag.item())]
                if result.numel() == 1
                else None,
            },
            "roofline": roofline,
            "precision": precision,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["distinct_gpus"] = len(set(devices))
        fn.close()
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tree, arrays)
        if extras and not args.headline_only:
            if os.path.abspath(args.tree) != os.path.abspath(TTS_TREE) and os.path.exists(TTS_TREE):
                out["time_to_solution_tree"] = tree_report(TTS_TREE, dev)
            if os.path.abspath(args.tree) != os.path.abspath(TTS33_TREE) and os.path.exists(TTS33_TREE):
                out["time_to_solution_tree_w33"] = tree_report(TTS33_TREE, dev, steps=3)
            if os.path.exists(TTS_GROUPS_TREE):
                out["time_to_solution_tree_groups"] = tree_report(TTS_GROUPS_TREE, dev)
            if os.path.abspath(args.tree) != os.path.abspath(PEAK_TREE) and os.path.exists(PEAK_TREE):
                out["peak_rate_tree"] = tree_report(PEAK_TREE, dev)
            # the other arithmetic of the fused pairs on the same trees (each entry with its own
            # precision check and rooflines): fp32 products on the fp32 matrix cores when the line
            # runs bf16 x 3 (the default), and the other way round
            # the other arithmetics of the fused pairs on the same trees (each entry with its own precision check and
            # rooflines): fp32 products on the fp32 matrix cores and the exact bf16 x 3 split when the line runs
            # fp16 x 2 (the default), ...
            for other in [a for a in ("fp32", "bf16x3", "fp16x2") if a != arith_run]:
                trees_o = (args.tree, TTS_TREE, TTS33_TREE) if other == "fp32" or arith_run == "fp32" else (args.tree,)
                out[other + "_arithmetic"] = {
                    os.path.basename(t): tree_report(t, dev, steps=3, mode=other) for t in trees_o if os.path.exists(t)
                }
            out["configs"] = other_configs(dev)
        if c3_amp is not None:
            out.setdefault("configs", {})["C3_amplitudes"] = c3_amp
        if c3_strong is not None:
            out.setdefault("configs", {})["C3_strong"] = c3_strong
        emit(out)
    else:
        fn.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
