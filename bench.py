#!/usr/bin/env python
"""bench.py -- sliced Sycamore-53 m20 amplitude contraction on MI355X.

One "step" = one slice of the m20 contraction tree per GPU (SURVEY.md section 8d:
unit = one slice).  The tree fixture was found offline by the reference's own
hyper-optimizer + dynamic slicing and then refined with this package's native
subtree reconfiguration (tests/golden/gen/refine_native.py; ``--tree`` takes
any other fixture, e.g. sycamore_m20_w32_c128.json = least time to the full
amplitude); inputs are synthetic tensors of the named
shapes (reference ``make_arrays_from_inputs`` semantics, seed 42, complex64,
rescaled by size**0.25 so fp32 does not underflow) and are resident in HBM
before the timed region.  With N GPUs every rank contracts its own slices
(round-robin, no data-path traffic) and the partial amplitudes are combined
by ONE RCCL reduce inside the timed region -- weak scaling.

Prints ONE JSON line (rank 0): whole-node contracted FLOP/s, the dominant
kernel's roofline numbers measured live with HIP events, and the numpy-oracle
CPU baseline on this node's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0
TREE = os.path.join(ROOT, "tests", "golden", "trees", "sycamore_m20_w32_c512.json")


def shrink_for_cpu(tree, log2_width):
    """Slice further indices (largest tensor first) until one slice fits a
    CPU-sized budget; the schedule is unchanged, so MACs/s stays comparable."""
    tree = tree.copy()
    while tree.max_size() > 2**log2_width:
        big = max((p for p, _, _ in tree.traverse()), key=tree.get_size)
        ix = next(iter(tree.get_legs(big)))
        tree.remove_ind_(ix)
    return tree


def cpu_baseline(tree, arrays, budget_s=20.0, log2_width=24):
    """Time the numpy oracle (oracle/contract_ref.py, a port of the reference
    executor) on slices of the same tree narrowed to 2^log2_width."""
    from oracle import contract_ref as orc

    small = shrink_for_cpu(tree, log2_width)
    macs = small.contraction_cost() // small.nslices
    ops = orc.extract_contractions(small)
    t0 = time.time()
    n = 0
    while True:
        orc.run_contractions(ops, orc.slice_arrays(small, arrays, n))
        n += 1
        if time.time() - t0 > budget_s or n >= 64:
            break
    dt = time.time() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    return {
        "value": 8.0 * macs * n / dt,
        "unit": "FLOP/s",
        "cores": cores,
        "kind": "port",
        "sample": (
            f"{n} slices of the same m20 tree narrowed to width 2^{log2_width} "
            f"({macs:.3e} complex MACs/slice), numpy complex64, {dt:.1f}s"
        ),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tree", default=TREE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-steps", default=None, help="write per-step timings JSON here")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        # launched by torch.distributed.run: one process per GPU over RCCL
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import cotengra_amd as ca
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

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warmup (untimed)
    ex.zero_result()
    if args.warmup:
        ex.run_slices(rank % nsl, args.warmup, world)
    if dist is not None:
        buf = torch.view_as_real(result)
        dist.reduce(buf, dst=0)
    barrier()

    # timed: K slices per rank + the single RCCL reduce of the partial amplitude
    ex.zero_result()
    barrier()
    t0 = time.perf_counter()
    ex.run_slices((args.warmup * world + rank) % nsl, args.steps, world)
    if dist is not None:
        dist.reduce(torch.view_as_real(result), dst=0)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # precision: one slice of the same tree narrowed to a CPU-sized width.  The
    # complex128 HIP path must reproduce the numpy complex128 oracle (same
    # schedule, 1e-10); the complex64 production path is held to an fp32 bound
    # and reported next to the error numpy itself makes in complex64.
    precision = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import contract_ref as orc

        small = shrink_for_cpu(tree, 20)
        ref = complex(orc.contract_slice(small, [a.astype("complex128") for a in arrays], 3))
        np64 = complex(orc.contract_slice(small, arrays, 3))
        got = complex(np.asarray(small.contract_slice(arrays, 3)))
        got128 = complex(np.asarray(small.contract_slice([a.astype("complex128") for a in arrays], 3)))
        precision = {
            "check": "slice 3 of the tree narrowed to width 2^20, vs numpy complex128",
            "rel_err": abs(got - ref) / abs(ref),
            "gate": 1e-4,
            "rel_err_complex128_path": abs(got128 - ref) / abs(ref),
            "gate_complex128_path": 1e-10,
            "numpy_complex64_rel_err": abs(np64 - ref) / abs(ref),
        }
        if precision["rel_err"] > precision["gate"] or precision["rel_err_complex128_path"] > 1e-10:
            raise SystemExit(f"precision check failed: {precision}")

    flops_slice = plan.flops_per_slice()
    total_slices = args.steps * world
    value = flops_slice * total_slices / dt

    out = None
    if rank == 0:
        # per-kernel timing of one slice with HIP events on the exec's stream
        ms = ex.profile_slice(0)
        rows = plan.describe_steps()
        for r, m in zip(rows, ms):
            r["ms"] = float(m)
        names = ex.step_kernels()
        for r, nm in zip(rows, names):
            r["kernel_name"] = nm
        # dominant kernel = the kernel symbol with the largest share of slice time
        by_name = {}
        for r in rows:
            d = by_name.setdefault(r["kernel_name"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
            d["ms"] += r["ms"]
            d["flops"] += 8.0 * r["macs"]
            d["bytes"] += r["bytes"]
            d["n"] += 1
        dom_name, dom = max(by_name.items(), key=lambda kv: kv[1]["ms"])
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        mf = [r for r in rows if r["kernel"] == "mfma"]
        all_flops = sum(8.0 * r["macs"] for r in mf)
        all_ms = sum(r["ms"] for r in mf)
        roofline = {
            "bound": "mfma",
            "kernel": dom_name,
            "achieved": achieved,
            "peak": PEAK_MFMA_F32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_MFMA_F32_TFLOPS,
            "launches_per_slice": dom["n"],
            "avg_launch_ms": dom["ms"] / dom["n"],
            "flops_per_launch": dom["flops"] / dom["n"],
            "algorithmic_bytes_per_launch": dom["bytes"] / dom["n"],
            "share_of_slice_time": dom["ms"] / max(float(ms.sum()), 1e-9),
            "traffic": None,
            "all_mfma_kernels": {
                "achieved": all_flops / (all_ms * 1e-3) / 1e12,
                "frac": all_flops / (all_ms * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS,
                "launches_per_slice": len(mf),
                "share_of_slice_time": all_ms / max(float(ms.sum()), 1e-9),
            },
            "by_kernel_ms": {k: round(v["ms"], 3) for k, v in sorted(by_name.items(), key=lambda kv: -kv[1]["ms"])},
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(pmc):
            try:
                pm = json.load(open(pmc))
                kv = pm.get("kernels", {}).get(dom_name)
                if kv:
                    roofline["traffic"] = kv["hbm_bytes_per_launch"]
                roofline["traffic_all_mfma_per_launch"] = pm.get("hbm_bytes_per_launch")
            except Exception:
                pass
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
            "dtype": "complex64 (fp32 MFMA, 8 real flops per complex MAC)",
            "data": "synthetic",
            "slices_per_sec": total_slices / dt,
            "tflops": value / 1e12,
            "cotengra_convention_gigaflops": 4.0 * plan.macs_per_slice * total_slices / dt / 1e9,
            "est_time_total_s": nsl / (total_slices / dt),
            "config": {
                "workload": "Sycamore circuit_n53_m20 amplitude (examples/benchmarks/"
                "sycamore_n53_m20_s0_e0_pABCDCDAB.json), tree sliced to width 2^%d, "
                "one slice per step per GPU" % int(round(np.log2(tree.max_size()))),
                "tree": os.path.basename(args.tree),
                "nslices_log2": float(np.log2(nsl)),
                "macs_per_slice": int(plan.macs_per_slice),
                "flops_per_slice": float(flops_slice),
                "algorithmic_bytes_per_slice": float(plan.bytes_per_slice()),
                "steps_per_slice": len(plan.steps),
                "parallelism": f"slice-parallel x{world}, 1 RCCL reduce",
                "partial_amplitude": [float(result.real.item()), float(result.imag.item())]
                if result.numel() == 1
                else None,
            },
            "roofline": roofline,
            "precision": precision,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tree, arrays)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    fn.close()


if __name__ == "__main__":
    main()
