#!/usr/bin/env python
"""Copy the summaries of the last tools/final_measure.sh run (gpurun_out/final)
into profiles/ (tracked).  Usage: python tools/publish_profiles.py [round-tag]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"

pairs = {
    "kernels_kernels.json": f"{tag}_bench_kernels.json",
    "kernels_kernels.txt": f"{tag}_bench_kernels.txt",
    "bench_line.json": f"{tag}_bench_line.json",
    "bench_compact.json": f"{tag}_bench_compact.json",
    "kernels_headline_kernels.json": f"{tag}_bench_kernels_headline.json",
    "kernels_headline_kernels.txt": f"{tag}_bench_kernels_headline.txt",
}
for tree in ("sycamore_m20_w32_c512", "sycamore_m20_native", "sycamore_m20_w32_r4"):
    pairs[f"pmc_summary_{tree}.json"] = f"pmc_summary_{tree}.json"   # (read by bench.py: roofline.traffic)
    pairs[f"steps_{tree}.txt"] = f"{tag}_steps_{tree}.txt"
    pairs[f"steps_{tree}_fp32.txt"] = f"{tag}_steps_{tree}_fp32.txt"
for w in ("C2", "C3", "C5"):
    pairs[f"timeline_{w}.txt"] = f"{tag}_timeline_{w}.txt"
    pairs[f"kernels_{w}.txt"] = f"{tag}_kernels_{w}.txt"
    pairs[f"steps_batched_{w}.txt"] = f"{tag}_steps_batched_{w}.txt"
for a, b in pairs.items():
    if os.path.exists(os.path.join(SRC, a)):
        shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))
    else:
        print("missing", a)
line = json.loads(open(os.path.join(SRC, "bench_line.json")).read())
r = line["roofline"]
if r.get("traffic") is None:
    # bench.py ran before this run's counter passes were summarised (it reads the COMMITTED
    # summary, whose kernel names may predate a kernel change): resolve the dominant kernel's
    # HBM bytes per launch from the passes of the same final_measure.sh run
    pm = os.path.join(SRC, "pmc_summary_%s.json" % os.path.splitext(os.path.basename(line["config"]["tree"]))[0])
    kv = json.load(open(pm)).get("kernels", {}).get(r["kernel"]) if os.path.exists(pm) else None
    if kv:
        r["traffic"] = kv["hbm_bytes_per_launch"]
        r["traffic_source"] = ("profiles/%s (separate --pmc passes of this tree in the same tools/final_measure.sh "
                               "run as this line; filled in by tools/publish_profiles.py)" % os.path.basename(pm))
        json.dump(line, open(os.path.join(DST, f"{tag}_bench_line.json"), "w"))
print(f"value {line['value']:.4e} {line['unit']}  {line['ms_per_step']:.1f} ms/step  est total {line['est_time_total_s']:.3e} s")
print(f"dominant {r['kernel']}: {r['achieved']:.1f} {r['unit']} frac {r['frac']:.3f} share {r['share_of_slice_time']:.2f}")
print("mixed:", r["mixed_per_step"])
for key in ("time_to_solution_tree", "time_to_solution_tree_w33", "peak_rate_tree"):
    t = line.get(key)
    if t:
        print(key + ":", {k: t.get(k) for k in ("tree", "ms_per_slice", "tflops", "frac_of_mfma_peak", "mixed_bound_frac", "est_time_total_s")})
for k, v in (line.get("configs") or {}).items():
    print(k, {x: v[x] for x in ("ms", "slices_per_sec", "tflops", "mixed_roofline_frac", "cpu_oracle_ms", "speedup_vs_cpu_oracle")})
print("precision:", line.get("precision"))
print("cpu:", line.get("cpu_baseline"))
