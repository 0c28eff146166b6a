#!/usr/bin/env python
"""Copy the summaries of the last tools/final_measure.sh run (gpurun_out/final)
into profiles/ (tracked) and put the PMC traffic of the dominant kernel into the
published bench line.  Usage: python tools/publish_profiles.py [round-tag]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"

pairs = {
    "pmc_summary.json": "pmc_summary.json",
    "kernels_kernels.json": f"{tag}_m20_bench_kernels.json",
    "kernels_kernels.txt": f"{tag}_m20_bench_kernels.txt",
    "pmc_fetch_k_kernels.json": f"{tag}_m20_pmc_fetch_kernels.json",
    "pmc_fetch_k_kernels.txt": f"{tag}_m20_pmc_fetch_kernels.txt",
    "pmc_write_k_kernels.json": f"{tag}_m20_pmc_write_kernels.json",
    "pmc_write_k_kernels.txt": f"{tag}_m20_pmc_write_kernels.txt",
    "steps.txt": f"{tag}_m20_steps.txt",
}
for a, b in pairs.items():
    shutil.copyfile(os.path.join(SRC, a), os.path.join(DST, b))
line = json.loads(open(os.path.join(SRC, "bench_line.json")).read())
pm = json.load(open(os.path.join(DST, "pmc_summary.json")))
dom = line["roofline"]["kernel"]
line["roofline"]["traffic"] = pm["kernels"][dom]["hbm_bytes_per_launch"]
line["roofline"]["traffic_all_mfma_per_launch"] = pm["hbm_bytes_per_launch"]
json.dump(line, open(os.path.join(DST, f"{tag}_bench_line.json"), "w"))
for extra in sorted(os.listdir(SRC)):
    if extra.startswith("bench_line_") and extra.endswith(".json"):
        shutil.copyfile(os.path.join(SRC, extra), os.path.join(DST, f"{tag}_{extra}"))
r = line["roofline"]
print(f"value {line['value']:.4e} {line['unit']}  {line['ms_per_step']:.1f} ms/step  est total {line['est_time_total_s']:.3e} s")
print(f"dominant {dom}: {r['achieved']:.1f} {r['unit']} frac {r['frac']:.3f} traffic {r['traffic']:.3e} B "
      f"(algorithmic {r['algorithmic_bytes_per_launch']:.3e}) share {r['share_of_slice_time']:.2f}")
print("all mfma:", r["all_mfma_kernels"])
print("precision:", line.get("precision"))
print("cpu:", line.get("cpu_baseline"))
