#!/bin/bash
# Arithmetic of the fused pairs, per pair and per slice, on ONE box: fp32 (row-interleaved step 2),
# bf16 x 3 (both steps on the bf16 matrix cores), mixed (step 1 bf16 x 3, step 2 fp32
# row-interleaved; CTG_STEM_MIXED=1).  Writes gpurun_out/mixed/steps_<tree>_<mode>.json + a table.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/mixed; mkdir -p $O
for t in sycamore_m20_native sycamore_m20_fused; do
  for mode in fp32 bf16x3 mixed; do
    case $mode in
      fp32) E="CTG_STEM_BF16X3=0";;
      bf16x3) E="CTG_STEM_BF16X3=1";;
      mixed) E="CTG_STEM_BF16X3=1 CTG_STEM_MIXED=1";;
    esac
    env $E timeout 300 python bench.py --headline-only --no-cpu-baseline --steps 6 --warmup 2 \
        --tree tests/golden/trees/$t.json --dump-steps $O/steps_${t}_$mode.json 2>/dev/null | tail -1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t $mode', round(d['ms_per_step'],2), 'ms/slice')"
  done
done
python - <<PY
import json
for t in ("sycamore_m20_native", "sycamore_m20_fused"):
    rows = {m: {r["step"]: r for r in json.load(open("$O/steps_%s_%s.json" % (t, m))) if r["ms"] > 0} for m in ("fp32", "bf16x3", "mixed")}
    print(t, "fused pairs: label | fp32 | bf16x3 | mixed (ms); plans differ between fp32 and bf16x3 pricing, mixed has bf16x3's plan")
    for st, r in sorted(rows["bf16x3"].items(), key=lambda kv: -kv[1]["ms"]):
        if r.get("kind") != "stem2":
            continue
        f = rows["fp32"].get(st)
        m = rows["mixed"].get(st)
        same = f is not None and f.get("label") == r.get("label")
        print("  %-46s %8s %8.2f %8.2f   %s" % (r["label"], ("%.2f" % f["ms"]) if same else "-", r["ms"], m["ms"] if m else -1, m["kernel_name"] if m else ""))
PY
