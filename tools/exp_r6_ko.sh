#!/bin/bash
# Round 6: where does the fp16 x 2 stem kernel's time go on specialised waves?  Knock-out builds of ctg_stem.hip (both
# objects) -- no MFMA = memory path alone; no gather / store = compute alone; no barriers; no scatter -- per-step times
# of one slice group of the headline tree, in the specialised-wave form (3) and the symmetric one (1).
#   build: CTG_VARIANT_SOURCES=ctg_stem.hip python tools/build_variants.py ko_mfma=-DCTG_STEM_KO_MFMA \
#          ko_mem=-DCTG_STEM_KO_GATHER,-DCTG_STEM_KO_STORE ko_bar=-DCTG_STEM_KO_BARRIER ko_scat=-DCTG_STEM_KO_SCATTER
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r6_ko; mkdir -p $O
X=$R/cotengra_amd/lib/exp
VARIANTS="default ko_mfma ko_mem ko_bar ko_scat"
for v in $VARIANTS; do
  for form in 3 1; do
    if [ $v = default ]; then unset CTG_LIB; else export CTG_LIB=$X/libctg_$v.so; fi
    [ $v != default ] && [ ! -f "$CTG_LIB" ] && continue
    CTG_STEM_FORM=$form timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only \
        --dump-steps $O/steps_${v}_f$form.json > $O/bench_${v}_f$form.out 2> $O/bench_${v}_f$form.err
    echo "$v form $form: $(python -c "import json; d=json.loads(open('$O/bench_${v}_f$form.out').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms/slice; dominant', d['roofline']['kernel'][-22:], round(d['roofline']['avg_launch_ms'],2), 'ms')" 2>&1 | tail -1)"
  done
done
unset CTG_LIB
VARIANTS="$VARIANTS" python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r6_ko"
V = os.environ["VARIANTS"].split()
def load(v, f):
    try:
        return {r["step"]: r for r in json.load(open(f"{O}/steps_{v}_f{f}.json")) if r.get("kind", "").startswith("stem")}
    except Exception:
        return {}
base = load("default", 3)
print("step  K1 N1 | K2 N2              " + " | ".join(f"{v}(f3) {v}(f1)" for v in V))
for st in sorted(base, key=lambda s: -base[s]["ms"])[:18]:
    row = [base[st]["label"][:30].ljust(30)]
    for v in V:
        for f in (3, 1):
            d = load(v, f)
            row.append("%8.2f" % d[st]["ms"] if st in d else "     n/a")
    print(st, " ".join(row))
PY
