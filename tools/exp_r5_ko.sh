#!/bin/bash
# Round 5: where does the bf16 x 3 stem kernel's time go?  (1) phase timeline of the dominant pair (32 32 | 64 64) in
# the XM and XM + LM forms; (2) knock-out builds (no MFMA = memory path alone; no gather / store = compute alone; no
# barriers), per-step times of one slice group of the headline tree, in both forms.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5_ko; mkdir -p $O
X=$R/cotengra_amd/lib/exp
for form in 1 2; do
  CTG_LIB=$X/libctg_tl.so CTG_TL_SHAPE=${CTG_TL_SHAPE:-32,32,64,64} CTG_STEM_FORM=$form timeout 200 python $R/tools/exp_stem_timeline.py > $O/timeline_form$form.txt 2>&1
  cat $O/timeline_form$form.txt | grep -v amdgpu.ids
done
for v in default ko_mfma ko_mem ko_bar; do
  for form in 1 2; do
    if [ $v = default ]; then unset CTG_LIB; else export CTG_LIB=$X/libctg_$v.so; fi
    CTG_STEM_FORM=$form timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only \
        --dump-steps $O/steps_${v}_f$form.json > $O/bench_${v}_f$form.out 2> $O/bench_${v}_f$form.err
    echo "$v form $form: $(python -c "import json; d=json.loads(open('$O/bench_${v}_f$form.out').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms/slice; dominant', d['roofline']['kernel'][-22:], round(d['roofline']['avg_launch_ms'],2), 'ms')" 2>&1 | tail -1)"
  done
done
unset CTG_LIB
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r5_ko"
def load(v, f):
    try:
        return {r["step"]: r for r in json.load(open(f"{O}/steps_{v}_f{f}.json")) if r.get("kind", "").startswith("stem")}
    except Exception:
        return {}
base = load("default", 1)
print("step  K1 N1 | K2 N2            full(f1) full(f2) | noMFMA(f1) noMFMA(f2) | noMEM(f1) noMEM(f2) | noBAR(f1) noBAR(f2)")
for st in sorted(base, key=lambda s: -base[s]["ms"])[:16]:
    row = [base[st]["label"][:28].ljust(28)]
    for v in ("default", "ko_mfma", "ko_mem", "ko_bar"):
        for f in (1, 2):
            d = load(v, f)
            row.append("%8.2f" % d[st]["ms"] if st in d else "     n/a")
    print(st, " ".join(row))
PY
