#!/bin/bash
# Round 5: socket power and shader clock (rocm-smi, ~3 samples/s) while bench.py times slices of the headline tree
# in the driver's form -- the power-bound reading of DESIGN 4.2 (profiles/r5_power_trace.txt).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/power; mkdir -p $O
STEPS=${1:-20}
python $R/bench.py --headline-only --no-cpu-baseline --steps $STEPS --warmup 5 > $O/bench.out 2> $O/bench.err &
BP=$!
: > $O/smi.csv
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> $O/smi.csv
  timeout 5 rocm-smi --showclocks --showpower --showuse --csv 2>/dev/null | grep -v "^$" >> $O/smi.csv
  sleep 0.2
done
wait $BP
python - "$O" <<'PY'
import sys, re, json
O = sys.argv[1]
rows, t, hdr = [], None, None
for line in open(O + "/smi.csv"):
    line = line.strip()
    if line.startswith("t="):
        t = float(line[2:]); continue
    if line.startswith("device"):
        hdr = line.split(","); continue
    if hdr and line.startswith("card"):
        d = dict(zip(hdr, line.split(",")))
        pw = next((v for k, v in d.items() if "ower" in k and "(W)" in k), None)
        ck = next((v for k, v in d.items() if "sclk" in k.lower()), None)
        use = next((v for k, v in d.items() if "GPU use" in k), None)
        rows.append((t, pw, ck, use))
t0 = rows[0][0] if rows else 0
print("# t_s  power_W  sclk  gpu_use_%")
for t, pw, ck, use in rows:
    print("%6.2f  %s  %s  %s" % (t - t0, pw, ck, use))
busy = [float(pw) for _, pw, _, use in rows if pw and use and use.strip().isdigit() and int(use) >= 90]
if busy:
    busy.sort()
    print("# samples at >= 90 %% use: %d, power median %.0f W, max %.0f W" % (len(busy), busy[len(busy) // 2], busy[-1]))
last = open(O + "/bench.out").read().strip().splitlines()[-1]
r = json.loads(last)
print("# bench: %.1f ms/slice over %d slices, dominant pair %.2f ms, roofline.frac %.3f" % (
    r["ms_per_step"], r["steps"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"]))
PY
