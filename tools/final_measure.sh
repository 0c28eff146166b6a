#!/bin/bash
# End-of-round measurement on the GPU box: tests, the bench line, a kernel trace of
# the same command and the HBM-traffic counter passes (separate --pmc runs, each
# bounded by its own timeout) for the headline tree, the time-to-solution tree and the peak-rate tree.
# Outputs land in gpurun_out/final/; tools/publish_profiles.py copies what is to be
# judged into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O   # (locally, delete gpurun_out/final before the call: results are merged, not mirrored)
cd $R
if [ "$1" != "notests" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/tests.log 2>&1
  grep -E "passed|failed|error" $O/tests.log | tail -2
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench_line.json
cut -c1-400 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -- python $R/bench.py > $O/trace.log 2>&1
cd $R
T=$(find $O/trace -name "*.db" | head -1)
python tools/rocprof_summary.py $T $O/kernels > /dev/null 2>&1; head -14 $O/kernels_kernels.txt | cut -c1-190
# the headline part alone: its per-kernel averages are what roofline.avg_launch_ms must agree with
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_headline -- python $R/bench.py --headline-only --no-cpu-baseline --steps 4 --warmup 1 > $O/trace_headline.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/trace_headline -name "*.db" | head -1) $O/kernels_headline > /dev/null 2>&1; head -4 $O/kernels_headline_kernels.txt | cut -c1-190
for tree in sycamore_m20_native sycamore_m20_w32_r4 sycamore_m20_w32_c512; do
  CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --tree $R/tests/golden/trees/$tree.json"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$tree -- $CMD > $O/pmc_fetch_$tree.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$tree -- $CMD > $O/pmc_write_$tree.log 2>&1
  cd $R
  F=$(find $O/pmc_fetch_$tree -name "*.db" | head -1); W=$(find $O/pmc_write_$tree -name "*.db" | head -1)
  # (2 timed + 1 warm-up + 1 profiled slice = 4 slices in the run)
  python tools/pmc_traffic.py $F $W 4 $O/pmc_summary_$tree.json $tree.json | tail -8
  timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --headline-only --tree tests/golden/trees/$tree.json --dump-steps $O/steps_$tree.json > /dev/null 2>&1
  python tools/steps_report.py $O/steps_$tree.json 40 > $O/steps_$tree.txt 2>&1
  # the same slice with fp32 products in the fused pairs (CTG_STEM_BF16X3=0; the plan is priced for it and differs)
  CTG_STEM_BF16X3=0 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --headline-only --tree tests/golden/trees/$tree.json --dump-steps $O/steps_${tree}_fp32.json > /dev/null 2>&1
  python tools/steps_report.py $O/steps_${tree}_fp32.json 40 > $O/steps_${tree}_fp32.txt 2>&1
done
# the small configurations: kernel timeline of one contraction / slice batch, per-kernel
# summary, per-step times with the slices batched
bash tools/exp_small_trace.sh C2 C3 C5 > $O/small_trace.log 2>&1
for W in C2 C3 C5; do
  cp $R/gpurun_out/trace_$W/timeline.txt $O/timeline_$W.txt 2>/dev/null
  cp $R/gpurun_out/trace_$W/sum_kernels.txt $O/kernels_$W.txt 2>/dev/null
  timeout 300 python tools/steps_batched.py $W 40 2>&1 | grep -v amdgpu.ids > $O/steps_batched_$W.txt
done
find $O -name "*.db" -delete
find $O -type d -empty -delete
