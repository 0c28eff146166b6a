#!/bin/bash
# End-of-round-6 measurement on the GPU box: the whole GPU test suite, smoke, the HBM-traffic counter passes of the
# headline tree (separate --pmc runs; their summary goes to profiles/ FIRST, so that the bench line's roofline.traffic is
# this build's), the bench line (compact + full), a kernel trace of the headline part, the per-step table.
# Outputs land in gpurun_out/final/; tools/publish_profiles.py r5 copies what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
if [ "$1" != "notests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=8 > $O/tests.log 2>&1
  grep -E "passed|failed|error" $O/tests.log | tail -3
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for tree in ${TREES:-sycamore_m20_native}; do
  CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --tree $R/tests/golden/trees/$tree.json"
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$tree -- $CMD > $O/pmc_fetch_$tree.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$tree -- $CMD > $O/pmc_write_$tree.log 2>&1
  cd $R
  F=$(find $O/pmc_fetch_$tree -name "*.db" | head -1); W=$(find $O/pmc_write_$tree -name "*.db" | head -1)
  # (2 timed + 1 warm-up + 1 profiled slice = 4 slices in the run)
  python tools/pmc_traffic.py $F $W 4 $O/pmc_summary_$tree.json $tree.json | tail -6
  cp $O/pmc_summary_$tree.json $R/profiles/pmc_summary_$tree.json
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --headline-only --tree tests/golden/trees/$tree.json --dump-steps $O/steps_$tree.json > /dev/null 2>&1
  python tools/steps_report.py $O/steps_$tree.json 40 > $O/steps_$tree.txt 2>&1
done
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench_compact.json
cp $R/bench_full.json $O/bench_line.json
cut -c1-600 $O/bench_compact.json; echo; wc -c $O/bench_compact.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_headline -- python $R/bench.py --headline-only --no-cpu-baseline --steps 4 --warmup 1 > $O/trace_headline.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/trace_headline -name "*.db" | head -1) $O/kernels_headline > /dev/null 2>&1; head -6 $O/kernels_headline_kernels.txt | cut -c1-200
# the small configurations: per-step tables with their slices batched, and the kernel trace of C2
for c in C2 C3 C5; do timeout 300 python tools/steps_batched.py $c 30 > $O/steps_batched_$c.txt 2>&1; done
# the three arithmetics of the stem kernels, same box
timeout 900 ./tools/exp_r6_h2.sh > $O/arithmetics.txt 2>&1; tail -4 $O/arithmetics.txt | cut -c1-220
find $O -name "*.db" -delete
find $O -type d -empty -delete
