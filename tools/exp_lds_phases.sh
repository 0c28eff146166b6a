#!/bin/bash
# time of the LDS-run launch of a small configuration when the components stop after phase p (results are wrong
# then: timing only) -- where inside a component the time goes
R="${GRAFT_REPO_ROOT:-/root/repo}"
which=${1:-C2}
for p in 0 1 2 3 4 5 6 99; do
  echo -n "max phase $p: "
  CTG_LDS_MAX_PHASE=$p python $R/tools/steps_batched.py $which 400 2>/dev/null | grep "lds_run_kernel" | sort -k7 -r | head -1 | awk '{printf "%s %s ms  ", $2, $7}'
  echo
done
