#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of the library (device code is
# compiled as usual: -fno-gpu-sanitize) and the host tests that drive the C ABI without a GPU
# under it:   bash tools/build_asan.sh [pytest args]
# Plan validation, the path-search code and the checkpoint / state entry points are plain C++ on the
# host; this is where an out-of-range table offset or a bad record would do damage.
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/cotengra_amd/csrc; O=$R/cotengra_amd/lib/obj/asan; L=$R/cotengra_amd/lib/exp/libctg_asan.so
mkdir -p $O $(dirname $L)
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -shared-libsan"
pids=()
for f in $S/*.hip $S/*.cpp; do
  o=$O/$(basename ${f%.*}).o
  if [ ! -e $o ] || [ $f -nt $o ]; then /opt/rocm/bin/hipcc $FLAGS -I$R/include -c $f -o $o & pids+=($!); fi
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $L $O/*.o -ldl || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $R
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT CTG_LIB=$L \
  python -m pytest -q -m "not gpu" -p no:cacheprovider ${@:-tests/test_cabi.py tests/test_host_round3.py tests/test_pathfind.py tests/test_host_round2.py}
