#!/bin/bash
# evaluation records (the root find ends in the round that finds the root): parity subset, same-box A/B against the committed build, round trace
O=gpurun_out/r3_21; mkdir -p $O
T="tests/test_gpu_parity.py"
VICGPU_LIB=$PWD/vic_amd/libvicgpu_evr.so timeout -k 10 600 python -m pytest $T -x -q -k "teacher_forced and (frozen or gf_ or noflux or exp_trans or tfallback or fail or stress)" > $O/pytest_evr.txt 2>&1
rc=$?; echo "evr pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest_evr.txt
[ $rc -ne 0 ] && exit 1
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { # label lib env...
  local label=$1 lib=$2; shift 2
  env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1
}
for rep in 1 2 3; do
  run "main          rep$rep" "" X=1
  run "evr           rep$rep" _evr X=1
  run "main 1chunk   rep$rep" "" VICGPU_CHUNKS=1
  run "evr 1chunk    rep$rep" _evr VICGPU_CHUNKS=1
done
VICGPU_LIB=$PWD/vic_amd/libvicgpu_evr.so VICGPU_TRACE_ROUNDS=1 VICGPU_CHUNKS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 7 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg > $O/bench.log 2> $O/rounds.txt
tail -24 $O/rounds.txt
