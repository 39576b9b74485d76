#!/bin/bash
# sparse evaluation rounds from the work list + evaluation-only list (no extra atomics): parity subset, then same-box A/B;
# last: the reconstructed source of the call-12 build that faulted (pack + identity order), -O1 first
O=gpurun_out/r3_15; mkdir -p $O
T="tests/test_gpu_parity.py"
for v in main l1; do
  lib=vic_amd/libvicgpu.so; [ $v != main ] && lib=vic_amd/libvicgpu_$v.so
  VICGPU_LIB=$PWD/$lib VICGPU_EVAL_LIST_PCT=100 timeout -k 10 300 python -m pytest $T -x -q -k "teacher_forced and frozen and not option" > $O/pytest_$v.txt 2>&1
  rc=$?; echo "$v pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest_$v.txt
  [ $rc -ne 0 ] && exit 1
done
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { # label lib env...
  local label=$1 lib=$2; shift 2
  env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1
}
for rep in 1 2; do
  run "d             rep$rep" _d X=1
  run "main pct0     rep$rep" "" VICGPU_EVAL_LIST_PCT=0
  run "main pct15    rep$rep" "" VICGPU_EVAL_LIST_PCT=15
  run "main pct30    rep$rep" "" VICGPU_EVAL_LIST_PCT=30
  run "l1 pct15      rep$rep" _l1 VICGPU_EVAL_LIST_PCT=15
  run "l1 pct30      rep$rep" _l1 VICGPU_EVAL_LIST_PCT=30
  run "main p15 noxcd rep$rep" "" VICGPU_EVAL_LIST_PCT=15 VICGPU_NO_XCD_MAP=1
  run "main p15 1chunk rep$rep" "" VICGPU_EVAL_LIST_PCT=15 VICGPU_CHUNKS=1
  run "main p0 1chunk rep$rep" "" VICGPU_EVAL_LIST_PCT=0 VICGPU_CHUNKS=1
done
TT="tests/test_gpu_parity.py::test_teacher_forced[frozen_fixed-brent]"
for v in bad_o1 bad; do
  VICGPU_LIB=$PWD/vic_amd/libvicgpu_$v.so timeout -k 10 120 python -m pytest "$TT" -x -q > $O/pytest_$v.txt 2>&1; rc=$?
  echo "$v exit $rc" | tee -a $O/ab.txt; grep -E "Error|passed|failed|fault" $O/pytest_$v.txt | head -3
  [ $rc -ne 0 ] && exit 0
done
