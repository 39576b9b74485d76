#!/bin/bash
# pending-list evaluation rounds: small parity subset per variant first, then same-box A/B of the bench (12 steps)
O=gpurun_out/r3_14; mkdir -p $O
T="tests/test_gpu_parity.py"
for v in el1 elp1 el; do
  VICGPU_LIB=$PWD/vic_amd/libvicgpu_$v.so VICGPU_EVAL_LIST_PCT=100 timeout -k 10 300 python -m pytest $T -x -q -k "teacher_forced and frozen and not option" > $O/pytest_$v.txt 2>&1
  rc=$?; echo "$v pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest_$v.txt
  [ $rc -ne 0 ] && exit 1
done
# packing with the identity launch order (irregular lists take this path): the combination of the build that faulted in call 12
VICGPU_LIB=$PWD/vic_amd/libvicgpu_elp1.so VICGPU_NO_XCD_MAP=1 VICGPU_EVAL_LIST_PCT=0 timeout -k 10 300 python -m pytest $T -x -q -k "teacher_forced and frozen and not option" > $O/pytest_elp1_noxcd.txt 2>&1
rc=$?; echo "elp1 noxcd pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest_elp1_noxcd.txt
[ $rc -ne 0 ] && exit 1
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { # label lib env...
  local label=$1 lib=$2; shift 2
  env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu_$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1
}
for rep in 1 2; do
  run "d            rep$rep" d X=1
  run "el1 pct0     rep$rep" el1 VICGPU_EVAL_LIST_PCT=0
  run "el1 pct15    rep$rep" el1 VICGPU_EVAL_LIST_PCT=15
  run "el1 pct30    rep$rep" el1 VICGPU_EVAL_LIST_PCT=30
  run "el1 pct50    rep$rep" el1 VICGPU_EVAL_LIST_PCT=50
  run "el(lag3) p15 rep$rep" el VICGPU_EVAL_LIST_PCT=15
  run "elp1 pct15   rep$rep" elp1 VICGPU_EVAL_LIST_PCT=15
  run "el1 p15 noxcd rep$rep" el1 VICGPU_EVAL_LIST_PCT=15 VICGPU_NO_XCD_MAP=1
  run "el1 p15 1chunk rep$rep" el1 VICGPU_EVAL_LIST_PCT=15 VICGPU_CHUNKS=1
done
