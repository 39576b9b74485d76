#!/bin/bash
# parked context with 8 (4) consecutive words per lane instead of 2: dense rounds vs sparse rounds, threshold sweep (same box)
O=gpurun_out/r3_31; mkdir -p $O
for v in g8 g4; do
  VICGPU_EVAL_LIST_PCT=50 VICGPU_LIB=$PWD/vic_amd/libvicgpu_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "teacher_forced and frozen and not option" > $O/pytest_$v.txt 2>&1
  rc=$?; echo "$v pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest_$v.txt
  [ $rc -ne 0 ] && exit 1
done
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { local label=$1 lib=$2; shift 2; env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1; }
for rep in 1 2; do
  run "main            rep$rep" "" X=1
  for pct in 4 15 30 50; do
    run "g8 pct$pct        rep$rep" _g8 VICGPU_EVAL_LIST_PCT=$pct
  done
  run "g4 pct4         rep$rep" _g4 VICGPU_EVAL_LIST_PCT=4
  run "g4 pct30        rep$rep" _g4 VICGPU_EVAL_LIST_PCT=30
  run "main 1chunk     rep$rep" "" VICGPU_CHUNKS=1
  run "g8 p4 1chunk    rep$rep" _g8 VICGPU_CHUNKS=1 VICGPU_EVAL_LIST_PCT=4
  run "g8 p30 1chunk   rep$rep" _g8 VICGPU_CHUNKS=1 VICGPU_EVAL_LIST_PCT=30
done
