#!/bin/bash
# device-side switch to sparse evaluation rounds: parity subset, threshold sweep (same box), 1-chunk trace
O=gpurun_out/r3_18; mkdir -p $O
T="tests/test_gpu_parity.py"
VICGPU_EVAL_LIST_PCT=100 timeout -k 10 300 python -m pytest $T -x -q -k "teacher_forced and frozen and not option" > $O/pytest_main.txt 2>&1
rc=$?; echo "main pct100 pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest_main.txt
[ $rc -ne 0 ] && exit 1
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { # label lib env...
  local label=$1 lib=$2; shift 2
  env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1
}
for rep in 1 2; do
  run "d             rep$rep" _d X=1
  for pct in 0 15 25 35 50 70; do
    run "main pct$pct    rep$rep" "" VICGPU_EVAL_LIST_PCT=$pct
  done
  run "main p35 1chunk rep$rep" "" VICGPU_EVAL_LIST_PCT=35 VICGPU_CHUNKS=1
  run "main p0 1chunk rep$rep" "" VICGPU_EVAL_LIST_PCT=0 VICGPU_CHUNKS=1
done
R=$PWD
(cd /tmp && export TMPDIR=/tmp && VICGPU_EVAL_LIST_PCT=35 VICGPU_CHUNKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace_1chunk -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $R/$O/bench_trace_1chunk.log 2>&1)
python tools/kstats.py $O/trace_1chunk 8 | head -8 | tee $O/kstats.txt
