#!/bin/bash
# final evaluations served like the others (class-conditional loads, partial write-back): parity subset, same-box A/B against the previous build, 1-chunk kernel stats
O=gpurun_out/r3_42; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_shapes or chunking or (teacher_forced and frozen and not option) or implicit" > $O/pytest.txt 2>&1
rc=$?; echo "pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { local label=$1 lib=$2; shift 2; env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1; }
for rep in 1 2 3; do
  run "prev          rep$rep" _d X=1
  run "new           rep$rep" "" X=1
  run "prev 1chunk   rep$rep" _d VICGPU_CHUNKS=1
  run "new 1chunk    rep$rep" "" VICGPU_CHUNKS=1
done
R=$PWD
(cd /tmp && export TMPDIR=/tmp && VICGPU_CHUNKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace_1chunk -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $R/$O/bench_trace_1chunk.log 2>&1)
python tools/kstats.py $O/trace_1chunk 8 | head -6 | tee $O/kstats.txt
