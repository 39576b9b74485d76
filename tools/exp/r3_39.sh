#!/bin/bash
# how late the host learns the pending counts (grid of the sparse rounds): read-back lag 3 (built in) vs 2 vs 1, same box
O=gpurun_out/r3_39; mkdir -p $O
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { local label=$1 lib=$2; shift 2; env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1; }
VICGPU_LIB=$PWD/vic_amd/libvicgpu_lag1.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "launch_shapes or (teacher_forced and frozen and not option)" > $O/pytest.txt 2>&1
rc=$?; echo "lag1 pytest exit $rc" | tee -a $O/ab.txt; [ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do
  run "lag3          rep$rep" "" X=1
  run "lag2          rep$rep" _lag2 X=1
  run "lag1          rep$rep" _lag1 X=1
  run "lag3 1chunk   rep$rep" "" VICGPU_CHUNKS=1
  run "lag2 1chunk   rep$rep" _lag2 VICGPU_CHUNKS=1
  run "lag1 1chunk   rep$rep" _lag1 VICGPU_CHUNKS=1
done
