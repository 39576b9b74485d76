#!/bin/bash
# closing stage kernel at 2 waves per SIMD (256 registers + 1.5 KB scratch) vs 1 wave (512 registers): same-box A/B
O=gpurun_out/r3_30; mkdir -p $O
VICGPU_LIB=$PWD/vic_amd/libvicgpu_st2.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "teacher_forced and frozen and not option" > $O/pytest.txt 2>&1
rc=$?; echo "st2 pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { local label=$1 lib=$2; shift 2; env "$@" VICGPU_LIB=$PWD/vic_amd/libvicgpu$lib.so timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1; }
for rep in 1 2 3; do
  run "main          rep$rep" "" X=1
  run "st2           rep$rep" _st2 X=1
  run "main 1chunk   rep$rep" "" VICGPU_CHUNKS=1
  run "st2 1chunk    rep$rep" _st2 VICGPU_CHUNKS=1
done
