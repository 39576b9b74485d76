#!/bin/bash
O=gpurun_out/r3_second; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "newton or headline or node_solvers" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt
tail -5 $O/pytest.txt
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
for rep in 1 2; do
  for v in a b; do
    lib=vic_amd/libvicgpu.so; [ $v = b ] && lib=vic_amd/libvicgpu_b.so
    VICGPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$v rep$rep" | tee -a $O/ab.txt || exit 1
  done
done
VICGPU_CHUNKS=2 timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "a chunks2" | tee -a $O/ab.txt || exit 1
VICGPU_NODE_SOLVER=newton VICGPU_LIB=$PWD/vic_amd/libvicgpu_prof.so timeout -k 10 400 python tools/prof_sections.py --prebuilt --ncell 20000 --steps 4 > $O/prof20k.txt 2>&1 || exit 1
tail -12 $O/prof20k.txt
