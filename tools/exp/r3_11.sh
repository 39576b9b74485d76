#!/bin/bash
O=gpurun_out/r3_11; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "frozen or headline or chunking or cfg3 or cfg4 or cfg5 or golden" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt
tail -2 $O/pytest.txt
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
for rep in 1 2; do
  for v in main d; do
    lib=vic_amd/libvicgpu.so; [ $v != main ] && lib=vic_amd/libvicgpu_$v.so
    VICGPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$v rep$rep" | tee -a $O/ab.txt || exit 1
    VICGPU_CHUNKS=1 VICGPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$v 1chunk rep$rep" | tee -a $O/ab.txt || exit 1
  done
done
VICGPU_CHUNKS=1 VICGPU_STATS=1 timeout -k 10 300 python bench.py $B 2>&1 | grep "vicgpu\]" | head -3
