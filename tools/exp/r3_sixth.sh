#!/bin/bash
O=gpurun_out/r3_sixth; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt
tail -3 $O/pytest.txt
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
for rep in 1 2; do
  for v in main d; do
    lib=vic_amd/libvicgpu.so; [ $v != main ] && lib=vic_amd/libvicgpu_$v.so
    VICGPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$v rep$rep" | tee -a $O/ab.txt || exit 1
  done
done
VICGPU_CHUNKS=2 timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "main chunks2" | tee -a $O/ab.txt || exit 1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace_main -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg > $R/$O/trace_main.log 2>&1 || exit 1
cd $R; python tools/kstats.py $O/trace_main 8 2>/dev/null | head -8
