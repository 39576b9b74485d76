#!/bin/bash
# per-kernel effect of the lane-group size of the parked context: 1-chunk kernel traces of G = 2 and G = 8 on the same box, twice
O=gpurun_out/r3_34; mkdir -p $O
R=$PWD
for rep in 1 2; do
for v in g2 main; do
  lib=$R/vic_amd/libvicgpu.so; [ $v != main ] && lib=$R/vic_amd/libvicgpu_$v.so
  (cd /tmp && export TMPDIR=/tmp && VICGPU_LIB=$lib VICGPU_CHUNKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace_${v}_$rep -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $R/$O/bench_${v}_$rep.log 2>&1) || exit 1
  echo "== $v rep $rep" | tee -a $O/kstats.txt
  python tools/kstats.py $O/trace_${v}_$rep 8 | head -6 | tee -a $O/kstats.txt
done
done
