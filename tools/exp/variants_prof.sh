#!/bin/bash
# per-kernel time of each library variant (rocprofv3 kernel trace of a short bench run)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for so in $R/tools/exp/variants/*.so; do
  n=$(basename $so .so)
  export VICGPU_LIB=$so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/vprof_$n -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg > $R/gpurun_out/vprof_$n.log 2>&1
  echo "== $n"; python3 $R/tools/kstats.py $R/gpurun_out/vprof_$n 8 | head -8
done
