#!/bin/bash
# round profile of the final build: cfg3 default (2 chunks) trace + PMC passes + calibration, then the 1-chunk trace
ROUND_DIR=round_r03 bash tools/profile_round.sh || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round_r03
(cd /tmp && export TMPDIR=/tmp && VICGPU_CHUNKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace_1chunk -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $O/bench_trace_1chunk.log 2>&1)
python tools/kstats.py gpurun_out/round_r03/trace_1chunk 8 | head -12
