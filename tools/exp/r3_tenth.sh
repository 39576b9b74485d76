#!/bin/bash
O=gpurun_out/r3_tenth; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt
grep -E "^FAILED|passed|failed" $O/pytest.txt | tail -8
ROUND_DIR=r3_round bash tools/profile_round.sh
R=$PWD
(cd /tmp && export TMPDIR=/tmp && VICGPU_CHUNKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_round/trace_1chunk -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $R/gpurun_out/r3_round/bench_trace_1chunk.log 2>&1)
timeout -k 10 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; tail -c 3000 $O/bench_default.json
