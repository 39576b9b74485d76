#!/bin/bash
# full GPU suite on the round's build, then a 1-chunk kernel trace of the bench
O=gpurun_out/r3_16; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $O/ab.txt; tail -2 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
R=$PWD
(cd /tmp && export TMPDIR=/tmp && VICGPU_CHUNKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/$O/trace_1chunk -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $R/$O/bench_trace_1chunk.log 2>&1)
python tools/kstats.py $O/trace_1chunk 8 | head -12 | tee $O/kstats.txt
