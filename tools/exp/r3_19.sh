#!/bin/bash
# what every round leaves pending (one step of the bench workload, 1 chunk), then the launch-shape test and the 2-rank test
O=gpurun_out/r3_19; mkdir -p $O
VICGPU_TRACE_ROUNDS=1 VICGPU_CHUNKS=1 timeout -k 10 300 python bench.py --steps 2 --warmup 6 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg > $O/bench.log 2> $O/rounds.txt
tail -40 $O/rounds.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank_gpu.py -x -q -m gpu -k "launch_shapes or two_ranks" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt; tail -2 $O/pytest.txt
