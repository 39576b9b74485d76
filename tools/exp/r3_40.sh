#!/bin/bash
# the GPU parity suite under the tuning knobs' extremes: one chunk / three chunks, sparse rounds always / never, identity launch order
O=gpurun_out/r3_40; mkdir -p $O
run() { local label=$1; shift; env "$@" timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size and not launch_shapes and not chunking" > $O/pytest_$label.txt 2>&1; rc=$?; echo "$label exit $rc: $(tail -1 $O/pytest_$label.txt)" | tee -a $O/ab.txt; [ $rc -ne 0 ] && exit 1; return 0; }
run chunks1 VICGPU_CHUNKS=1 || exit 1
run chunks3_sparse_always VICGPU_CHUNKS=3 VICGPU_EVAL_LIST_PCT=100 || exit 1
run sparse_never_noxcd VICGPU_EVAL_LIST_PCT=0 VICGPU_NO_XCD_MAP=1 || exit 1
