#!/bin/bash
# IMPLICIT with lane-persistent trial loop: parity tests that use it, then its step time on the cfg3 domain
O=gpurun_out/r3_48; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "implicit or combo" > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $O/ab.txt; tail -1 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python tools/exp/implicit_time.py 100000 6 2>&1 | tee $O/implicit.txt
