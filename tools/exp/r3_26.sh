#!/bin/bash
# final build of the round: full GPU suite, then the round profile (cfg3: trace + PMC + calibration + 1-chunk trace)
O=gpurun_out/r3_26; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest exit $rc" | tee -a $O/ab.txt; tail -2 $O/pytest.txt
[ $rc -ne 0 ] && exit 1
bash tools/exp/r3_20.sh
