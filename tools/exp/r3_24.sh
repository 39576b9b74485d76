#!/bin/bash
O=gpurun_out/r3_24; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt; tail -15 $O/pytest.txt
