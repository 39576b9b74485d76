#!/bin/bash
O=gpurun_out/r3_ninth; mkdir -p $O
timeout -k 10 300 python tools/exp/diag_derive.py > $O/diag_main.txt 2>&1; tail -6 $O/diag_main.txt | cut -c1-200
VICGPU_LIB=$PWD/vic_amd/libvicgpu_o1.so timeout -k 10 300 python tools/exp/diag_derive.py > $O/diag_o1.txt 2>&1; tail -6 $O/diag_o1.txt | cut -c1-200
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg5 or n24 or n21 or blowing" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt
tail -3 $O/pytest.txt
timeout -k 10 600 python bench.py --config cfg5 --steps 48 --warmup 6 > $O/cfg5.json 2>$O/cfg5.err; tail -c 1800 $O/cfg5.json
