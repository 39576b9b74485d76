#!/bin/bash
# does today's source still need -mllvm -sgpr-regalloc=basic?  the parity suite on a build with hipcc's default (greedy) SGPR allocator
O=gpurun_out/r3_44; mkdir -p $O
VICGPU_LIB=$PWD/vic_amd/libvicgpu_greedy.so timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "not full_size" > $O/pytest_greedy.txt 2>&1; echo "greedy pytest exit $?" | tee -a $O/ab.txt; tail -25 $O/pytest_greedy.txt | cut -c1-200
