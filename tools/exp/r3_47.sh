#!/bin/bash
O=gpurun_out/r3_47; mkdir -p $O
timeout -k 10 600 python tools/exp/implicit_time.py 100000 6 2>&1 | tee $O/implicit.txt
