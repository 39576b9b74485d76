#!/bin/bash
# round profile of one GPU's share of cfg4 (trace + PMC passes; the calibration was done with cfg3)
BENCH_ARGS="--config cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg" CALIB=0 ROUND_DIR=round_r03_cfg4 bash tools/profile_round.sh
