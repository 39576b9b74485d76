#!/bin/bash
# chunk count / resident profile waves sweep on the final build (same box)
O=gpurun_out/r3_25; mkdir -p $O
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
run() { local label=$1; shift; env "$@" timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$label" | tee -a $O/ab.txt || exit 1; }
for rep in 1 2; do
  run "chunks2            rep$rep" VICGPU_CHUNKS=2
  run "chunks3            rep$rep" VICGPU_CHUNKS=3
  run "chunks4            rep$rep" VICGPU_CHUNKS=4
  run "chunks2 waves50    rep$rep" VICGPU_CHUNKS=2 VICGPU_PROFILE_WAVES_PCT=50
  run "chunks3 waves50    rep$rep" VICGPU_CHUNKS=3 VICGPU_PROFILE_WAVES_PCT=50
  run "chunks4 waves50    rep$rep" VICGPU_CHUNKS=4 VICGPU_PROFILE_WAVES_PCT=50
  run "chunks4 waves25    rep$rep" VICGPU_CHUNKS=4 VICGPU_PROFILE_WAVES_PCT=25
done
