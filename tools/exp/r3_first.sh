#!/bin/bash
# round 3, first GPU call: same-box baseline, instrumented counters, chunk / resident-wave matrix
O=gpurun_out/r3_first; mkdir -p $O
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f kernel %.2f strict %s' % ('$1', d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['config'].get('strict_replay_ms_per_step')))"; }
timeout -k 10 400 python bench.py $B 2>$O/base.err | ms base | tee -a $O/ab.txt || exit 1
VICGPU_LIB=$PWD/vic_amd/libvicgpu_prof.so timeout -k 10 400 python tools/prof_sections.py --prebuilt --ncell 20000 --steps 4 > $O/prof20k.txt 2>&1 || exit 1
B="$B --no-strict-leg"
for cfgs in "2 100" "2 50" "2 70" "3 50" "4 50" "4 30" "1 50"; do
  set -- $cfgs
  VICGPU_CHUNKS=$1 VICGPU_PROFILE_WAVES_PCT=$2 timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "chunks$1_pct$2" | tee -a $O/ab.txt || exit 1
done
timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms base2 | tee -a $O/ab.txt || exit 1
