#!/bin/bash
# cell chunks x share of resident profile waves: does the evaluation kernel of one chunk hide under the profile solves of another?
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-strict-leg --no-stream-leg 2>/dev/null | python -c "import sys,json; print('%.2f ms/step' % json.loads(sys.stdin.read())['ms_per_step'])"; }
echo "chunks 1 waves 100: $(run)"
for ch in 2 3 4; do for pct in 100 50 34; do echo "chunks $ch waves $pct: $(VICGPU_CHUNKS=$ch VICGPU_PROFILE_WAVES_PCT=$pct run)"; done; done
echo "chunks 1 waves 50: $(VICGPU_PROFILE_WAVES_PCT=50 run)"
