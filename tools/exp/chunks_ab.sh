#!/bin/bash
# cell chunks as independent pipelines: same-box A/B on the default bench workload
run() { python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-strict-leg --no-stream-leg 2>/dev/null | python -c "import sys,json; print('%.2f ms/step' % json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "chunks 1: $(VICGPU_CHUNKS=1 run)   chunks 2: $(VICGPU_CHUNKS=2 run)"
done
