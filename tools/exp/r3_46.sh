#!/bin/bash
# ten model days of the bench workloads in one run: error flags, mean step time over the whole diurnal cycle
O=gpurun_out/r3_46; mkdir -p $O
for c in cfg3 cfg4; do
  timeout -k 10 500 python bench.py --config $c --steps 240 --warmup 4 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c exit $?" | tee -a $O/ab.txt
  python -c "import json; d=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); c=d['config']; print('$c 240 steps: %.2f ms/step, %.3f M cell-steps/s, cells with error flags %s, mean runoff %.5f mm/step, mean SWE at end %.3f mm' % (d['ms_per_step'], d['value']/1e6, c.get('cells_with_error_flags'), c.get('mean_runoff_mm_per_step'), c.get('mean_swe_mm_end')))" | tee -a $O/ab.txt
done
