#!/bin/bash
# the monolithic QUICK_FLUX kernel on a domain that fills the GPU (cfg2 itself is 469 waves: a latency figure)
O=gpurun_out/r3_45; mkdir -p $O
for n in 100000 500000; do
  timeout -k 10 400 python bench.py --config cfg2 --ncell $n --no-cpu-baseline --no-stream-leg > $O/bench_cfg2_$n.json 2> $O/bench_cfg2_$n.err; echo "cfg2 ncell $n exit $?" | tee -a $O/ab.txt
  python -c "import json; d=json.loads(open('$O/bench_cfg2_$n.json').read().strip().splitlines()[-1]); print('cfg2 at $n cells: %.3f ms/step, %.1f M cell-steps/s, roofline frac %.4f' % (d['ms_per_step'], d['value']/1e6, d['roofline']['frac']))" | tee -a $O/ab.txt
done
