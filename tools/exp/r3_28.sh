#!/bin/bash
# bench lines of the final build: default (cfg3, all legs), cfg4, cfg5, cfg2
O=gpurun_out/r3_28; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke exit $?" | tee -a $O/ab.txt
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default exit $?" | tee -a $O/ab.txt
cut -c1-300 $O/bench_default.json
timeout -k 10 400 python bench.py --config cfg4 --no-cpu-baseline --no-compat-leg > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "cfg4 exit $?" | tee -a $O/ab.txt
timeout -k 10 600 python bench.py --config cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; echo "cfg5 exit $?" | tee -a $O/ab.txt
timeout -k 10 300 python bench.py --config cfg2 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 exit $?" | tee -a $O/ab.txt
