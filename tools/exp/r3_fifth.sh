#!/bin/bash
O=gpurun_out/r3_fifth; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "newton or headline or node_solvers" > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt
tail -3 $O/pytest.txt
B="--steps 12 --warmup 4 --no-cpu-baseline --no-stream-leg --no-strict-leg --no-compat-leg"
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s ms_per_step %.2f' % ('$1', d['ms_per_step']))"; }
for rep in 1 2; do
  for v in main d; do
    lib=vic_amd/libvicgpu.so; [ $v != main ] && lib=vic_amd/libvicgpu_$v.so
    VICGPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B 2>>$O/ab.err | ms "$v rep$rep" | tee -a $O/ab.txt || exit 1
  done
done
VICGPU_NODE_SOLVER=newton VICGPU_LIB=$PWD/vic_amd/libvicgpu_prof.so timeout -k 10 400 python tools/prof_sections.py --prebuilt --ncell 20000 --steps 4 > $O/prof20k.txt 2>&1 || exit 1
grep -i "cold-nose\|Newton" $O/prof20k.txt
PMC_DIR=r3_fifth/pmc PASSES="1 2 3 6" bash tools/pmc_round.sh
python tools/pmc_summary.py $O/pmc 3 > $O/pmc_summary.txt 2>&1; head -60 $O/pmc_summary.txt
