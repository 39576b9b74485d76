#!/bin/bash
# same-box A/B of library variants on the bench workloads (ms per step; default cfg3)
for so in tools/exp/variants/*.so; do
  n=$(basename $so .so)
  for cfg in ${CFGS:-cfg3}; do
    VICGPU_LIB=$PWD/$so timeout -k 10 300 python bench.py --config $cfg --steps 12 --warmup 4 --no-cpu-baseline --no-strict-leg --no-stream-leg > gpurun_out/vbench_${n}_$cfg.json 2> gpurun_out/vbench_${n}_$cfg.err
    python - <<PY
import json
try:
    j = json.load(open("gpurun_out/vbench_${n}_$cfg.json"))
    print("$so $cfg", "ms/step %.3f" % j["ms_per_step"])
except Exception as e:
    print("$so $cfg failed", e)
PY
  done
done
