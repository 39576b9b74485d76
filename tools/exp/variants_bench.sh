#!/bin/bash
for so in tools/exp/variants/*.so; do
  VICGPU_LIB=$PWD/$so timeout -k 10 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/vbench_$(basename $so .so).json 2> gpurun_out/vbench_$(basename $so .so).err
  python - <<PY
import json
try:
    j = json.load(open("gpurun_out/vbench_$(basename $so .so).json"))
    print("$so", "ms/step %.2f" % j["ms_per_step"], "strict %.2f" % (j["config"].get("strict_replay_ms_per_step") or 0))
except Exception as e:
    print("$so failed", e)
PY
done
