#!/bin/bash
# every library variant through the quick teacher-forced cases
for so in vic_amd/libvicgpu.so tools/exp/variants/*.so; do
  VICGPU_LIB=$PWD/$so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "teacher_forced and not option and not newton" -p no:cacheprovider > gpurun_out/variant_$(basename $so .so).log 2>&1
  echo "$so: $(tail -1 gpurun_out/variant_$(basename $so .so).log)"
done
