#!/bin/bash
# every library variant through the quick teacher-forced cases + the forcing-stream derivation cases
for so in vic_amd/libvicgpu.so tools/exp/variants/*.so; do
  n=$(basename $so .so)
  VICGPU_LIB=$PWD/$so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_forcing_stream.py tests/test_putdata.py -m gpu -q -k "(teacher_forced and not option and not newton) or derivation or device_put" -p no:cacheprovider > gpurun_out/variant_$n.log 2>&1
  echo "$so: $(tail -1 gpurun_out/variant_$n.log)"
done
