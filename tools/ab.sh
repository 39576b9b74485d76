# Same-box A/B of two builds of the library (GPU boxes differ by a few percent, so variants are compared inside one
# gpurun call):  python vic_amd/build.py -f -o vic_amd/libvicgpu_b.so -DVARIANT ; gpurun -- 'bash tools/ab.sh'
ARGS=${BENCH_ARGS:---config cfg3 --steps 4 --warmup 1 --no-cpu-baseline}
for rep in 1 2; do
  for v in a b; do
    lib=vic_amd/libvicgpu.so; [ $v = b ] && lib=vic_amd/libvicgpu_b.so
    VICGPU_LIB=$PWD/$lib timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep$rep ms_per_step %.2f' % d['ms_per_step'])" || exit 1
  done
done
