# PMC passes over the finite-difference pipeline kernels (run on the GPU box: gpurun -- 'bash tools/pmc_split.sh')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PASSES=${PASSES:-"1 2"}
for i in $PASSES; do
  case $i in
    1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU";;
    2) C="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU";;
    3) C="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_INST_LDS";;
  esac
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_split$i -o p --output-format csv -- python3 $R/bench.py --config cfg3 ${BENCH_ARGS:---ncell 20000} --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc$i.log 2>&1 || exit 1
done
echo ok
