# PMC passes over one bench workload at full size (run on the GPU box: gpurun -- 'bash tools/pmc_round.sh').  Counters only
# (--kernel-trace for the names; no other trace domain).  Output: gpurun_out/${PMC_DIR:-pmc}/passN/, summary by tools/pmc_summary.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${PMC_DIR:-pmc}; mkdir -p $O
ARGS=${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg}
PASSES=${PASSES:-"1 2 3"}
for i in $PASSES; do
  case $i in
    1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY";;
    2) C="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM";;
    3) C="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT";;
    4) C="FETCH_SIZE";;
    5) C="WRITE_SIZE";;
    6) C="TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum";;
  esac
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C -d $O/pass$i -o p --output-format csv -- python3 $R/bench.py $ARGS > $O/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pass$i.log; }
done
echo ok
