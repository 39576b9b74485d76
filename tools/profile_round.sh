# Round profile of the default bench workload (run on the GPU box: gpurun -- 'bash tools/profile_round.sh').
# 1) kernel trace + stats of `bench.py` (the command the bench line comes from), 2) HBM traffic: FETCH_SIZE and WRITE_SIZE in
# separate PMC passes (they do not fit one pass on gfx950), 3) SQ counters (lanes active, VALU share), 4) the FETCH_SIZE
# calibration kernels (tools/calib/).  Counters and traces never share a pass.  Output under gpurun_out/${ROUND_DIR:-round}/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${ROUND_DIR:-round}
rm -rf $O && mkdir -p $O
ARGS=${BENCH_ARGS:---steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg}
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py $ARGS > $O/bench_trace.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/bench.py $ARGS > $O/bench_fetch.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/bench.py $ARGS > $O/bench_write.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY -d $O/sq -o p --output-format csv -- python3 $R/bench.py $ARGS > $O/bench_sq.log 2>&1 || exit 1
if [ "${CALIB:-1}" = 1 ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/calib/fetch_calib.hip -o /tmp/fetch_calib > $O/calib_build.log 2>&1 \
    && timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/calib -o p --output-format csv -- /tmp/fetch_calib > $O/calib.log 2>&1
fi
echo ok
