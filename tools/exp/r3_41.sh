#!/bin/bash
# the N > 1 code path over RCCL with one rank (what a 1-GPU box can show): process group "nccl", all_reduce, all_gather, gather to root
O=gpurun_out/r3_41; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-leg --no-stream-leg --no-compat-leg > $O/bench_cfg4_torchrun.json 2> $O/bench_cfg4_torchrun.err; echo "cfg4 torchrun exit $?" | tee -a $O/ab.txt
tail -c 700 $O/bench_cfg4_torchrun.json | head -c 400; echo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --config cfg5 --steps 24 --warmup 6 --no-cpu-baseline > $O/bench_cfg5_torchrun.json 2> $O/bench_cfg5_torchrun.err; echo "cfg5 torchrun exit $?" | tee -a $O/ab.txt
grep -o '"ranks_seen_by_rccl": [^,]*\|"output_gather_ms": [^,]*\|"ms_per_step": [^,]*' $O/bench_cfg4_torchrun.json $O/bench_cfg5_torchrun.json
