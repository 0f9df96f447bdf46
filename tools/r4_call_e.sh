#!/bin/bash
# round 4, GPU call E: the bench line, and the two-ranks-on-one-GPU rehearsal of the N > 1 line (shared host segment + hipIpc sub-phase)
O=gpurun_out/r4_e
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 300 $O/bench.err
export GF_BENCH_DEVICE=0 GF_RCCL_TIMEOUT=30
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "2 ranks rc $?"
tail -c 600 $O/bench_2ranks.err
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_3ranks.json 2> $O/bench_3ranks.err; echo "3 ranks rc $?"
tail -c 300 $O/bench_3ranks.err
