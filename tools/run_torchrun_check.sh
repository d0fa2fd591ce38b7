#!/bin/bash
# single-GPU sanity of the distributed bench path: one rank over RCCL (the driver runs N = 2, 4, 8)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline
