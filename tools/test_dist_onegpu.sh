#!/bin/bash
# N > 1 code paths on a ONE-GPU box: (a) RCCL with world_size 1 -> the sharded iteration (rollout + record launches, the
# record all-gather, the combine kernel, the env step) captured in a hipGraph; (b) two ranks sharing cuda:0 over gloo ->
# rank / sharding / barrier / MAX-reduce plumbing of bench.py
cd "$(dirname "$0")/.."
echo "== (a) nccl world 1, graph"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 180 python tools/rccl_world1.py
echo "== (b) gloo world 2 on cuda:0"
timeout 300 python bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo --device 0 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-700
