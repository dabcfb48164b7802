#!/bin/bash
# usage (on the GPU box): tools/two_rank_rehearsal.sh > gpurun_out/two_rank.txt -- profiles/r04_two_rank_rehearsal.txt
# two ranks on the one GPU of the box, gloo for the collectives: the launch contract, timing, shards and gathers of bench.py --gpus 2
cd $GRAFT_REPO_ROOT
port=29611
for args in "--config 2 --batch 64 --steps 2 --warmup 1" "--config 4 --batch 16 --T 8192 --steps 1 --warmup 1" "--config 5 --batch 8 --T 8192 --steps 1 --warmup 1" "--config 4 --method locomp --batch 16 --T 8192 --steps 1 --warmup 1" "--config 5 --method locomp --batch 8 --T 8192 --steps 1 --warmup 1"; do
  echo "== $args"
  port=$((port+1))
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 $args --backend gloo --no-cpu-baseline --no-secondary 2>gpurun_out/rehearse.err | python -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); print({k:d[k] for k in ('n_gpus','value','ms_per_step','gather') if k in d})
" || { echo FAILED; tail -5 gpurun_out/rehearse.err; exit 1; }
done
