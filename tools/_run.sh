#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/dist4; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-cpu-baseline > $O/dist.log 2>&1
grep '^{' $O/dist.log > $O/forcedist.json; python -c "
import json; d=json.load(open('$O/forcedist.json')); print(d['ms_per_step'], d['value'], d.get('allreduce_ms'))"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'])"
