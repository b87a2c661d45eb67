#!/bin/bash
# soak: 30 k NAML steps and 20 k NRMS steps (finite losses, steady step time), then the GPU suite on the current build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/soak.txt; rm -f $O
for m in naml nrms; do
timeout 600 python bench.py --model $m --steps 20000 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$m 20000 steps', d['ms_per_step'], d['value'], 'final loss', d['final_loss'], 'long_run', d.get('long_run'))" | tee -a $O
done
timeout 2400 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3 | tee -a $O
