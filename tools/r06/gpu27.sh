#!/bin/bash
# the driver's 20-step window against the same process's 2 000-step run, five fresh processes on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for i in 1 2 3 4 5; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('window', d['ms_per_step'], d['value'], 'without prewarm', d['without_prewarm']['ms_per_step_without_prewarm'], 'host', d['host_enqueue_ms_per_step'])" | tee -a gpurun_out/r06/window_spread.txt
done
for i in 1 2 3; do
timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 5 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('2000 steps', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r06/window_spread.txt
done
