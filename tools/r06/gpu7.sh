#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for i in 1 2 3; do for f in 0 1; do
LEGO_X_TND_FIRST=$f timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('tnd_first=$f', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r06/nrms_tnd_first.txt
done; done
timeout 900 python tools/gather_sweep.py 2>&1 | tee gpurun_out/r06/gather_sweep.txt
