#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for n in 1 2 3 4 6 8 12 16 24 32 48 64 96; do for m in 0 1; do BENCH_PREWARM_STEPS=0 BENCH_SPIN_MS=0 LEGO_NRMS_DROPCORR=$m timeout 300 python bench.py --model nrms --steps $n --warmup 0 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nrms steps $n dropcorr=$m loss', d['final_loss'])"; done; done | tee gpurun_out/r05/nrms_loss_trace.txt
