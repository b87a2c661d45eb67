#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for wgs in 2 3 4 6; do echo "== LEGO_ROWS2_WGS=$wgs"; LEGO_ROWS2_WGS=$wgs timeout 300 python tools/rows2_check.py 2>&1 | grep "us " | head -3; done | tee gpurun_out/r05/rows2_wgs.txt
for i in 1 2; do for m in 0 1; do LEGO_ROWS2=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('naml rows2=$m', d['ms_per_step'], d['value'], 'loss', d['final_loss'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('additive_fwd_item','additive_bwd_data')})"; done; done | tee gpurun_out/r05/rows2_step_ab.txt
