#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
run() { timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], d['value'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('additive_bwd_weight_item','proj_bwd_weight','conv3_bwd_weight','additive_bwd_weight_user') if t in k})"; }
for i in 1 2; do
LEGO_TND=0 run "tnd=0"
LEGO_TND=1 run "tnd=1"
LEGO_TND=1 LEGO_TND_MAX_NK=65536 run "tnd=1 only 256x256"
LEGO_TND=1 LEGO_TND_MAX_NK=65536 LEGO_TND_WGS=256 run "tnd=1 only 256x256 wgs256"
LEGO_TND=1 LEGO_TND_WGS=256 run "tnd=1 wgs256"
done | tee gpurun_out/r05/tnd_step_ab2.txt
