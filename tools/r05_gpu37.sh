#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for sg in 0 2 4 6 8 12; do echo "== LEGO_ROWS2_STAGGER=$sg"; LEGO_ROWS2_STAGGER=$sg timeout 300 python tools/rows2_check.py 2>&1 | grep "us " | grep "R=27613" ; done | tee gpurun_out/r05/rows2_stagger.txt
for sg in 0 4 8; do LEGO_ROWS2_STAGGER=$sg timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('naml stagger=$sg', d['ms_per_step'], d['value'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('additive_fwd_item','additive_bwd_data')})"; done | tee -a gpurun_out/r05/rows2_stagger.txt
