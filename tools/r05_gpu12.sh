#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for i in 1 2 3; do for m in 0 1; do LEGO_PROJ_SIDE=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('naml proj_side=$m', d['ms_per_step'], d['value'], 'loss', d['final_loss'])"; done; done | tee gpurun_out/r05/proj_side_ab.txt
for v in "LEGO_NRMS_DEDUP=0" "LEGO_TND=0" "LEGO_NRMS_DROPCORR=0" "LEGO_NRMS_DROPCORR=1"; do env $v timeout 300 python bench.py --model nrms --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nrms $v', d['ms_per_step'], d['value'], 'loss', d['final_loss'])"; done | tee gpurun_out/r05/nrms_loss_variants.txt
