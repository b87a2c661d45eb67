#!/bin/bash
# round 5, GPU call: LDS-free weight-gradient kernel (gemm_tnd.hpp): correctness + timings vs the tile kernels, in-step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for m in 0 1; do LEGO_TND=$m timeout 300 python tools/tnd_check.py > gpurun_out/r05/tnd_check_mode$m.txt 2>&1; echo "mode $m rc $?"; cat gpurun_out/r05/tnd_check_mode$m.txt | grep -v "^ok   R=.*[^s]$" | tail -12; done
for w in 256 1024; do echo "LEGO_TND_WGS=$w"; LEGO_TND_WGS=$w timeout 300 python tools/tnd_check.py --time-only 2>&1 | grep "us"; done | tee gpurun_out/r05/tnd_wgs.txt
for i in 1 2; do for m in 0 1; do LEGO_TND=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('tnd=$m', d['ms_per_step'], d['value'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('additive_bwd_weight_item','proj_bwd_weight','conv3_bwd_weight','additive_bwd_weight_user') if t in k}, 'loss', d['final_loss'])"; done; done | tee gpurun_out/r05/tnd_step_ab.txt
for m in 0 1; do LEGO_TND=$m timeout 300 python bench.py --model nrms --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nrms tnd=$m', d['ms_per_step'], d['value'], 'loss', d['final_loss'])"; done | tee -a gpurun_out/r05/tnd_step_ab.txt
