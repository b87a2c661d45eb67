#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for i in 1 2; do for n in 256 768; do LEGO_ROWS2_MAX_N=$n timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('nrms max_n=$n', d['ms_per_step'], d['value'], 'loss', d['final_loss'], {t: round(k[t]['avg_ms']*1e3,1) for t in k if t.startswith('qkv')})"; done; done | tee gpurun_out/r05/rows2_maxn.txt
for n in 256 768; do LEGO_ROWS2_MAX_N=$n timeout 300 python bench.py --model nrms --embed null --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nrms null max_n=$n', d['ms_per_step'], d['value'])"; done | tee -a gpurun_out/r05/rows2_maxn.txt
for n in 256 768 3072; do LEGO_ROWS2_MAX_N=$n timeout 600 python tools/bert_naml_bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170; done | tee -a gpurun_out/r05/rows2_maxn.txt
