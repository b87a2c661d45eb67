#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_plugin_api.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
run() { timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], d['value'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('proj_bwd_segsum','proj_bwd_weight','proj_expand') if t in k}, d['final_loss'])"; }
for i in 1 2; do
LEGO_SEGSUM_VEC=4 run "segsum vec4"
LEGO_SEGSUM_VEC=1 run "segsum vec1"
done | tee gpurun_out/r05/segsum_ab.txt
for v in 4 1; do LEGO_SEGSUM_VEC=$v timeout 300 python bench.py --model nrms --embed null --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nrms null vec$v', d['ms_per_step'], d['value'])"; done | tee -a gpurun_out/r05/segsum_ab.txt
