#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "in_projection_per_key or nrms_projection_once or trajectory_per_key" 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -8
timeout 300 python tools/dropcorr_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/dropcorr_time.txt
for m in 0 1; do LEGO_NRMS_DROPCORR=$m timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('nrms dropcorr=$m', d['ms_per_step'], d['value'], 'loss', d['final_loss'], {t: round(k[t]['avg_ms']*1e3,1) for t in k if t.startswith('qkv')})"; done | tee gpurun_out/r05/nrms_after_fix.txt
