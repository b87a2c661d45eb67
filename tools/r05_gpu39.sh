#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for m in 0 1; do LEGO_ROWS2=$m timeout 300 python tools/rows2_check.py 2>&1 | grep -v amdgpu.ids | grep -E "FAIL|K=300|N=300|K=20|K=4 |N=12|ALL|FAILED|LEGO_ROWS2"; done | tee gpurun_out/r05/rows2_ktail.txt
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "two_workgroups or golden" 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -4
for i in 1 2; do for m in 0 1; do LEGO_ROWS2=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('naml rows2=$m', d['ms_per_step'], d['value'], 'loss', d['final_loss'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('proj_fwd','additive_fwd_item','additive_bwd_data')})"; done; done | tee -a gpurun_out/r05/rows2_ktail.txt
