#!/bin/bash
# round 5, GPU call 1: wino2 correctness + standalone timings per mode, the conv parity tests, in-step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for m in 0 1 2; do LEGO_WINO2=$m timeout 300 python tools/wino2_check.py > gpurun_out/r05/wino2_check_mode$m.txt 2>&1; echo "mode $m rc $?"; tail -4 gpurun_out/r05/wino2_check_mode$m.txt; done
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "winograd or wino or conv" 2>&1 | tail -5
for i in 1 2; do for m in 0 1 2; do LEGO_WINO2=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('wino2=$m', d['ms_per_step'], d['value'], 'conv3_fwd', k['conv3_fwd']['avg_ms'], 'bwd_data', k['conv3_bwd_data']['avg_ms'], 'loss', d['final_loss'])"; done; done
