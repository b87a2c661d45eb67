#!/bin/bash
# round 5, GPU call 2: wino2 (branch-free k loop, counted waits) correctness + timing, ablation variants (tune library), in-step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for m in 0 1; do LEGO_WINO2=$m timeout 300 python tools/wino2_check.py > gpurun_out/r05/wino2b_check_mode$m.txt 2>&1; echo "mode $m rc $?"; tail -3 gpurun_out/r05/wino2b_check_mode$m.txt; done
for a in 0 128 256 1; do echo "ABL=$a"; LEGO_HIP_LIB=$GRAFT_REPO_ROOT/legommenders_amd/csrc/liblego_hip_tune.so LEGO_WINO2_ABL=$a timeout 120 python tools/wino2_check.py --time-only 2>&1 | grep "p=0.1" | sed 's/.*| fwd/fwd/'; done 2>&1 | tee gpurun_out/r05/wino2_ablation.txt
for i in 1 2; do for m in 0 1; do LEGO_WINO2=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('wino2=$m', d['ms_per_step'], d['value'], 'conv3_fwd', k['conv3_fwd']['avg_ms'], 'bwd_data', k['conv3_bwd_data']['avg_ms'], 'loss', d['final_loss'])"; done; done | tee gpurun_out/r05/wino2b_step_ab.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r05/pytest_gpu_a.txt
