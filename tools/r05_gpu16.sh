#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 3000 python -m pytest tests/ -x -q -m gpu --deselect tests/test_train_band.py 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -8 | tee gpurun_out/r05/pytest_gpu_b.txt
for i in 1 2; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('naml', d['ms_per_step'], d['value'], 'loss', d['final_loss'], 'gather', d['roofline_gather'].get('avg_launch_ms'))"; done | tee gpurun_out/r05/naml_after_gather.txt
for m in 0 1; do LEGO_NRMS_DROPCORR=$m timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nrms dropcorr=$m', d['ms_per_step'], d['value'], 'loss', d['final_loss'])"; done | tee gpurun_out/r05/nrms_after_fix.txt
