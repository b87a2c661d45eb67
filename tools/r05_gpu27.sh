#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1800 python -m pytest tests/test_hip_parity.py tests/test_split_bf16.py -x -q -m gpu -k "nrms or split or mode" 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -6
for e in glove null; do timeout 300 python bench.py --model nrms --embed $e --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nrms $e', d['ms_per_step'], d['value'], 'host', d['host_enqueue_ms_per_step'], 'loss', d['final_loss'])"; done | tee gpurun_out/r05/nrms_keys_fused.txt
