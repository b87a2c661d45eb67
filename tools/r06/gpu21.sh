#!/bin/bash
# the cleaned per-key in-projection (pair lists through the scalar unit; plain kernel without Dropout): tests, the kernel alone, the NRMS step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/dropcorr_scalar_pairs_final.txt; rm -f $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_plugin_api.py tests/test_train_band.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tee -a $O
python tools/dropcorr_time.py 2>&1 | grep -v amdgpu.ids | head -6 | tee -a $O
for i in 1 2; do
timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('nrms', d['ms_per_step'], d['value'], 'qkv_expand', round(k['qkv_expand_item']['avg_ms']*1e3,1), 'loss', d['final_loss'])" | tee -a $O
done
