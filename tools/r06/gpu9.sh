#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "gather or unique or switches or fixture or naml" 2>&1 | tail -4 | tee gpurun_out/r06/pytest_gpu_9.txt
for i in 1 2; do for cfg in "0 131072" "1 131072" "0 65536" "1 65536"; do set -- $cfg
LEGO_X_TN_MAIN=$1 LEGO_X_TND_MIN_NK=$2 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('tn_main=$1 tnd_min_nk=$2', d['ms_per_step'], d['value'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('conv3_bwd_data','conv3_bwd_weight','additive_bwd_weight_item','additive_bwd_weight_user','proj_bwd_weight')})" | tee -a gpurun_out/r06/naml_tn_ab.txt
done; done
python tools/gather_hbm.py 105600 8 --uniform-only | tee gpurun_out/r06/gather_hbm.json
