#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_8; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_plugin_api.py tests/test_dp_device.py -q -m gpu -x -k "nrms or null or table or fixture or engine or routes or headline or eval or eight_ranks" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
for i in 1 2; do for q in 1 0; do
LEGO_NRMS_QKV_DEDUP=$q timeout 300 python bench.py --model nrms --embed null --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qkv_dedup=$q', d['value'], d['ms_per_step'], 'loss', d['final_loss'], {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if 'qkv' in k or 'embed' in k})"
done; done 2>&1 | tee $O/nrms_null.txt
