#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_2; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_custom_ops.py tests/test_plugin_api.py -x -q -m gpu -k "mhsa or nrms or mha or attention or headline or opcheck or plugin" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python tools/mhsa_bulk_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/mhsa_probe.txt
for i in 1 2 3; do for rc in 0 1; do
LEGO_MHSA_RECOMPUTE=$rc timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nrms recompute=$rc', d['value'], d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items() if 'mhsa' in k})"
done; done 2>&1 | tee $O/nrms.txt
