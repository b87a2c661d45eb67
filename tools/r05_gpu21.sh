#!/bin/bash
# final lines of the round: the driver's command (full line with secondaries and CPU baseline), the NRMS model as the main line, the timing of the
# dropout-correction kernel's final form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
S=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
echo "driver command took $(( $(date +%s) - S )) s"
timeout 900 python bench.py --model nrms --steps 200 --warmup 20 --no-secondary > gpurun_out/r05/nrms_bench.json 2> gpurun_out/r05/nrms_bench.err
timeout 300 python tools/dropcorr_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/dropcorr_time.txt
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05/bench_n1.json') if l.startswith('{')][0])
print('naml', d['ms_per_step'], d['value'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), 'step', d['roofline_step']['frac'], 'cpu', d.get('cpu_baseline', {}) and d['cpu_baseline'].get('value'))
for k, v in (d.get('secondary') or {}).items():
    print('  ', k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'frac', 'achieved', 'traffic')} if isinstance(v, dict) else v)
d = json.loads([l for l in open('gpurun_out/r05/nrms_bench.json') if l.startswith('{')][0])
print('nrms', d['ms_per_step'], d['value'])
PY
