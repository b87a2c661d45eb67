#!/bin/bash
# round 6, call 1: the GPU suite on the round's first tree (new headline NRMS-GloVe tests, bench line changes) + the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
S=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/r06/pytest_gpu_1.txt
echo "gpu suite took $(( $(date +%s) - S )) s" | tee -a gpurun_out/r06/pytest_gpu_1.txt
S=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_1.json 2> gpurun_out/r06/bench_1.err
echo "driver command took $(( $(date +%s) - S )) s rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_1.json') if l.startswith('{')][0])
print('naml', d['ms_per_step'], d['value'], 'cold', d.get('value_without_prewarm'), 'long', d['long_run']['ms_per_step'], 'roofline', d['roofline']['frac'], 'step', d['roofline_step']['frac'], 'fracs>1', d['fracs_over_one'])
for k, v in (d.get('secondary') or {}).items():
    print('  ', k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'error')} if isinstance(v, dict) else v)
b = d['secondary'].get('bert_naml_base', {})
print('   bert', {k: (v.get('value'), v.get('step_ms')) for k, v in b.items() if isinstance(v, dict)})
print('   dist', d.get('dist_path_check'))
PY
