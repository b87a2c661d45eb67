#!/bin/bash
# the round's bench lines only (profiles are taken by r05_gpu28.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
S=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
echo "driver command took $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05/bench_n1.json') if l.startswith('{')][0])
print('naml', d['ms_per_step'], d['value'], 'long', d['long_run']['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['traffic_source']['stale'], 'step', d['roofline_step']['frac'], 'cpu', d['cpu_baseline']['value'])
for k, v in (d.get('secondary') or {}).items():
    print('  ', k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'traffic')} if isinstance(v, dict) else v)
print('   bert', {k: (v.get('value'), v.get('step_ms')) for k, v in d['secondary']['bert_naml_base'].items() if isinstance(v, dict)})
print('   split', {k: v.get('value') for k, v in d['secondary']['split_bf16_opt_in'].items() if isinstance(v, dict)})
PY
