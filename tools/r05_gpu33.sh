#!/bin/bash
# the round's closing run: the whole GPU suite (train bands with their report), then the bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/train_band_report.json
S=$(date +%s)
LEGO_BAND_REPORT=gpurun_out/r05/train_band_report.json timeout 3400 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -12 | tee gpurun_out/r05/pytest_gpu_final.txt
echo "gpu suite took $(( $(date +%s) - S )) s" | tee -a gpurun_out/r05/pytest_gpu_final.txt
S=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
echo "driver command took $(( $(date +%s) - S )) s"
timeout 900 python bench.py --model nrms --steps 200 --warmup 20 --no-secondary > gpurun_out/r05/nrms_bench.json 2> gpurun_out/r05/nrms_bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05/train_band_report.json'))
for k, v in d.items():
    print(k, v['seeds'], {m: (x['mi355x_mean'], x['reference_mean'], x['abs_diff'], x['tolerance']) for m, x in v['metrics'].items() if m == 'GAUC'})
d = json.loads([l for l in open('gpurun_out/r05/bench_n1.json') if l.startswith('{')][0])
print('naml', d['ms_per_step'], d['value'], 'long', d['long_run']['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline']['traffic_source']['stale'], 'step', d['roofline_step']['frac'], 'cpu', d['cpu_baseline']['value'])
for k, v in (d.get('secondary') or {}).items():
    print('  ', k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'traffic')} if isinstance(v, dict) else v)
print('   bert', {k: (v.get('value'), v.get('step_ms')) for k, v in d['secondary']['bert_naml_base'].items() if isinstance(v, dict)})
print('   split', {k: v.get('value') for k, v in d['secondary']['split_bf16_opt_in'].items() if isinstance(v, dict)})
print('   gather', d['secondary']['gather_rows_hbm_bound']['alone_cold_cache']['frac'])
d = json.loads([l for l in open('gpurun_out/r05/nrms_bench.json') if l.startswith('{')][0])
print('nrms', d['ms_per_step'], d['value'])
PY
