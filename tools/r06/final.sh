#!/bin/bash
# the round's closing sequence on the final tree: GPU suite (train bands with their report), profiles (NAML + NRMS), then -- with the
# traffic summary in place -- the bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/train_band_report.json
S=$(date +%s)
LEGO_BAND_REPORT=gpurun_out/r06/train_band_report.json timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -12 | tee gpurun_out/r06/pytest_gpu_final.txt
echo "gpu suite took $(( $(date +%s) - S )) s" | tee -a gpurun_out/r06/pytest_gpu_final.txt
./tools/prof_round.sh gpurun_out/prof_r06 naml > gpurun_out/prof_r06_naml.log 2>&1
./tools/prof_round.sh gpurun_out/prof_r06_nrms nrms > gpurun_out/prof_r06_nrms.log 2>&1
cp gpurun_out/prof_r06/traffic.json profiles/r06_traffic.json
S=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_n1.json 2> gpurun_out/r06/bench_n1.err
echo "driver command took $(( $(date +%s) - S )) s"
timeout 900 python bench.py --model nrms --steps 200 --warmup 20 --no-secondary > gpurun_out/r06/nrms_bench.json 2> gpurun_out/r06/nrms_bench.err
timeout 600 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/r06/bench_200.json 2> gpurun_out/r06/bench_200.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_n1.json') if l.startswith('{')][0])
print('naml', d['ms_per_step'], d['value'], 'cold', d.get('value_without_prewarm'), 'long', d['long_run']['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline']['traffic_source']['stale'], 'step', d['roofline_step']['frac'], d['fracs_over_one'])
for k, v in (d.get('secondary') or {}).items():
    print('  ', k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'traffic', 'error')} if isinstance(v, dict) else v)
print('   bert', {k: (v.get('value'), v.get('us_per_live_row_max_over_min')) for k, v in d['secondary']['bert_naml_base'].items() if isinstance(v, dict)})
print('   gather', {k: (v.get('frac') if isinstance(v, dict) else v) for k, v in d['secondary']['gather_rows_hbm_bound'].items()})
print('   dist', d.get('dist_path_check'))
d = json.loads([l for l in open('gpurun_out/r06/nrms_bench.json') if l.startswith('{')][0])
print('nrms', d['ms_per_step'], d['value'], d['roofline_step'], d['fracs_over_one'])
d = json.loads([l for l in open('gpurun_out/r06/bench_200.json') if l.startswith('{')][0])
print('naml 200 steps', d['ms_per_step'], d['value'])
PY
