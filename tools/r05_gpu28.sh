#!/bin/bash
# final passes of the round on the final kernel sources: profiles (NAML + NRMS), then -- with the traffic summary in place -- the bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
./tools/prof_r05.sh gpurun_out/prof_r05 naml > gpurun_out/prof_r05_naml.log 2>&1
./tools/prof_r05.sh gpurun_out/prof_r05_nrms nrms > gpurun_out/prof_r05_nrms.log 2>&1
cp gpurun_out/prof_r05/traffic.json profiles/r05_traffic.json
S=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
echo "driver command took $(( $(date +%s) - S )) s"
timeout 900 python bench.py --model nrms --steps 200 --warmup 20 --no-secondary > gpurun_out/r05/nrms_bench.json 2> gpurun_out/r05/nrms_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05/bench_n1.json') if l.startswith('{')][0])
print('naml', d['ms_per_step'], d['value'], 'long', d['long_run']['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline']['traffic_source']['stale'], 'step', d['roofline_step']['frac'])
for k, v in (d.get('secondary') or {}).items():
    print('  ', k, {kk: vv for kk, vv in v.items() if kk in ('value', 'ms_per_step', 'traffic')} if isinstance(v, dict) else v)
print('   bert', {k: v.get('value') for k, v in d['secondary']['bert_naml_base'].items() if isinstance(v, dict)})
d = json.loads([l for l in open('gpurun_out/r05/nrms_bench.json') if l.startswith('{')][0])
print('nrms', d['ms_per_step'], d['value'])
PY
