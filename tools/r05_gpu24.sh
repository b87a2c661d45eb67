#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/train_band_report.json
S=$(date +%s)
LEGO_BAND_REPORT=gpurun_out/r05/train_band_report.json timeout 3000 python -m pytest tests/test_train_band.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -12 | tee gpurun_out/r05/pytest_band.txt
echo "band tests took $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05/train_band_report.json'))
for k, v in d.items():
    print(k, v['seeds'], {m: (x['mi355x_mean'], x['reference_mean'], x['abs_diff'], x['tolerance']) for m, x in v['metrics'].items() if m == 'GAUC'})
PY
