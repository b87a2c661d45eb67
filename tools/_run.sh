#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final5; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
grep '^{' $O/bench.log > $O/bench.json; tail -3 $O/bench.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/final5/bench.json'))
print(d['value'], d['ms_per_step'], d['long_run'])
for k,v in d['secondary'].items(): print(k, v['ms_per_step'], v['value'])
PY
timeout 900 python -m pytest tests/test_dp_device.py tests/test_trainer_cli.py -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
