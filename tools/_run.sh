#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep FAILED $O/pytest.log
timeout 300 python bench.py --model nrms --embed null --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/nrms_null.json
timeout 300 python bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/nrms.json
python -c "
import json
for f in ('nrms','nrms_null'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
