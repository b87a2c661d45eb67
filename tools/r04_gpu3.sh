#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_3; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/r04_3/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("long_run"), d.get("dist_path_check"))
print(d["roofline_gather"]); print(d["cpu_baseline"])
for k,s in d["secondary"].items(): print(k, s.get("value"), s.get("ms_per_step"), s.get("roofline_gather"))
P
