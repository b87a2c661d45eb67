#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep FAILED $O/pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
tail -4 $O/bench.log | cut -c1-300
