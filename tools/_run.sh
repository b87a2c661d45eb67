#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2>&1
tail -4 $O/bench.log
