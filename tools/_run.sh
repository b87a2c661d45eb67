#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/fold2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep FAILED $O/pytest.log
