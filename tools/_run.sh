#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/foldtest; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "folded or mask_dropout" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
