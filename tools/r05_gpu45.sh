#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "weight_gradient or golden or nrms" 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -4
./tools/r05_gpu28.sh
