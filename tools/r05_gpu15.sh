#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "in_projection_per_key or nrms_projection_once" 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -8
timeout 900 python tools/nrms_dropcorr_trajectory.py 600 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/nrms_dropcorr_trajectory.txt | tail -12
