#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "in_projection_per_key or nrms_projection_once or trajectory_per_key or gather" 2>&1 | grep -E "passed|failed|FAILED|Error|error" | tail -4
./tools/prof_r05.sh gpurun_out/prof_r05 naml > gpurun_out/prof_r05_naml.log 2>&1
./tools/prof_r05.sh gpurun_out/prof_r05_nrms nrms > gpurun_out/prof_r05_nrms.log 2>&1
tail -14 gpurun_out/prof_r05_naml.log
