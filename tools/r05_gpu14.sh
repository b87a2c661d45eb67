#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python tools/nrms_dropcorr_trajectory.py 600 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/nrms_dropcorr_trajectory.txt
