#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 300 python tools/dropcorr_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/dropcorr_time.txt
