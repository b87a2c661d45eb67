#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for m in 0 1; do LEGO_ROWS2=$m timeout 300 python tools/rows2_check.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/rows2_check.txt
