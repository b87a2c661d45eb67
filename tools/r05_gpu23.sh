#!/bin/bash
# ablation table of the two plain-row strip products of the NAML step (tuning library, LEGO_DMA_ABL bits) + the launch floor
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
T=$GRAFT_REPO_ROOT/legommenders_amd/csrc/liblego_hip_tune.so
{
echo "# tools/strip_ablation.py on the tuning library: LEGO_DMA_ABL bits 1 no DMA in the k loop, 4 no wait / barrier, 8 no epilogue, 32 no MFMAs"
for a in 0 1 4 8 32 9 41 45; do echo "ABL=$a: $(LEGO_HIP_LIB=$T LEGO_DMA_ABL=$a timeout 120 python tools/strip_ablation.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"; done
echo "# tools/launch_floor.py (product library)"
timeout 120 python tools/launch_floor.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r05/strip_ablation.txt
