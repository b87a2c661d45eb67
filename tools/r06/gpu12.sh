#!/bin/bash
# attention core at 5 / 6 waves per SIMD (forced register budgets: 5 / 17 spilled registers forward, 37 backward) against the shipped 4
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for lib in liblego_hip.so liblego_hip_occ5.so liblego_hip_occ6.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib python tools/mhsa_probe.py 2>&1 | grep "nrms item side" | tee -a gpurun_out/r06/mhsa_occ.txt
done
for i in 1 2; do for lib in liblego_hip.so liblego_hip_occ5.so liblego_hip_occ6.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('$lib', d['ms_per_step'], d['value'], 'fwd', round(k['mhsa_core_fwd_item']['avg_ms']*1e3,1), 'bwd', round(k['mhsa_core_bwd_item']['avg_ms']*1e3,1))" | tee -a gpurun_out/r06/mhsa_occ.txt
done; done
