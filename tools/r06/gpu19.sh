#!/bin/bash
# wave index made uniform (readfirstlane) in the pool / user-tower / column-sum / LayerNorm kernels: row indices and addresses become scalar
# (additive_pool_bwd_fast 131 -> 106 VGPRs = 4 waves per SIMD, ln_bwd<3> 181 -> 164 = 3 waves) against the build before, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/uniform_wave.txt; rm -f $O
timeout 1200 python -m pytest tests/test_hip_parity.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tee -a $O
timeout 900 python -m pytest tests/test_bert_operator.py -q -m gpu 2>&1 | grep -E "passed|failed" | tee -a $O
for i in 1 2 3; do for lib in liblego_hip.so liblego_hip_prev.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib naml', d['ms_per_step'], d['value'])" | tee -a $O
done; done
for i in 1 2; do for lib in liblego_hip.so liblego_hip_prev.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib nrms', d['ms_per_step'], d['value'])" | tee -a $O
done; done
for lib in liblego_hip.so liblego_hip_prev.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 600 python tools/bert_naml_bench.py --steps 8 --warmup 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260 | tee -a $O
done
