#!/bin/bash
# attention core with ONE buffer descriptor per operand tile and a running per-lane row offset (no per-row descriptors: the scalar registers
# no longer spill into vector lanes) against the build before it, same box: parity tests, the core alone, the NRMS step, config 5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/mhsa_one_descriptor.txt; rm -f $O
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "mhsa or nrms or attention" 2>&1 | tail -3 | tee -a $O
for lib in liblego_hip.so liblego_hip_oldmhsa.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib python tools/mhsa_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O
done
for i in 1 2; do for lib in liblego_hip.so liblego_hip_oldmhsa.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('$lib', d['ms_per_step'], d['value'], 'fwd', round(k['mhsa_core_fwd_item']['avg_ms']*1e3,1), 'bwd', round(k['mhsa_core_bwd_item']['avg_ms']*1e3,1), 'user fwd/bwd', round(k['mhsa_core_fwd_user']['avg_ms']*1e3,1), round(k['mhsa_core_bwd_user']['avg_ms']*1e3,1))" | tee -a $O
done; done
for lib in liblego_hip.so liblego_hip_oldmhsa.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 600 python tools/bert_naml_bench.py --steps 8 --warmup 4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-700 | tee -a $O
done
