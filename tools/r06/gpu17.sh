#!/bin/bash
# attention core: segments dealt to XCDs with all their heads (+ one branch around the probability stores) against the build before (liblego_hip_prev.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/mhsa_xcd_deal.txt; rm -f $O
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "mhsa or nrms or attention" 2>&1 | tail -3 | tee -a $O
timeout 900 python -m pytest tests/test_bert_operator.py tests/test_plugin_api.py -q -m gpu 2>&1 | tail -3 | tee -a $O
for lib in liblego_hip.so liblego_hip_prev.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib python tools/mhsa_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O
done
for i in 1 2; do for lib in liblego_hip.so liblego_hip_prev.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('$lib', d['ms_per_step'], d['value'], 'fwd', round(k['mhsa_core_fwd_item']['avg_ms']*1e3,1), 'bwd', round(k['mhsa_core_bwd_item']['avg_ms']*1e3,1), 'user fwd/bwd', round(k['mhsa_core_fwd_user']['avg_ms']*1e3,1), round(k['mhsa_core_bwd_user']['avg_ms']*1e3,1))" | tee -a $O
done; done
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/liblego_hip.so python tools/mhsa_scaling.py 2>&1 | grep -v amdgpu.ids | tee -a $O
