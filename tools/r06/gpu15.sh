#!/bin/bash
# Philox products as v_mad_u64_u32 (one instruction per 32 x 32 -> 64 product) against v_mul_hi_u32 + v_mul_lo_u32: the instruction rates, then the
# library built either way (liblego_hip_mul32.so = the build before), same box: attention core alone, NRMS step, NAML step; dropout parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/philox_mad64.txt; rm -f $O
./tools/bin/philox_rate 2>&1 | tee -a $O
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "dropout or mhsa or mask or switches" 2>&1 | tail -3 | tee -a $O
for lib in liblego_hip.so liblego_hip_mul32.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib python tools/mhsa_probe.py 2>&1 | grep "nrms item side" | tee -a $O
done
for i in 1 2; do for lib in liblego_hip.so liblego_hip_mul32.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('$lib nrms', d['ms_per_step'], d['value'], 'fwd', round(k['mhsa_core_fwd_item']['avg_ms']*1e3,1), 'bwd', round(k['mhsa_core_bwd_item']['avg_ms']*1e3,1), 'user fwd/bwd', round(k['mhsa_core_fwd_user']['avg_ms']*1e3,1), round(k['mhsa_core_bwd_user']['avg_ms']*1e3,1))" | tee -a $O
done; done
for i in 1 2; do for lib in liblego_hip.so liblego_hip_mul32.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib naml', d['ms_per_step'], d['value'])" | tee -a $O
done; done
