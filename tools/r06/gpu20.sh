#!/bin/bash
# NRMS per-key in-projection: dropped coordinates as (offset, multiplier) pairs written by a launch of their own and read through the scalar unit
# (s_load_dwordx16 = 8 pairs) against the in-kernel list (LEGO_DROPCORR_PAIRS=0), same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/dropcorr_scalar_pairs.txt; rm -f $O
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "dropcorr or per_key or nrms" 2>&1 | tail -3 | tee -a $O
for v in 1 0; do
echo "LEGO_DROPCORR_PAIRS=$v" | tee -a $O
LEGO_DROPCORR_PAIRS=$v python tools/dropcorr_time.py 2>&1 | grep -v amdgpu.ids | head -5 | tee -a $O
done
for i in 1 2; do for v in 1 0; do
LEGO_DROPCORR_PAIRS=$v timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('pairs=$v nrms', d['ms_per_step'], d['value'], 'qkv_expand', round(k['qkv_expand_item']['avg_ms']*1e3,1), 'loss', d['final_loss'])" | tee -a $O
done; done
