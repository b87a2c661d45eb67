#!/bin/bash
# round 6, call 4: attention core with the 16 x 16 tile path: its tests, the probe against the previous kernels on the same box, NRMS bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "mhsa or mha or nrms or attention or fold" 2>&1 | tail -8 | tee gpurun_out/r06/pytest_gpu_4.txt
timeout 600 python -m pytest tests/test_bert_operator.py -q -m gpu -x 2>&1 | tail -4 | tee -a gpurun_out/r06/pytest_gpu_4.txt
python tools/mhsa_probe.py 2>&1 | tee gpurun_out/r06/mhsa_probe.txt
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/liblego_hip_oldmhsa.so python tools/mhsa_probe.py 2>&1 | tee -a gpurun_out/r06/mhsa_probe.txt
for i in 1 2; do for lib in liblego_hip.so liblego_hip_oldmhsa.so; do
LEGO_HIP_LIB=$PWD/legommenders_amd/csrc/$lib timeout 600 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('$lib', d['ms_per_step'], d['value'], 'fwd', round(k['mhsa_core_fwd_item']['avg_ms']*1e3,1), 'bwd', round(k['mhsa_core_bwd_item']['avg_ms']*1e3,1))" | tee -a gpurun_out/r06/mhsa_probe.txt
done; done
