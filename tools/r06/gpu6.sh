#!/bin/bash
# round 6, call 6: NAML with the next step's prologue behind Adam on the main stream (no side chain / waits in the forward pass): parity tests,
# then A/B against the previous commit's engine on the same box (git stash is not available on the box: the old files ride along as *_old.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_dp_device.py tests/test_train_band.py tests/test_trainer_cli.py -q -m gpu -x -k "not bert" 2>&1 | tail -8 | tee gpurun_out/r06/pytest_gpu_6.txt
for i in 1 2 3; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['value'], 'host', d['host_enqueue_ms_per_step'])" | tee -a gpurun_out/r06/naml_pre_ab.txt
LEGO_X_OLD=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d['value'], 'host', d['host_enqueue_ms_per_step'])" | tee -a gpurun_out/r06/naml_pre_ab.txt
done
