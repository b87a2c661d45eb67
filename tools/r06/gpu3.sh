#!/bin/bash
# round 6, call 3: config 5 on the workspace arena + flat Adam + GELU epilogues: its tests, then the stand-alone bench (6 warm-up, 12 timed steps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_bert_operator.py tests/test_trainer_cli.py tests/test_plugin_api.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -25 | tee gpurun_out/r06/pytest_gpu_3.txt
for fused in 1 0; do
LEGO_BERT_STEP_TIMES=1 timeout 600 python - <<PY 2>&1 | tail -12 | tee -a gpurun_out/r06/bert_3.txt
import sys, json
sys.path.insert(0, 'tools')
import bert_naml_bench
from legommenders_amd import bert_native
bert_native.FUSED_GELU = bool($fused)
r = bert_naml_bench.run(batch=64, steps=12, warmup=6, layers=12, hidden=256, tune_from=0)
print('fused_gelu', $fused, {k: r[k] for k in ('impressions_per_s', 'step_ms', 'step_ms_max_over_min', 'workspace_arena', 'torch_allocator', 'loss')})
print({k: (v['ms_per_step'], v['tflops']) for k, v in (r['kernels'] or {}).items()})
PY
done
