#!/bin/bash
# config 5: wide weight gradients as 768-wide chunks (tnd_kernel's window) against one launch each, same box; then the BERT tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for chunk in 768 0 768 0; do
timeout 600 python - <<PY 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a gpurun_out/r06/bert_wgrad_chunks.txt
import sys
sys.path.insert(0, 'tools')
import bert_naml_bench
from legommenders_amd import bert_native
bert_native.WGRAD_CHUNK = $chunk
r = bert_naml_bench.run(batch=64, steps=10, warmup=5, layers=12, hidden=256, tune_from=0)
print('wgrad_chunk', $chunk, r['impressions_per_s'], 'us/row', r['us_per_live_row'], 'loss', round(r['loss'], 4))
print({k: (v['ms_per_step'], v['tflops']) for k, v in (r['kernels'] or {}).items() if 'weight' in k})
PY
done
timeout 900 python -m pytest tests/test_bert_operator.py tests/test_split_bf16.py -q -m gpu 2>&1 | tail -4 | tee -a gpurun_out/r06/bert_wgrad_chunks.txt
