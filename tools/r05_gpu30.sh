#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-secondary > gpurun_out/r05/bench_forcedist_world1.json 2> gpurun_out/r05/bench_forcedist_world1.err
tail -3 gpurun_out/r05/bench_forcedist_world1.err | cut -c1-300
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r05/bench_forcedist_world1.json') if l.startswith('{')][0])
print({k: d.get(k) for k in ('ms_per_step', 'value', 'n_gpus', 'ranks_seen', 'allreduce_ms', 'prewarm_scratch_steps')}, d.get('config', {}).get('parallelism'))
PY
LEGO_BERT_STEP_TIMES=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r05/bench_dbg.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bert', {k: v.get('value') for k, v in d['secondary']['bert_naml_base'].items() if isinstance(v, dict)}, d['secondary']['split_bf16_opt_in']['bert_naml_base_tune_from_0']['value'])"
