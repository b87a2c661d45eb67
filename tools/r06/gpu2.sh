#!/bin/bash
# round 6, call 2: the whole GPU suite after the prune (no -x), then two short bench lines (NAML, NRMS)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
S=$(date +%s)
timeout 1800 python -m pytest tests/ -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r06/pytest_gpu_2.txt
echo "gpu suite took $(( $(date +%s) - S )) s" | tee -a gpurun_out/r06/pytest_gpu_2.txt
for m in naml nrms; do
timeout 600 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2> gpurun_out/r06/b2_$m.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'], 'host', d['host_enqueue_ms_per_step'], 'step', d['roofline_step'] and d['roofline_step']['frac'], d['fracs_over_one'])
print({k: round(v['avg_ms'] * 1e3, 1) for k, v in d['kernels'].items()})"
done
