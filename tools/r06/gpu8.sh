#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "per_key or nrms or dropout_correction" 2>&1 | tail -6 | tee gpurun_out/r06/pytest_gpu_8.txt
python tools/dropcorr_time.py 2>&1 | tee gpurun_out/r06/dropcorr_time.txt
for i in 1 2; do
timeout 300 python bench.py --model nrms --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; print('nrms', d['ms_per_step'], d['value'], 'qkv_expand', round(k['qkv_expand_item']['avg_ms']*1e3,1), 'step frac', d['roofline_step']['frac'], d['fracs_over_one'])" | tee -a gpurun_out/r06/dropcorr_time.txt
done
# gather policy in the dense / de-duplication-off step
for nt in 2 0; do for u in 4 2; do
LEGO_DEDUP=0 LEGO_GATHER_NT=$nt LEGO_GATHER_U=$u timeout 300 python - <<PY 2>/dev/null | tee -a gpurun_out/r06/gather_in_step.txt
import sys, os, json, torch
sys.path.insert(0, '.')
import bench
from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_naml_params, make_world
from legommenders_amd.train_step import DeviceData, TrainStep
dev = torch.device('cuda:0')
cfg = dict(MIND_SMALL); world = make_world(seed=2023, **cfg)
dd = DeviceData(bench.dense_world(world), dev, seed=2023)
glove = glove_like(cfg['V'], 300, seed=2024, device=dev)
ts = TrainStep('naml', init_naml_params(D=256, V=cfg['V'], n_cat=cfg['n_cat'], glove=glove), dd, 64, K=4, lr=1e-3, seed=2023, dropout=True, tail='drop')
bar = torch.cuda.synchronize
d, tm, _ = bench.timed_steps(ts, 40, 10, bar, 1, tags={'gather_rows_in_step'})
g = bench.kernel_table(tm)['gather_rows_in_step']
rows = ts.counter_sum.tolist()[0] / 40
print('dense dedup-off NT=$nt U=$u: step', round(d / 40 * 1e3, 4), 'ms, gather', round(g['avg_ms'] * 1e3, 1), 'us =', round(rows * 2404 / (g['avg_ms'] * 1e-3) / 1e12, 3), 'TB/s')
PY
done; done
