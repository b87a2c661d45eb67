#!/bin/bash
# NAML: the side stream's large weight-gradient product (additive hidden layer, item side) held back until the conv weight gradient is done, so
# that it runs beside the step's tail (per-token sums + the projection's weight gradient) instead of beside the two conv kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
O=gpurun_out/r06/naml_late_tn.txt; rm -f $O
for i in 1 2 3; do for v in 1 0; do
LEGO_X_LATE_TN=$v timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']; g = lambda t: round(1e3 * (k[t].get('every_kernel_bracketed') or k[t])['avg_ms'], 1)
print('late=$v naml', d['ms_per_step'], d['value'], 'bwd_data', g('conv3_bwd_data'), 'bwd_weight', g('conv3_bwd_weight'), 'big tn', g('additive_bwd_weight_item'), 'segsum', g('proj_bwd_segsum'), 'tn proj', g('proj_bwd_weight'))" | tee -a $O
done; done
