#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for m in 0 1; do LEGO_TNDP=$m timeout 300 python tools/wino2_check.py > gpurun_out/r05/tndp_check_mode$m.txt 2>&1; echo "tndp $m rc $?"; grep -c "^ok" gpurun_out/r05/tndp_check_mode$m.txt; grep "FAIL\|Error\|error" gpurun_out/r05/tndp_check_mode$m.txt | head -5; tail -3 gpurun_out/r05/tndp_check_mode$m.txt | sed 's/.*wgrad direct/wgrad direct/'; done
run() { timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-dist-check 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], d['value'], {t: round(k[t]['avg_ms']*1e3,1) for t in ('additive_bwd_weight_item','proj_bwd_weight','conv3_bwd_weight','conv3_bwd_data') if t in k}, d['final_loss'])"; }
for i in 1 2; do
LEGO_TNDP=0 run "tndp=0"
LEGO_TNDP=1 run "tndp=1"
LEGO_TNDP=1 LEGO_TND=1 run "tndp=1 tnd=1"
done | tee gpurun_out/r05/tndp_step_ab.txt
