#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/unpack; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_custom_ops.py -m gpu -q -x -k "wino or conv or slab or naml or engine or golden or custom or opcheck" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('naml', d['ms_per_step'], d['value'])"; done
cd /tmp; export TMPDIR=/tmp
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ser -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/ser.log 2>&1
grep -E "unpack|Name" $O/ser/p_kernel_stats.csv | cut -c1-200
rm -rf $O/ser
