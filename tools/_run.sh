#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/nrms3; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_custom_ops.py -m gpu -q -x -k "mhsa or nrms or engine or custom or opcheck" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
LEGO_SERIAL=1 rocprofv3 --kernel-trace --output-format csv -d $O/ser -o p -- python3 $GRAFT_REPO_ROOT/bench.py --model nrms --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --time-every 1000 > $O/ser.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $O/ser/p_kernel_trace.csv > $O/ser_timeline.txt
rm -rf $O/ser
grep mhsa $O/ser_timeline.txt
timeout 300 python bench.py --model nrms --steps 60 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
