#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/prof2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/naml -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/naml.log 2>&1
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/naml_ser -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/naml_ser.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nrms -o p -- python3 $GRAFT_REPO_ROOT/bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/nrms.log 2>&1
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nrms_ser -o p -- python3 $GRAFT_REPO_ROOT/bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/nrms_ser.log 2>&1
cd $GRAFT_REPO_ROOT
for d in naml naml_ser nrms nrms_ser; do cp $O/$d/p_kernel_stats.csv $O/${d}_kernel_stats.csv; grep '^{' $O/$d.log > $O/${d}_benchline.json; rm -rf $O/$d; done
python bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/nrms_bench.json
python bench.py --model nrms --embed null --steps 100 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/nrms_null_bench.json
python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{' > $O/naml_400.json
ls -la $O
