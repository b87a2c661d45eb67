#!/bin/bash
# kernel-trace timeline of one NRMS (GloVe) step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/nrms_tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-bert --no-dist-check > $O/bench.json 2> $O/err
f=$(ls $O/t/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f > $O/timeline.txt; rm -rf $O/t
sed -n 45,80p $O/timeline.txt | cut -c1-105
