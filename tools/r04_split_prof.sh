#!/bin/bash
# timeline + kernel statistics of the NAML step in the split-bf16 product mode
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/split_prof; rm -rf $O; mkdir -p $O
export LEGO_SPLIT_BF16=1
B="python3 bench.py --model naml --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-bert --no-dist-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.err
f=$(ls $O/stats/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f > $O/timeline.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- $B > /dev/null 2> $O/stats_serial.err
cp $(ls $O/stats_serial/*/*kernel_stats.csv | head -1) $O/kernel_stats_serial.csv
rm -rf $O/stats $O/stats_serial
cut -c1-120 $O/timeline.txt
