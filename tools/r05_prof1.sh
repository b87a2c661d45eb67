#!/bin/bash
# kernel trace of the NAML bench on the current tree: per-queue timeline of one step + kernel statistics (overlapped and serial)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/prof1; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-dist-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- $B > $O/bench_serial_under_rocprof.json 2> $O/stats_serial.err
f=$(ls $O/stats/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f > $O/timeline.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/stats_serial/*/*kernel_stats.csv | head -1) $O/kernel_stats_serial.csv
rm -rf $O/stats $O/stats_serial
cat $O/timeline.txt
