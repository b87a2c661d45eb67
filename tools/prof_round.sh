#!/bin/bash
# kernel trace (overlapped + serial), PMC passes (FETCH_SIZE / WRITE_SIZE / issue counters) of the NAML bench and of the HBM-sized
# row gather, timeline of one step, the plain bench line.   tools/prof_round.sh [outdir] [model]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export BENCH_SPIN_MS=0 BENCH_PREWARM_STEPS=0      # (profiling passes: no wake-up loop, no scratch instance -- the kernel statistics hold the measured steps only)
O=${1:-gpurun_out/prof_r06}; M=${2:-naml}; rm -rf $O; mkdir -p $O
B="python3 bench.py --model $M --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-dist-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- $B > $O/bench_serial_under_rocprof.json 2> $O/stats_serial.err
S="python3 bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-dist-check"
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $S > /dev/null 2> $O/pmc_fetch.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $S > /dev/null 2> $O/pmc_write.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_issue -- $S > /dev/null 2> $O/pmc_issue.err
FF=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); WW=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
if [ "$M" = naml ]; then
  G="python3 tools/gather_hbm.py 105600 8 --uniform-only"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gfetch -- $G > $O/gather_under_pmc.json 2> $O/pmc_gfetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_gwrite -- $G > /dev/null 2> $O/pmc_gwrite.err
  FF=$FF,gather_rows_hbm=$(ls $O/pmc_gfetch/*/*counter_collection.csv | head -1); WW=$WW,gather_rows_hbm=$(ls $O/pmc_gwrite/*/*counter_collection.csv | head -1)
fi
f=$(ls $O/stats/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f > $O/timeline.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/stats_serial/*/*kernel_stats.csv | head -1) $O/kernel_stats_serial.csv
python3 tools/traffic_from_pmc.py $FF $WW $O/traffic.json > $O/traffic.log 2>&1
python3 tools/pmc_summary.py $O/pmc_issue.json $(ls $O/pmc_issue/*/*counter_collection.csv | head -1) > $O/pmc_issue.log 2>&1
rm -rf $O/stats $O/stats_serial $O/pmc_fetch $O/pmc_write $O/pmc_issue $O/pmc_gfetch $O/pmc_gwrite
ls -la $O; tail -30 $O/traffic.log
