#!/bin/bash
# round 4 profiles: NAML (tools/prof_round.sh), then NRMS with the same passes (kernel stats overlapped / serial, FETCH / WRITE / issue PMC)
cd $GRAFT_REPO_ROOT
bash tools/prof_round.sh gpurun_out/prof_r04 > gpurun_out/prof_r04.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04_nrms; rm -rf $O; mkdir -p $O
B="python3 bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_under_rocprof.json 2> $O/stats.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- $B > $O/bench_serial_under_rocprof.json 2> $O/stats_serial.err
S="python3 bench.py --model nrms --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $S > /dev/null 2> $O/pmc_fetch.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $S > /dev/null 2> $O/pmc_write.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_issue -- $S > /dev/null 2> $O/pmc_issue.err
f=$(ls $O/stats/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f > $O/timeline.txt
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/stats_serial/*/*kernel_stats.csv | head -1) $O/kernel_stats_serial.csv
python3 tools/traffic_from_pmc.py $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) $O/traffic.json > $O/traffic.log 2>&1
python3 tools/pmc_summary.py $O/pmc_issue.json $(ls $O/pmc_issue/*/*counter_collection.csv | head -1) > $O/pmc_issue.log 2>&1
rm -rf $O/stats $O/stats_serial $O/pmc_fetch $O/pmc_write $O/pmc_issue
python3 bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
# NRMS with the trainable token table (per-key in-projection): kernel statistics, serial and overlapped, and the bench line
Q=gpurun_out/prof_r04_nrms_null; rm -rf $Q; mkdir -p $Q
BN="python3 bench.py --model nrms --embed null --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-dist-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $Q/stats -- $BN > $Q/bench_under_rocprof.json 2> $Q/stats.err
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $Q/stats_serial -- $BN > $Q/bench_serial_under_rocprof.json 2> $Q/stats_serial.err
f=$(ls $Q/stats/*/*kernel_trace.csv | head -1); python3 tools/timeline.py $f > $Q/timeline.txt
cp $(ls $Q/stats/*/*kernel_stats.csv | head -1) $Q/kernel_stats.csv
cp $(ls $Q/stats_serial/*/*kernel_stats.csv | head -1) $Q/kernel_stats_serial.csv
rm -rf $Q/stats $Q/stats_serial
$BN > $Q/bench.json 2> $Q/bench.err
# BERT-base NAML (config 5): kernel statistics of the native blocks
rocprofv3 --kernel-trace --stats --output-format csv -d $Q/bert -- python3 tools/bert_naml_bench.py --tune_from 0 --steps 3 --warmup 1 > $Q/bert_bench.txt 2>&1
cp $(ls $Q/bert/*/*kernel_stats.csv | head -1) $Q/bert_kernel_stats.csv; rm -rf $Q/bert
ls -la gpurun_out/prof_r04 $O $Q; tail -3 gpurun_out/prof_r04.log
