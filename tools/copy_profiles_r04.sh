#!/bin/bash
# gpurun_out/prof_r04* (tools/prof_r04.sh, tools/r04_split_prof.sh) -> profiles/r04_*
set -e
P=profiles; A=gpurun_out/prof_r04; B=gpurun_out/prof_r04_nrms; C=gpurun_out/prof_r04_nrms_null
cp $A/kernel_stats.csv $P/r04_bench_n1_kernel_stats.csv; cp $A/kernel_stats_serial.csv $P/r04_bench_n1_kernel_stats_serial.csv
cp $A/bench_under_rocprof.json $P/r04_bench_n1_under_rocprof.json; cp $A/traffic.json $P/r04_traffic.json
cp $A/pmc_issue.json $P/r04_pmc_issue.json; cp $A/timeline.txt $P/r04_timeline.txt
cp $B/kernel_stats.csv $P/r04_nrms_kernel_stats.csv; cp $B/kernel_stats_serial.csv $P/r04_nrms_kernel_stats_serial.csv
cp $B/traffic.json $P/r04_traffic_nrms.json; cp $B/pmc_issue.json $P/r04_pmc_issue_nrms.json; cp $B/timeline.txt $P/r04_nrms_timeline.txt
cp $C/bench.json $P/r04_nrms_null_bench.json; cp $C/kernel_stats.csv $P/r04_nrms_null_kernel_stats.csv
cp $C/kernel_stats_serial.csv $P/r04_nrms_null_kernel_stats_serial.csv; cp $C/timeline.txt $P/r04_nrms_null_timeline.txt
cp $C/bert_kernel_stats.csv $P/r04_bert_kernel_stats.csv; cp $C/bert_bench.txt $P/r04_bert_bench_under_rocprof.txt
cp gpurun_out/split_prof/timeline.txt $P/r04_split_bf16_naml_timeline.txt; cp gpurun_out/split_prof/kernel_stats.csv $P/r04_split_bf16_naml_kernel_stats.csv
