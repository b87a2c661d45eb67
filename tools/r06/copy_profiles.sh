#!/bin/bash
# the closing run's artefacts (gpurun_out/ is scratch) into profiles/r06_*
cd "$(dirname "$0")/../.."
g=gpurun_out
pick() { grep '^{' "$1" | head -1 > "$2"; }
pick $g/r06/bench_n1.json profiles/r06_bench_n1.json
pick $g/r06/nrms_bench.json profiles/r06_nrms_bench.json
pick $g/r06/bench_200.json profiles/r06_bench_200steps.json
cp $g/r06/pytest_gpu_final.txt profiles/r06_pytest_gpu_final.txt
cp $g/r06/train_band_report.json profiles/r06_train_band.json
cp $g/prof_r06/kernel_stats.csv profiles/r06_bench_n1_kernel_stats.csv
cp $g/prof_r06/kernel_stats_serial.csv profiles/r06_bench_n1_kernel_stats_serial.csv
cp $g/prof_r06/bench_under_rocprof.json profiles/r06_bench_n1_under_rocprof.json
cp $g/prof_r06/gather_under_pmc.json profiles/r06_gather_hbm_under_pmc.json
cp $g/prof_r06/pmc_issue.json profiles/r06_pmc_issue.json
cp $g/prof_r06/timeline.txt profiles/r06_timeline.txt
cp $g/prof_r06/traffic.json profiles/r06_traffic.json
cp $g/prof_r06_nrms/kernel_stats.csv profiles/r06_nrms_kernel_stats.csv
cp $g/prof_r06_nrms/kernel_stats_serial.csv profiles/r06_nrms_kernel_stats_serial.csv
cp $g/prof_r06_nrms/pmc_issue.json profiles/r06_pmc_issue_nrms.json
cp $g/prof_r06_nrms/timeline.txt profiles/r06_nrms_timeline.txt
cp $g/prof_r06_nrms/traffic.json profiles/r06_traffic_nrms.json
git status --short profiles | head -30
