#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/train_band_report.json
S=$(date +%s)
LEGO_BAND_REPORT=gpurun_out/r05/train_band_report.json timeout 3400 python -m pytest tests/ -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error|error|assert" | tail -12 | tee gpurun_out/r05/pytest_gpu_final.txt
echo "gpu suite took $(( $(date +%s) - S )) s" | tee -a gpurun_out/r05/pytest_gpu_final.txt
