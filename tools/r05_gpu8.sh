#!/bin/bash
mkdir -p gpurun_out/r05
timeout 600 python tools/dropcorr_debug.py > gpurun_out/r05/dropcorr_debug.txt 2>&1
tail -40 gpurun_out/r05/dropcorr_debug.txt
