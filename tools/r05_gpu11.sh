#!/bin/bash
cd $GRAFT_REPO_ROOT
./tools/prof_r05.sh gpurun_out/prof_r05 naml > gpurun_out/prof_r05_naml.log 2>&1
./tools/prof_r05.sh gpurun_out/prof_r05_nrms nrms > gpurun_out/prof_r05_nrms.log 2>&1
tail -5 gpurun_out/prof_r05_naml.log; tail -5 gpurun_out/prof_r05_nrms.log
