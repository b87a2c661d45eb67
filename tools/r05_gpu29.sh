#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
LEGO_BERT_STEP_TIMES=1 timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_dbg.json 2> gpurun_out/r05/bench_dbg.err
grep "step times" gpurun_out/r05/bench_dbg.err
LEGO_BERT_STEP_TIMES=1 timeout 600 python tools/bert_naml_bench.py --steps 5 --warmup 2 2>&1 | grep "step times"
