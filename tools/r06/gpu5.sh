#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_bert_operator.py -q -m gpu -x -k "arena" 2>&1 | grep -v "^E   *[+|]" | tail -60 > gpurun_out/r06/pytest_gpu_5.txt
tail -5 gpurun_out/r06/pytest_gpu_5.txt
