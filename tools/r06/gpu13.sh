#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_bert_operator.py tests/test_arena.py tests/test_trainer_cli.py tests/test_plugin_api.py -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r06/pytest_gpu_13.txt
