#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_7; mkdir -p $O
timeout 900 python -m pytest tests/test_bert_operator.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do timeout 600 python tools/bert_naml_bench.py --tune_from 0 --steps 4 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170; done
