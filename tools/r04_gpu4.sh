#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_4; mkdir -p $O
timeout 1200 python -m pytest tests/test_bert_operator.py tests/test_abi.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
for nat in 1 0; do for tf in 0 9; do
LEGO_BERT_NATIVE=$nat timeout 600 python tools/bert_naml_bench.py --tune_from $tf --steps 4 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -2 | sed "s/^/native=$nat tune_from=$tf: /"
done; done 2>&1 | tee $O/bert_bench.txt
