#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_6; mkdir -p $O
timeout 1500 python -m pytest tests/test_bert_operator.py tests/test_plugin_api.py tests/test_trainer_cli.py tests/test_dp_device.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
for one in 1 0 1 0; do
LEGO_ONE_ITEM_CALL=$one timeout 600 python tools/bert_naml_bench.py --tune_from 0 --steps 4 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-170 | sed "s/^/one_call=$one: /"
done 2>&1 | tee $O/bert_bench.txt
