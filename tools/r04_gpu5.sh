#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_5; mkdir -p $O
timeout 900 python -m pytest tests/test_bert_operator.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
for blas in 0 1; do
LEGO_BERT_BLAS=$blas timeout 600 python tools/bert_naml_bench.py --tune_from 0 --steps 4 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/blas=$blas: /"
done 2>&1 | tee $O/bert_bench.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/bert_naml_bench.py --tune_from 0 --steps 3 --warmup 1 > $O/prof.log 2>&1
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/bert_kernel_stats.csv; rm -rf $O/prof; head -30 $O/bert_kernel_stats.csv | cut -c1-200
