#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in "LEGO_TND=1" "LEGO_TND=0" "LEGO_TND=1" "LEGO_TND=0"; do echo "== $v"; env $v timeout 600 python tools/bert_naml_bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400; done | tee gpurun_out/r05/bert_tnd_ab.txt
