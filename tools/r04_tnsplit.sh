#!/bin/bash
# step time against the k split of the small-output weight-gradient (TN) launches (LEGO_TN_SPLIT; 0 = the built-in rule)
P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["value"], j["ms_per_step"])'
for m in nrms naml; do
F="--model $m --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-bert --no-dist-check"
for rep in 1 2; do
for s in 0 96 128 192 256; do echo -n "$m TN_SPLIT=$s "; LEGO_TN_SPLIT=$s python bench.py $F 2>/dev/null | python3 -c "$P"; done
done
done
