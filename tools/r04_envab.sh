#!/bin/bash
# A/B of runtime environment settings that change launch latency (read by the HIP runtime at initialisation)
P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j["value"], j["ms_per_step"], j["host_enqueue_ms_per_step"])'
for m in naml nrms; do
F="--model $m --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-bert --no-dist-check"
for rep in 1 2; do
echo -n "$m base "; python bench.py $F 2>/dev/null | python3 -c "$P"
echo -n "$m HIP_FORCE_DEV_KERNARG=1 "; HIP_FORCE_DEV_KERNARG=1 python bench.py $F 2>/dev/null | python3 -c "$P"
echo -n "$m HIP_FORCE_DEV_KERNARG=0 "; HIP_FORCE_DEV_KERNARG=0 python bench.py $F 2>/dev/null | python3 -c "$P"
echo -n "$m HSA_ENABLE_INTERRUPT=0 "; HSA_ENABLE_INTERRUPT=0 python bench.py $F 2>/dev/null | python3 -c "$P"
done
done
