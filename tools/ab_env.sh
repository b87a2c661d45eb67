#!/bin/bash
# Same-box A/B of one environment switch on the headline step: tools/ab_env.sh VAR value_a value_b [repeats] [bench flags...]
# Prints ms/step and impressions/s of `bench.py --steps 200` alternately for the two values (boxes differ by 2-3 %: compare within a run only).
cd "$(dirname "$0")/.."
var=$1; a=$2; b=$3; reps=${4:-3}; shift 4 2>/dev/null
for i in $(seq "$reps"); do
  for v in "$a" "$b"; do
    printf '%s=%s  ' "$var" "$v"
    env "$var=$v" timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
