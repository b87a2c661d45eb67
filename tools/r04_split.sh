#!/bin/bash
# split-bf16 product mode (opt-in): tests, then step time of the three training configurations in both modes
mkdir -p gpurun_out/split
python -m pytest tests/test_split_bf16.py -q -m gpu -x -s 2>&1 | grep -v amdgpu.ids | tail -15
F="--steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-bert --no-dist-check"
for cfg in "naml glove" "nrms glove" "nrms null"; do
  set -- $cfg
  for mode in 0 1; do
    LEGO_SPLIT_BF16=$mode python bench.py --model $1 --embed $2 $F > gpurun_out/split/$1_$2_$mode.json 2> gpurun_out/split/$1_$2_$mode.err
  done
done
LEGO_WINO=0 python bench.py --model naml $F > gpurun_out/split/naml_glove_direct.json 2> gpurun_out/split/naml_glove_direct.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/split/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j.get("final_loss"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
