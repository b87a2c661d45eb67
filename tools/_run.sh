cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "mhsa or nrms or eval_path" 2>&1 | tail -15 > gpurun_out/r2_t6.log
for m in 0 1 2; do
  LEGO_TN_MODE=$m LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tn$m -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r2_prof_tn$m.log 2>&1
  LEGO_TN_MODE=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/r2_b4_tn$m.json 2>/dev/null
done
python bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r2_b4_nrms.json 2>/dev/null
