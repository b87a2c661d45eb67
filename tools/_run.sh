cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r2_t19.log
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_xcd -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r2_prof_xcd.log 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/r2_b15_naml.json 2>/dev/null
python bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r2_b15_nrms.json 2>/dev/null
