cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r2_t13.log
python bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r2_b11_nrms.json 2>/dev/null
python bench.py --model nrms --embed null --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r2_b11_nrms_null.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nrms_b -o p -- python3 bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r2_prof_nrms_b.log 2>&1
