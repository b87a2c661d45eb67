cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final2
mkdir -p $O
python -m pytest tests/test_hip_parity.py -m gpu -q -k full_vocabulary 2>&1 | grep -E "^E  |Error|passed|failed" | cut -c1-300 | head -20 > $O/vocab.log
python -m pytest tests/test_hip_parity.py tests/test_dp_device.py -m gpu -q 2>&1 | tail -6 > $O/gputests.log
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/pmc_fetch.log 2>&1
LEGO_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/pmc_write.log 2>&1
LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/prof_serial.log 2>&1
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $O/bench_200.json 2>/dev/null
python bench.py --model nrms --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_nrms.json 2>/dev/null
