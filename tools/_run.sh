cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pp; mkdir -p $O
LEGO_PP=1 python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -4 > $O/tests.log
for m in 1 0; do
LEGO_PP=$m LEGO_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/prof_$m.log 2>&1
LEGO_PP=$m python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/b_$m.json 2>/dev/null
done
