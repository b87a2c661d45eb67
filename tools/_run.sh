cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/order
mkdir -p $O
for r in 1 2; do for m in 0 1 2 3; do
LEGO_BWD_ORDER=$m python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/b_${m}_$r.json 2>/dev/null
done; done
LEGO_BWD_ORDER=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -o p -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > $O/trace1.log 2>&1
