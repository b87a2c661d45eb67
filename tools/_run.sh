cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/enq; mkdir -p $O
for r in 1 2 3; do for m in main side; do
LEGO_ENQ=$m python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/b_${m}_$r.json 2>/dev/null
done; done
