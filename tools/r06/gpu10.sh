#!/bin/bash
# what bounds the attention core on cold caches: issue / wait / memory-pipe counters of the two launches (PMC passes of their own)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/mhsa_pmc; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TA_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TD_[A-Z_0-9]+|TCC_[A-Z_0-9]+)\b" | sort -u > $O/counters.txt
wc -l $O/counters.txt
cat > $O/probe.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from legommenders_amd._lib import call
from legommenders_amd.kernels import _ptr, _stream, _drop
dev = torch.device("cuda:0")
D, heads, n, Lmax = 256, 8, 1500, 33
rs = np.random.RandomState(0)
lens = rs.randint(8, 33, size=n)
seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
R = int(lens.sum())
qkv, go = torch.randn(R, 3 * D, device=dev), torch.randn(R, D, device=dev)
out, gq = torch.empty(R, D, device=dev), torch.empty(R, 3 * D, device=dev)
probs = torch.zeros(R, heads, Lmax, device=dev)
dr = _drop((0.1, 5, 3))
flush = torch.empty(160 << 20, dtype=torch.float32, device=dev)
for _ in range(6):
    flush.sum()
    call("lego_mhsa_core_fwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(out), D, None, _ptr(probs), Lmax, dr, R, 0, None, None, _stream())
    flush.sum()
    call("lego_mhsa_core_bwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(go), D, None, _ptr(probs), Lmax, dr, R, _ptr(gq), 3 * D, None, 0, None, None, _stream())
torch.cuda.synchronize()
PY
pass1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
pass2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
pass3="TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
pass4="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for p in "$pass1" "$pass2" "$pass3" "$pass4"; do i=$((i+1))
  ok=""; for c in $p; do grep -qx "$c" $O/counters.txt && ok="$ok $c"; done
  echo "pass $i:$ok"
  rocprofv3 --kernel-trace --pmc $ok --output-format csv -d $O/p$i -- python3 $O/probe.py > /dev/null 2> $O/p$i.err
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "mhsa" in k:
        agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done 2>&1 | tee $O/summary.txt
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
