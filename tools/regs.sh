#!/bin/bash
# VGPR / spill / scratch / LDS of every kernel in a HIP source: tools/regs.sh legommenders_amd/csrc/gemm_ops.hip [filter]
src=$1; filt=${2:-.}
out=/tmp/regs_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S $EXTRA "$src" -o $out 2>/dev/null
python3 - "$out" "$filt" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
filt = re.compile(sys.argv[2])
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("lego::", "")
    if filt.search(name):
        print(f"vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}  {name[:150]}")
PY
rm -f $out
