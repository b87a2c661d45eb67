"""Instruction statistics of one kernel in a hipcc -save-temps / -S assembly file: counts of the instruction kinds that decide
a GEMM loop (MFMA, LDS reads, LDS-DMA, waits, barriers, scratch), and every distinct s_waitcnt vmcnt form.
    python tools/isa_stats.py file.s <substring of the mangled kernel name> [--loops]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
key = sys.argv[2]
for m in re.finditer(r"^(_Z\S+):\s*; @.*?\n(.*?)\n\s+s_endpgm", txt, re.S | re.M):
    if key not in m.group(1):
        continue
    lines = m.group(2).split("\n")
    c = collections.Counter()
    for l in lines:
        t = l.strip().split(" ")[0]
        if t.startswith(("global_load", "ds_read", "ds_write", "flat_", "v_mfma", "s_barrier", "global_store", "scratch", "v_cndmask", "buffer_", "global_atomic")):
            c[t] += 1
        if "s_waitcnt" in l:
            c[l.strip().split(";")[0].strip()] += 1
    print(m.group(1)[:150], len(lines), "lines")
    for k, v in sorted(c.items()):
        print(f"   {v:5d} {k}")
    if "--loops" in sys.argv:       # basic blocks that contain MFMAs: per-block counts
        blocks = re.split(r"\n(\.LBB\S+):", m.group(2))
        for name, body in zip(blocks[1::2], blocks[2::2]):
            n = body.count("v_mfma")
            if n >= 8:
                w = re.findall(r"s_waitcnt [^\n;]*", body)
                print(f"   block {name}: {n} mfma, {body.count('ds_read')} ds_read, {body.count('global_load_lds')} glds, "
                      f"{body.count('s_barrier')} barrier, {body.count('v_cndmask')} cndmask, waits: {collections.Counter(w).most_common(8)}")
