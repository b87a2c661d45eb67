"""Print one training step of a rocprofv3 --kernel-trace CSV as a per-queue timeline (us relative to the previous Adam)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) * 3 // 4
a, b = idx[k], idx[k + 1]
t0 = rows[a]['e']
def short(n):
    n = n.replace('lego::', '').replace('void ', '')
    if 'strip_kernel' in n or 'gemm_kernel' in n:
        kind = ('dma_strip' if 'dma_strip_kernel' in n else 'strip') if 'strip_kernel' in n else 'gemm'
        tc = re.search(r'TileCfg<([\d, ]+)', n)
        ld = re.findall(r'(Kc\w+|Mc\w+)', n)[:2]
        ep = re.search(r'EpiT<([^>]*)>', n).group(1).replace('false', '0').replace('true', '1').replace(' ', '')
        return f"{kind}{'[' + tc.group(1).replace(' ', '') + ']' if tc else ''} {','.join(ld)} E<{ep}>"
    return n.split('(')[0][:60]
for r in rows[a:b + 1]:
    print(f"{(r['s']-t0)/1e3:8.1f} {(r['e']-t0)/1e3:8.1f} {(r['e']-r['s'])/1e3:7.1f} q{r['Queue_Id']} {short(r['Kernel_Name'])}")
