"""Pretty-print a rocprofv3 --stats kernel summary: python tools/kstats.py <..._kernel_stats.csv> [top_n] [steps]"""
import csv, re, sys


def short(n):
    n = n.replace("lego::", "").replace("void ", "")
    m = re.match(r"(\w+)<(.*)>\(", n)
    if "gemm_kernel" in n or "strip_kernel" in n or "tn_kernel" in n or "oneshot_kernel" in n:
        kind = "dma_strip" if "dma_strip_kernel" in n else re.match(r"(\w+?)_kernel", n).group(1)
        tc = re.search(r"TileCfg<([\d, ]+)", n)
        ld = re.findall(r"(Kc\w+|Mc\w+)", n)[:2]
        ep = re.search(r"EpiT<([^>]*)>", n)
        ep = ep.group(1).replace("false", "0").replace("true", "1").replace(" ", "") if ep else ""
        return f"{kind}{'[' + tc.group(1).replace(' ', '') + ']' if tc else ''} {','.join(ld)} E<{ep}>"
    return n.split("(")[0][:70]


rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms" + (f" = {tot / 1e3 / steps:.1f} us/step over {steps:.0f} steps" if steps else ""))
for r in rows[:top]:
    per = f" {float(r['TotalDurationNs']) / 1e3 / steps:8.1f} us/step" if steps else ""
    print(f"{int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):6.2f}%{per}  {short(r['Name'])}")
