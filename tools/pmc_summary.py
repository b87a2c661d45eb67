"""Per-kernel means of SQ counters from rocprofv3 PMC passes (`LEGO_SERIAL=1 bench.py`, one `--pmc` group per run; never with
trace domains other than the kernel trace):

    python tools/pmc_summary.py out.json <counter_collection.csv> [<counter_collection.csv> ...]

Derived: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): the busy counter is summed over SIMDs (it equals
32 cycles x the number of fp32 MFMAs issued: SQ_INSTS_VALU_MFMA_MOPS_F32 / 4 x 32 on these kernels) and SQ_BUSY_CU_CYCLES over
CUs, four SIMDs each -- the share of a SIMD's cycles in which its matrix pipe works while the CU has work;
lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wait_share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import csv, json, re, sys
from collections import defaultdict


def short(n):
    n = n.replace("lego::", "").replace("void ", "")
    if "strip_kernel" in n or "gemm_kernel" in n:
        kind = ("dma_strip" if "dma_strip_kernel" in n else "strip") if "strip_kernel" in n else "gemm"
        tc = re.search(r"TileCfg<([\d, ]+)", n)
        ld = re.findall(r"(Kc\w+|Mc\w+)", n)[:2]
        ep = re.search(r"EpiT<([^>]*)>", n).group(1).replace("false", "0").replace("true", "1").replace(" ", "")
        return f"{kind}{'[' + tc.group(1).replace(' ', '') + ']' if tc else ''} {','.join(ld)} E<{ep}>"
    return n.split("(")[0][:48]


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not any(t in k for t in ("strip", "gemm", "wino", "oneshot", "pool", "tower", "gather_rows", "adam", "tnd", "tn_kernel", "dropcorr", "mhsa", "rows2")):
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d["launches"] = max(len(v) for v in cs.values())
        if d.get("SQ_BUSY_CU_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            d["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"]), 4)
        if d.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in d:
            d["wait_share"] = round(d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 4)
        if d.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in d:
            d["lds_conflict"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 4)
        res[k] = d
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
        print(f"{k[:70]:70s} mfma_busy {d.get('mfma_busy', '-')}  lds_conflict {d.get('lds_conflict', '-')}  launches {d['launches']}")


if __name__ == "__main__":
    main()
