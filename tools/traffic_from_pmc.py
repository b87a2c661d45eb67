"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of `LEGO_SERIAL=1 bench.py`) into
profiles/rNN_traffic.json, which bench.py reads for `roofline.traffic`.

    python tools/traffic_from_pmc.py <fetch_counter_collection.csv>[,<more>] <write_counter_collection.csv>[,<more>] profiles/r01_traffic.json

(comma-separated lists; an entry `gather_rows_hbm=<csv>` is a pass over `tools/gather_hbm.py 105600 8 --uniform-only`: its row-gather launches are
the HBM-sized ones -- the kernel is the one the step uses, it decides to stream on the live row count -- and take that tag)

Units and corrections follow MI355X_MICROARCH.md: both counters are in KiB; on gfx950 FETCH_SIZE counts the 128-B
requests of 16-B-per-lane coalesced reads at 64 B, so the fetched bytes are doubled."""
import csv, json, re, sys
from collections import defaultdict


def tag_of(name):
    n = name.replace("lego::", "")
    if n.startswith("void gather_rows") or n.startswith("gather_rows_kernel") or n.startswith("gather_rows_wave_kernel"):
        return "gather_rows"
    if "wino2_kernel" in n:               # one kernel for both directions (round 5): a training step launches forward, then data gradient
        return "wino2"
    if "tndp_kernel" in n:
        return "conv3_bwd_weight"
    m = re.search(r"rows2_kernel<(\w+), (\w+), (\w+)>", n)            # round 5: plain-row products, <NN form, accumulate, ReLU reference>
    if m:
        return "rows2<%s,%s>" % ("NN" if m.group(1) == "true" else "NT", "accum+relu'" if m.group(3) == "true" else ("accum" if m.group(2) == "true" else "plain"))
    if "wino_kernel<false>" in n:
        return "conv3_fwd"
    if "wino_kernel<true>" in n:
        return "conv3_bwd_data"
    if ("gemm_kernel" in n or "tn_kernel" in n) and "McPair" in n:
        return "conv3_bwd_weight"
    if "conv3_wino_unpack_add_kernel" in n:
        return "conv3_bwd_weight_unpack"
    m = re.match(r"(?:void )?(mhsa_(?:fwd|bwd)_kernel)<(\d+), (\d+)[,>]", n)      # (round 4: the backward has two more template arguments)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    if "dma_strip_kernel" in n:
        ep = re.search(r"EpiT<([^>]*)>", n).group(1).replace("false", "0").replace("true", "1").replace(" ", "")
        return "dma_strip<%s,%s>" % ("NN" if "McRows" in n else "NT", ep)
    if "strip_kernel" in n and "KcConvA" in n and "KcTapW" in n:
        return "conv3_fwd"
    if "strip_kernel" in n and "KcConvA" in n:
        return "conv3_bwd_data"
    if "gemm_kernel" in n and "McShiftRows" in n:
        return "conv3_bwd_weight"
    return None


def per_kernel(paths, counter):
    vals = defaultdict(list)
    for path in paths.split(","):
        retag = None
        if "=" in path:
            retag, path = path.split("=", 1)
        rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == counter]
        disp = {}                                  # one value per dispatch, in dispatch order
        for i, r in enumerate(rows):
            key = int(r["Dispatch_Id"]) if r.get("Dispatch_Id") else i
            if key not in disp:
                disp[key] = [r["Kernel_Name"], 0.0]
            disp[key][1] += float(r["Counter_Value"])
        nth = 0
        for key in sorted(disp):
            name, v = disp[key]
            t = tag_of(name)
            if retag is not None:
                t = retag if t == "gather_rows" else None
            if t == "wino2":
                t = ("conv3_fwd", "conv3_bwd_data")[nth % 2]
                nth += 1
            if t is not None:
                vals[t].append(v)
    return vals


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for t in sorted(set(f) | set(w)):
        fv, wv = f.get(t, []), w.get(t, [])
        if t == "gather_rows":            # the token-row gather, not the small category gather that shares the kernel
            big = max(wv) if wv else 0
            keep = [i for i, v in enumerate(wv) if v > 0.5 * big]
            wv = [wv[i] for i in keep]
            bigf = max(fv) if fv else 0
            fv = [v for v in fv if v > 0.5 * bigf]
        fv, wv = fv[len(fv) // 4:], wv[len(wv) // 4:]
        fk = sum(fv) / max(1, len(fv))
        wk = sum(wv) / max(1, len(wv))
        kernels[t] = {"FETCH_SIZE": fk, "WRITE_SIZE": wk, "fetch_bytes_corrected": fk * 1024 * 2, "write_bytes": wk * 1024,
                      "hbm_bytes_per_launch": fk * 1024 * 2 + wk * 1024, "launches_averaged": [len(fv), len(wv)]}
    import os, subprocess
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        commit = None
    json.dump({"kernel_sources_sha": bench.kernel_sources_sha(), "commit": commit, "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `LEGO_SERIAL=1 bench.py --steps 20 "
                       "--warmup 5`; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B "
                       "for 16 B/lane coalesced reads); averages over the last 3/4 of the launches",
               "kernels": kernels}, open(out, "w"), indent=1)
    for t, v in kernels.items():
        print(t, {k: (round(x / 1e6, 2) if isinstance(x, float) else x) for k, x in v.items()})


if __name__ == "__main__":
    main()
