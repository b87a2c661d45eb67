#!/usr/bin/env python3
"""bench.py -- training impressions/s of the MI355X-native NAML hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full training step over one batch of B = 64 impressions (BASELINE config[1]:
MIND-small NAML hidden=256 bs=64 GloVe): device-side negative sampling + history fetch, ragged
forward, backward, (N>1: one RCCL all-reduce of the flat gradient buffer), Adam -- dropout ON,
fp32 arithmetic (exact-f32 MFMA), all inputs resident in HBM before the timed region.
Synthetic MIND-small-shaped data, random-init weights (no dataset / GloVe on disk).

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the oracle's port of the reference CPU training step on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (impressions per step)")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--model", default="naml", choices=["naml", "nrms"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--time-every", type=int, default=8, help="bracket the tagged kernels with HIP events every N-th timed step")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even at world size 1 (path check)")
    ap.add_argument("--small", action="store_true", help="shrunken world for quick checks (NOT the metric config)")
    return ap.parse_args()


def pmc_traffic():
    """HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, run separately --
    counters cannot be read from inside this process); newest profiles/r*_traffic.json, else {}."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return {}
    try:
        return {k: v.get("hbm_bytes_per_launch") for k, v in json.load(open(files[-1]))["kernels"].items()}
    except Exception:
        return {}


def cpu_baseline(world, B, D, steps):
    """Oracle port of the reference's `--cuda -1` NAML training step, timed on this host's cores."""
    import numpy as np
    from oracle import lego_oracle as O
    from legommenders_amd.synthetic import init_naml_params
    cores = min(os.cpu_count() or 1, 32)     # 32 threads: oversubscribing a 256-thread host is slower for these GEMM sizes
    torch.set_num_threads(cores)
    P = init_naml_params(D=D, V=world["V"], n_cat=world["n_cat"])
    train = []
    for k, v in P.items():
        if not k.endswith("glove.embedding.weight"):
            v.requires_grad_(True)
            train.append(v)
    opt = torch.optim.Adam(train, lr=1e-3)
    rs = np.random.RandomState(0)
    tt = torch.from_numpy(world["title_tok"].astype("int64"))
    ct = torch.from_numpy(world["cat"].astype("int64"))
    S = world["S"]

    def batch():
        rows = rs.randint(0, world["n_rows"], size=B)
        u = world["row_user"][rows]
        cand = np.concatenate([world["row_item"][rows][:, None], rs.randint(0, world["n_items"], size=(B, 4))], 1)
        return (torch.from_numpy(cand.astype("int64")), torch.from_numpy(world["user_hist"][u].astype("int64")),
                torch.from_numpy(world["user_hist_len"][u].astype("int64")))
    O.naml_train_step_cpu(P, opt, tt, ct, *batch())            # warm-up (thread pools, mkldnn primitives)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.naml_train_step_cpu(P, opt, tt, ct, *batch())
    dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 2), "unit": "impressions/s", "cores": cores, "kind": "port",
            "sample": f"{steps} training steps (fwd+bwd+Adam, dropout on) of B={B} NAML hidden={D} on the same synthetic "
                      f"MIND-small-shaped world after 1 warm-up step; torch {torch.__version__} CPU, {cores} threads"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    assert world_size == args.gpus or world_size == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    dist_on = world_size > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
        pg = torch.distributed.group.WORLD

    from legommenders_amd import _lib
    _lib.lib()                                     # fail loudly if the HIP extension is missing
    from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_naml_params, init_nrms_params, make_world
    from legommenders_amd.train_step import DeviceData, TrainStep

    cfg = dict(MIND_SMALL)
    if args.small:
        cfg.update(n_items=5000, n_users=4000, n_rows=20000, V=20000)
    world = make_world(seed=2023, **cfg)
    data = DeviceData(world, dev, rank=rank, world_size=world_size, seed=2023)
    glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
    if args.model == "naml":
        params = init_naml_params(D=args.hidden, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove)
    else:
        params = init_nrms_params(D=args.hidden, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove)
    B = args.batch
    ts = TrainStep(args.model, params, data, B, K=4, lr=1e-3, total_steps=0, seed=2023,
                   process_group=pg, world_size=world_size, dropout=True, force_allreduce=args.force_dist, tail="drop")

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step()
    barrier()
    ts.counter_sum.zero_()
    timers = {}                                      # HIP-event pairs around the tagged kernels, timed region only,
    t0 = time.perf_counter()                         # on every `time_every`-th step (22 event records per step are not free)
    n_timed = 0
    for i in range(args.steps):
        on = args.time_every > 0 and i % args.time_every == 0
        for e in ts.engines:
            e.timers = timers if on else None
        n_timed += int(on)
        loss = ts.step()
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    # ---- the token-row gather (the HBM-bound kernel the north star names) runs on the prefetch stream, overlapped
    # with the previous step, so it is timed here on its own: same kernel, same plan (the last prefetched batch)
    gather_ms, gather_rows = None, 0
    eng = ts.engines[0]
    if hasattr(eng, "gather_tokens") and getattr(eng, "Rc", 0) > 0:
        for _ in range(4):
            eng.gather_tokens()
        blocker = torch.zeros(1 << 27, dtype=torch.float32, device=dev)    # ~0.3 ms of GPU work: the host enqueues the
        torch.cuda.synchronize()                                           # whole timed chain while it runs, so the
        for _ in range(3):                                                 # chain is not host-bound
            blocker.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            eng.gather_tokens()
        b.record()
        torch.cuda.synchronize()
        del blocker
        gather_ms = a.elapsed_time(b) / 20
        gather_rows = int(eng.counters[0].item())

    # ---- per-kernel roofline from the HIP events recorded inside the timed region
    cs = ts.counter_sum.tolist()
    rows_tok, n_inst = cs[0], cs[1]                   # summed over the timed steps
    D, E0 = args.hidden, 300
    kern = {}
    for tag, evs in timers.items():
        ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))
        kern[tag] = {"avg_ms": ms, "launches": len(evs)}
    for e in ts.engines:
        e.timers = None
    launches = max(1, args.steps)                      # one launch of each tagged kernel per step
    rows_per_launch = rows_tok / launches
    yrows_per_launch = (rows_tok + n_inst) / launches
    flops = {                                          # algorithmic flops per launch
        "proj_fwd": 2.0 * rows_per_launch * D * E0,
        "conv3_fwd": 2.0 * rows_per_launch * D * 3 * D,
        "conv3_bwd_data": 2.0 * rows_per_launch * D * 3 * D,
        "conv3_bwd_weight": 2.0 * rows_per_launch * D * 3 * D,
        "proj_bwd_weight": 2.0 * rows_per_launch * D * E0,
        "additive_fwd_item": 2.0 * yrows_per_launch * D * 256,
        "additive_bwd_data": 2.0 * rows_per_launch * D * 256,
        "additive_bwd_weight_item": 2.0 * yrows_per_launch * D * 256,
    }
    for tag, f in flops.items():
        if tag in kern and kern[tag]["avg_ms"] > 0:
            kern[tag]["tflops"] = f / (kern[tag]["avg_ms"] * 1e-3) / 1e12
            kern[tag]["frac_of_f32_mfma_peak"] = kern[tag]["tflops"] / PEAK_F32_MFMA_TFLOPS
    roofline, roofline_gather = None, None
    traffic = pmc_traffic()
    # dominant kernel = the largest GEMM that runs ALONE on the GPU (forward, main stream).  The backward conv
    # GEMMs do the same flops but overlap with side-stream kernels (engine.py), so their event brackets include
    # time-sharing and would under-state the kernel; they are still listed in "kernels" with "overlapped": true.
    solo = ("conv3_fwd", "proj_fwd", "additive_fwd_item")
    for k in kern:
        kern[k]["overlapped"] = k not in solo and k != "gather_rows"
    mf = [(v["avg_ms"], k) for k, v in kern.items() if "tflops" in v and k in solo]
    if mf:
        _, dom = max(mf)
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(kern[dom]["tflops"], 3),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(kern[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic.get(dom),
                    "avg_launch_ms": round(kern[dom]["avg_ms"], 5),
                    "algorithmic_flops_per_launch": flops[dom]}
    if gather_ms:
        kern["gather_rows"] = {"avg_ms": gather_ms, "launches": 20, "overlapped": False, "timed": "standalone, after the timed region"}
        gbytes = gather_rows * E0 * 4 * 2 + gather_rows * 4     # row read + row write + index
        gbs = gbytes / (gather_ms * 1e-3) / 1e9
        roofline_gather = {"kernel": "gather_rows", "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
                           "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": traffic.get("gather_rows"),
                           "avg_launch_ms": round(kern["gather_rows"]["avg_ms"], 5),
                           "algorithmic_bytes_per_launch": gbytes,
                           "dense_reference_bytes_per_launch": B * 55 * 30 * 1200}

    if rank != 0:
        if dist_on:
            torch.distributed.destroy_process_group()
        return
    out = {
        "metric": "train impressions/sec on MIND-small NAML" if args.model == "naml" else "train impressions/sec on MIND-small NRMS",
        "value": round(B * world_size * args.steps / dt, 1), "unit": "impressions/s",
        "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"MIND-small-shaped {args.model.upper()} hidden={args.hidden} bs={B}/GPU GloVe(300d frozen) "
                               f"K=4 negatives S=50 T=30, full train step (sample+fwd+bwd+allreduce+Adam), dropout 0.1"
                               + (" [SMALL WORLD - not the metric config]" if args.small else ""),
                   "global_batch": B * world_size, "parallelism": f"dp{world_size}",
                   "live_token_rows_per_step": round(rows_tok / max(1, args.steps), 1),
                   "item_instances_per_step": round(n_inst / max(1, args.steps), 1)},
        "final_loss": round(final_loss, 5),
        "roofline": roofline, "roofline_gather": roofline_gather,
        "kernels": {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kern.items()},
    }
    if not args.no_cpu_baseline and world_size == 1:
        out["cpu_baseline"] = cpu_baseline(world, B, args.hidden, args.cpu_steps)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
