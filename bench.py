#!/usr/bin/env python3
"""bench.py -- training impressions/s of the MI355X-native NAML hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one full training step over one batch of B = 64 impressions per GPU (BASELINE config[1]:
MIND-small NAML hidden=256 bs=64 GloVe): device-side negative sampling + history fetch, ragged
forward, backward, (N>1: one RCCL all-reduce of the flat gradient buffer), Adam -- dropout ON,
fp32 arithmetic (exact-f32 MFMA), all inputs resident in HBM before the timed region.
Synthetic MIND-small-shaped data, random-init weights (no dataset / GloVe on disk).

Rank 0 prints ONE JSON line.  `value` / `ms_per_step` come from EXACTLY --steps steps between two
barrier + synchronize brackets, after exactly --warmup untimed steps; before those, half a second of a dummy product wakes the device
(BENCH_SPIN_MS, default 500; 0 = off: see the comment at its place).  Beside it (N = 1 only, all after the timed region, each with its own
steps / ms so the driver's wall clock still bounds them):
  roofline           dominant forward GEMM, HIP-event timed on its launch stream inside the timed region;
                     `frac` prices the DIRECT-conv flops, `mfma_issue_frac` the flops the Winograd form really issues
  roofline_gather    the GloVe row gather: `frac_in_step` = in-step duration on the prefetch stream (HIP events there,
                     overlapped with the previous step's user-side chain); `frac` = back-to-back standalone launches
                     (cache-assisted: most rows then hit the Infinity Cache)
  roofline_step      the whole step against the fp32 matrix peak: algorithmic flops of its big products / step time (`frac`), and the
                     flops the kernels issue (projection per distinct token, Winograd 2/3) the same way (`issued_frac`)
  long_run           2000 steps of the same configuration when --steps is smaller (a 20-step window is 14 ms)
  secondary          NRMS config 3 (with its dominant-kernel roofline, and the trainable-table variant), the worst-case dense NAML world (every history
                     50, every title 30 tokens), a world with MIND-like length statistics, BERT config 5 and the OPT-IN split-bf16 product
                     mode (`split_bf16_opt_in`, its dtype stated there) are NOT the metric; they are printed so that the number's dependence
                     on the model and on raggedness is on record
  allreduce_ms       (N > 1 or --force-dist) one RCCL all-reduce of the flat gradient buffer, timed alone
  dist_path_check    (N = 1) the same on a one-rank RCCL communicator created after the timed region: all-reduce alone + 100 steps with it
  cpu_baseline       the oracle's port of the reference CPU training step on the host cores
"""
from __future__ import annotations

import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP initialises: see legommenders_amd/__init__.py

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_json_out = sys.stdout
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
WINO_ISSUE = 2.0 / 3.0            # Winograd F(2,3): 4 C MACs per row pair and output column instead of 6 C


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (impressions per step)")
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--model", default="naml", choices=["naml", "nrms"])
    ap.add_argument("--embed", default="glove", choices=["glove", "null"],
                    help="nrms only: null = trainable [V, hidden] token table (config/embed/null.yaml) instead of frozen GloVe + Linear")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the long run and the secondary configurations")
    ap.add_argument("--no-bert", action="store_true", help="skip the BERT-base secondary line (config 5)")
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed CPU-baseline steps (after 3 warm-up steps; BASELINE.md section 3)")
    ap.add_argument("--no-dist-check", action="store_true", help="N = 1: skip the one-rank RCCL path check after the timed region")
    ap.add_argument("--time-every", type=int, default=4, help="bracket the roofline kernel and the in-step gather with HIP events every N-th timed step")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even at world size 1 (path check)")
    ap.add_argument("--no-balance", action="store_true", help="N > 1: deal rows r::W instead of by live-row cost (A/B)")
    ap.add_argument("--small", action="store_true", help="shrunken world for quick checks (NOT the metric config)")
    return ap.parse_args()


def kernel_sources_sha():
    """sha256 (16 hex) over the HIP sources the kernels are built from: a PMC summary is valid for exactly one such tree"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "legommenders_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, run separately -- counters cannot be
    read from inside this process): the newest profiles/r*_traffic.json, ONLY when it was taken on the kernel sources of this
    tree (`kernel_sources_sha` recorded by tools/traffic_from_pmc.py).  Returns (bytes per tag, source record); a stale or
    unmarked file yields no bytes and says so in the record -- `roofline.traffic` is then null rather than a number measured
    on other kernels (VERDICT r2 weak #9a)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    now = kernel_sources_sha()
    if not files:
        return {}, {"file": None, "stale": True, "kernel_sources_sha": now}
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return {}, {"file": os.path.relpath(files[-1], ROOT), "stale": True, "kernel_sources_sha": now}
    src = {"file": os.path.relpath(files[-1], ROOT), "taken_on_kernel_sources_sha": d.get("kernel_sources_sha"),
           "taken_on_commit": d.get("commit"), "kernel_sources_sha": now, "stale": d.get("kernel_sources_sha") != now}
    if src["stale"]:
        return {}, src
    return {k: v.get("hbm_bytes_per_launch") for k, v in d["kernels"].items()}, src


def cpu_baseline(world, B, D, steps, with_config1=True):
    """Oracle port of the reference's `--cuda -1` NAML training step, timed on this host's cores."""
    import numpy as np
    from oracle import lego_oracle as O
    from legommenders_amd.synthetic import init_naml_params
    host = os.cpu_count() or 1
    cores = min(host, 32)     # 32 threads: oversubscribing a 256-thread host is slower for these GEMM sizes
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

    def batch():
        rows = rs.randint(0, world["n_rows"], size=B)
        u = world["row_user"][rows]
        cand = np.concatenate([world["row_item"][rows][:, None], rs.randint(0, world["n_items"], size=(B, 4))], 1)
        return (torch.from_numpy(cand.astype("int64")), torch.from_numpy(world["user_hist"][u].astype("int64")),
                torch.from_numpy(world["user_hist_len"][u].astype("int64")))
    warm = 3
    for _ in range(warm):                                      # warm-up (thread pools, mkldnn primitives)
        O.naml_train_step_cpu(P, opt, tt, ct, *batch())
    t0 = time.perf_counter()
    for _ in range(steps):
        O.naml_train_step_cpu(P, opt, tt, ct, *batch())
    dt = time.perf_counter() - t0
    out = {"value": round(B * steps / dt, 2), "unit": "impressions/s", "cores": cores, "host_cores": host, "kind": "port",
           "sample": f"{steps} training steps (fwd+bwd+Adam, dropout on) of B={B} NAML hidden={D}, DENSE reference layout "
                     f"(all 50 history slots x 30 title positions, as the reference computes), on the same synthetic "
                     f"MIND-small-shaped world after {warm} warm-up steps; torch {torch.__version__} CPU, {cores} of {host} "
                     f"host threads (the GPU path skips pad rows; it is timed on the ragged layout)"}
    if with_config1:
        # BASELINE config 1 (the reference's own CPU-runnable case): NAML hidden=64 bs=32, same world, same port
        B1, D1 = 32, 64
        P1 = init_naml_params(D=D1, V=world["V"], n_cat=world["n_cat"], glove=P["embedding_vocab_table.glove.embedding.weight"])
        tr1 = []
        for k, v in P1.items():
            if not k.endswith("glove.embedding.weight"):
                v.requires_grad_(True)
                tr1.append(v)
        opt1 = torch.optim.Adam(tr1, lr=1e-3)

        def batch1():
            c, h, hl = batch()
            return c[:B1], h[:B1], hl[:B1]
        for _ in range(warm):
            O.naml_train_step_cpu(P1, opt1, tt, ct, *batch1())
        t0 = time.perf_counter()
        for _ in range(steps):
            O.naml_train_step_cpu(P1, opt1, tt, ct, *batch1())
        d1 = time.perf_counter() - t0
        out["config1"] = {"value": round(B1 * steps / d1, 2), "unit": "impressions/s", "cores": cores,
                          "sample": f"{steps} training steps of B={B1} NAML hidden={D1} (BASELINE config 1), after {warm} warm-up steps"}
    return out


def dense_world(world):
    """the same world with every title at T tokens and every history at S clicks: no pad row to skip"""
    import numpy as np
    rs = np.random.RandomState(77)
    w = dict(world)
    n_items, T, n_users, S = world["n_items"], world["T"], world["n_users"], world["S"]
    w["title_len"] = np.full(n_items, T, dtype=np.int32)
    w["title_tok"] = np.minimum(rs.zipf(1.2, size=(n_items, T)) - 1, world["V"] - 1).astype(np.int32)
    w["user_hist_len"] = np.full(n_users, S, dtype=np.int32)
    w["user_hist"] = rs.randint(0, n_items, size=(n_users, S)).astype(np.int32)
    return w


def mind_like_world(world):
    """the same world with length statistics close to the real MIND-small tables (approximate; no MIND on disk): titles
    ~ N(11.5, 3.5) word pieces clipped to [3, T] (the MIND paper reports 11.5 words per title), histories ~ geometric with
    mean 33 clipped to [1, S] (about 27 clicks after the cap of 50: a large share of users fills all 50 slots)"""
    import numpy as np
    rs = np.random.RandomState(78)
    w = dict(world)
    n_items, T, n_users, S = world["n_items"], world["T"], world["n_users"], world["S"]
    tl = np.clip(np.rint(rs.normal(11.5, 3.5, size=n_items)), 3, T).astype(np.int32)
    tok = np.minimum(rs.zipf(1.2, size=(n_items, T)) - 1, world["V"] - 1).astype(np.int32)
    w["title_len"] = tl
    w["title_tok"] = np.where(np.arange(T)[None, :] < tl[:, None], tok, -1).astype(np.int32)
    hl = np.clip(rs.geometric(1.0 / 33.0, size=n_users), 1, S).astype(np.int32)
    w["user_hist_len"] = hl
    w["user_hist"] = (rs.randint(0, n_items, size=(n_users, S)) * (np.arange(S)[None, :] < hl[:, None])).astype(np.int32)
    return w


def timed_steps(ts, steps, warmup, barrier, time_every=0, tags=None):
    """`warmup` untimed steps, then exactly `steps` steps between two barriers; returns (seconds, timers, loss).  On every
    `time_every`-th timed step the kernels tagged `tags` (None = all tagged kernels) are bracketed with HIP events on their
    launch stream."""
    for _ in range(warmup):
        ts.step()
    barrier()
    ts.counter_sum.zero_()
    timers = {}
    ts.engine.timer_tags = tags
    # (no gc.collect() / gc.disable() here: the collection takes tens of milliseconds with the device idle, and the first timed
    # steps then run on ramped-down clocks -- measured 0.676 against 0.642 ms over the driver's 20-step window, same box)
    t0 = time.perf_counter()
    loss = None
    for i in range(steps):
        on = time_every > 0 and i % time_every == 0
        ts.engine.timers = timers if on else None
        loss = ts.step()
    timed_steps.host_s = time.perf_counter() - t0      # the host has enqueued everything; the device is still working
    barrier()
    dt = time.perf_counter() - t0
    ts.engine.timers, ts.engine.timer_tags = None, None
    return dt, timers, loss


def tagged_steps(ts, steps, barrier):
    """`steps` untimed steps with EVERY tagged kernel bracketed by HIP events: the per-kernel table.  Bracketing all ~12
    kernels costs a step ~90 us (tools/step_profile.py), so it is kept out of the timed region, which brackets only the roofline
    kernel and the in-step gather.  Returns (timers, live token rows per step, item instances per step)."""
    barrier()
    ts.counter_sum.zero_()
    timers = {}
    ts.engine.timers, ts.engine.timer_tags = timers, None
    for _ in range(steps):
        ts.step()
    barrier()
    ts.engine.timers = None
    cs = ts.counter_sum.tolist()
    tagged_steps.uniq = cs[6] / steps
    tagged_steps.hist = cs[3] / steps                       # clicked-item instances (the user side's rows)
    return timers, cs[0] / steps, cs[1] / steps


def kernel_table(timers):
    return {tag: {"avg_ms": sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs)), "launches": len(evs)} for tag, evs in timers.items()}


def launcher_command(args_list, gpus, port):
    """argv of the N-rank launch a plain `python bench.py --gpus N` turns into (the driver's own command shape)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(args_list)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`--gpus N` with N > 1 outside torchrun: this parent (which has made NO GPU call -- device_count() does not initialise
    HIP on this image) starts N fresh ranks as a child process and relays rank 0's JSON line and the exit code.  It never
    prints a line itself, so `n_gpus` can not differ from --gpus."""
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible; refusing to measure fewer ranks than asked",
              file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    res = subprocess.run(launcher_command(argv, args.gpus, free_port()), env=env)
    return res.returncode


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))
    # the ONE JSON line goes to the process's real stdout; everything else that writes to file descriptor 1 from here on
    # (RCCL prints a version banner there when a communicator is created) is sent to stderr, so the line stays alone
    global _json_out
    _json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        print(f"bench.py: --gpus {args.gpus} under a launcher with WORLD_SIZE={world_size}: launch with "
              f"--nproc-per-node == --gpus", file=sys.stderr)
        sys.exit(2)
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
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
    data = DeviceData(world, dev, rank=rank, world_size=world_size, seed=2023, balance=None if args.no_balance else args.batch)
    glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
    B, D, E0 = args.batch, args.hidden, 300

    def make_ts(kind, d, force=False, embed="glove", pg=pg):
        init = init_naml_params if kind == "naml" else init_nrms_params
        use_glove = not (kind == "nrms" and embed == "null")
        params = init(D=D, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove if use_glove else None)
        return TrainStep(kind, params, d, B, K=4, lr=1e-3, total_steps=0, seed=2023, process_group=pg, world_size=world_size,
                         dropout=True, force_allreduce=force, tail="drop", glove=use_glove)

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ts = make_ts(args.model, data, force=args.force_dist, embed=args.embed)
    # Device wake-up before the W warm-up steps: half a second of a dummy product.  The driver's command (`--steps 20 --warmup 5`) reaches
    # the timed region 3 ms after the first launch of a process that has spent seconds building tables on the host; on a fresh box that
    # window measured 0.696-0.702 ms per step without this and 0.642-0.645 with it (four alternating fresh-box runs; the 2 000-step
    # `long_run` of the same processes: 0.625-0.630 either way).  Nothing inside the timed region changes; BENCH_SPIN_MS=0 turns it off.
    spin_ms = float(os.environ.get("BENCH_SPIN_MS", "500"))
    if spin_ms > 0:                                # (round 6: the path's own product, so that rocprof summaries of this command hold no library kernel)
        from legommenders_amd import kernels as _K
        a_, w_ = torch.randn(16384, 1024, device=dev), torch.randn(1024, 1024, device=dev) * 0.03
        o_ = torch.empty(16384, 1024, device=dev)
        t_ = time.perf_counter()
        while (time.perf_counter() - t_) * 1e3 < spin_ms:
            for _ in range(10):
                _K.linear_fwd(a_, w_, None, act=2, out=o_)
            torch.cuda.synchronize()
        del a_, w_, o_
    # Round 5: what the first ~15 steps of a process pay is the RUNTIME's start-up besides the device's clocks -- steps 5-9 run 5 % above what their
    # batch sizes predict (tools/step_profile.py), whatever model instance runs them.  BENCH_PREWARM_STEPS (default 30; 0 = off) runs that many
    # training steps of a SCRATCH instance of the same model first (its own parameters, optimiser state and sample stream; the measured instance
    # `ts` is untouched, its W warm-up steps and K timed steps are what the contract says).  Round 6: the scratch instance's FIRST steps are themselves
    # taken in the contract's form -- min(W, 5) warm-up steps, then min(K, 20) steps between two barriers -- and reported as `value_without_prewarm`:
    # what this command measures when its window is the first thing the process does (the rest of the scratch steps follow untimed).
    prewarm = int(os.environ.get("BENCH_PREWARM_STEPS", "30"))
    cold = None
    if prewarm > 0:
        scratch = make_ts(args.model, data, force=args.force_dist, embed=args.embed)
        k0, w0 = min(args.steps, 20), min(args.warmup, 5)
        d0, _, _ = timed_steps(scratch, k0, w0, barrier)
        cold = {"value_without_prewarm": round(B * world_size * k0 / d0, 1), "ms_per_step_without_prewarm": round(d0 / k0 * 1e3, 4),
                "steps": k0, "warmup": w0, "note": "a scratch model instance's first steps in this process, taken before the measured instance runs"}
        for _ in range(max(0, prewarm - k0 - w0)):
            scratch.step()
        torch.cuda.synchronize()
        del scratch
        torch.cuda.empty_cache()
    in_region = {"naml": {"conv3_fwd", "gather_rows_in_step", "expand_rows_in_step"}, "nrms": {"qkv_fwd_item"}}[args.model]
    # every bracketed step costs ~0.08 ms of event / barrier-packet overhead (tools/step_profile.py): short runs bracket two steps
    every = args.time_every if args.steps > 40 or args.time_every == 0 else max(args.time_every, args.steps // 2)
    dt, timers, loss = timed_steps(ts, args.steps, args.warmup, barrier, every, tags=in_region)
    host_ms = timed_steps.host_s / args.steps * 1e3
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    cs = ts.counter_sum.tolist()
    rows_per_launch, inst_per_launch = cs[0] / max(1, args.steps), cs[1] / max(1, args.steps)
    dedup = bool(getattr(ts.engine, "dedup", False))
    uniq_per_launch = cs[6] / max(1, args.steps) if dedup else rows_per_launch     # distinct tokens: the rows the projection runs on
    # the roofline kernel on 16 more steps in the timed region's own conditions (only the in-region tags bracketed: bracketing EVERY
    # kernel serialises the streams and the conv then runs ~8 % faster than it does in the step; the region itself brackets 2 of the
    # driver's 20 steps, too few for a stable mean) ...
    ts.counter_sum.zero_()
    _, tm_roof, _ = timed_steps(ts, 16, 0, barrier, 1, tags=in_region)
    cs_roof = ts.counter_sum.tolist()
    # ... and every tagged kernel on a few extra steps; the kernels bracketed inside the region keep their in-region figures
    tm_all, rows_tab, inst_tab = tagged_steps(ts, 8, barrier)
    uniq_tab = tagged_steps.uniq
    kern = kernel_table(tm_all)
    kern_in = kernel_table(timers)

    # ---- the token-row gather back to back, on its own (cache-assisted: repeat launches of the same plan)
    eng = ts.engine
    gather_ms, gather_rows = None, 0
    if hasattr(eng, "gather_tokens") and getattr(eng, "Rc", 0) > 0:
        from legommenders_amd.engine import _ptr, _stream
        from legommenders_amd._lib import call as _call
        glove_w = eng.P["embedding_vocab_table.glove.embedding.weight"]

        def gather_once():                              # the row gather alone, on the plan of the last step
            if dedup:
                _call("lego_gather_rows", _ptr(glove_w), E0, E0, _ptr(eng.uniq), eng.Uc, eng.cnt(6), _ptr(eng.Xu), E0, 0, _stream())
            else:
                eng.gather_tokens()
        for _ in range(4):
            gather_once()
        blocker = torch.zeros(1 << 27, dtype=torch.float32, device=dev)    # ~0.3 ms of GPU work: the host enqueues the
        torch.cuda.synchronize()                                           # whole timed chain while it runs, so the
        for _ in range(3):                                                 # chain is not host-bound
            blocker.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            gather_once()
        b.record()
        torch.cuda.synchronize()
        del blocker
        gather_ms = a.elapsed_time(b) / 20
        gather_rows = int(eng.counters[6 if dedup else 0].item())

    # ---- per-kernel roofline: algorithmic flops per launch over the HIP-event durations
    def flops_for(rows, inst, uniq=None):
        yrows = rows + inst
        if args.model == "naml":
            return {"proj_fwd": 2.0 * (rows if uniq is None else uniq) * D * E0,     # de-duplicated: one row per distinct token
                    "conv3_fwd": 2.0 * rows * D * 3 * D,
                    "conv3_bwd_data": 2.0 * rows * D * 3 * D,
                    "conv3_bwd_weight": 2.0 * rows * D * 3 * D,
                    "proj_bwd_weight": 2.0 * (rows if (uniq is None or not getattr(eng, "dedup_bwd", False)) else uniq) * D * E0,
                    "additive_fwd_item": 2.0 * yrows * D * 256,
                    "additive_bwd_data": 2.0 * rows * D * 256,
                    "additive_bwd_weight_item": 2.0 * yrows * D * 256}
        return nrms_flops(rows, D, E0, per_key=uniq if per_key_of(eng) else None)

    if args.model == "naml":
        solo = ("conv3_fwd", "proj_fwd", "additive_fwd_item")
        wino = ("conv3_fwd", "conv3_bwd_data", "conv3_bwd_weight") if getattr(eng, "wino", False) else ()
    else:
        solo, wino = ("qkv_fwd_item", "out_proj_fwd_item", "linear_fwd_item", "outlin_fwd_item", "additive_fwd_item"), ()
    flops_tab = flops_for(rows_tab, inst_tab, uniq_tab if dedup else None)
    price(kern, flops_tab, wino)
    if args.model == "naml":
        price(kern, naml_small_flops(tagged_steps.hist, D), ())
        price_hbm(kern, naml_stream_bytes(rows_tab, inst_tab, uniq_tab if dedup else rows_tab, D))
    flops = flops_for(rows_per_launch, inst_per_launch, uniq_per_launch if dedup else None)
    price(kern_in, flops, wino)
    if args.model == "nrms":
        price_hbm(kern, nrms_core_bytes(rows_tab, D))
    for k in kern:
        kern[k]["timed"] = "8 steps after the timed region, every tagged kernel bracketed"
    kern_roof = kernel_table(tm_roof)
    flops_roof = flops_for(cs_roof[0] / 16, cs_roof[1] / 16, cs_roof[6] / 16 if dedup else None)
    price(kern_roof, flops_roof, wino)
    rnd = lambda v: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()}
    for k, v in kern_in.items():
        v["timed"] = "inside the timed region"
        if k in solo and k in kern_roof:           # the roofline kernel: 16 launches in the region's conditions are the figure; the region's
            every_tagged = kern.get(k)             # own 2 launches and the all-kernels-bracketed form stay beside it
            kern[k] = dict(kern_roof[k], timed="16 steps after the timed region, only the in-region tags bracketed")
            kern[k]["in_region"] = rnd(v)
            if every_tagged is not None:
                kern[k]["every_kernel_bracketed"] = rnd(every_tagged)
        else:                                      # prefetch-stream kernels keep their in-region figures
            kern[k] = v
    for k in kern:
        kern[k]["overlapped"] = k not in solo and not k.startswith("gather_rows") and not k.startswith("expand_rows")
    traffic, traffic_src = pmc_traffic()
    roofline = dominant(kern, dict(flops_tab, **{k: flops_roof[k] for k in kern_roof if k in flops_roof and k in solo}), solo, traffic)   # flops of the batches its launches were timed on
    if roofline is not None:
        roofline["traffic_source"] = traffic_src
    roofline_gather = None
    if gather_ms:
        gbytes = gather_rows * E0 * 4 * 2 + gather_rows * 4     # row read + row write + index
        gbs = gbytes / (gather_ms * 1e-3) / 1e9
        # `frac` / `achieved` are the IN-STEP figures (set below from the HIP events on the prefetch stream inside the timed region):
        # the launch as the training step really runs it.  The back-to-back repeat of one plan is cache-assisted (the PMC pass shows a
        # third of its row reads reaching HBM) and is kept beside it as `*_back_to_back` -- it is not an HBM fraction
        roofline_gather = {"kernel": "gather_rows", "bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS,
                           "unit": "GB/s", "frac": None, "traffic": traffic.get("gather_rows"),
                           "achieved_back_to_back": round(gbs, 1), "frac_back_to_back": round(gbs / PEAK_HBM_GBS, 4),
                           "avg_launch_ms_back_to_back": round(gather_ms, 5), "timed_back_to_back": "20 back-to-back launches of one plan "
                           "after the timed region: cache-assisted (the PMC pass shows a third of the row reads reaching HBM)",
                           "algorithmic_bytes_per_launch": gbytes, "rows_gathered_per_launch": gather_rows,
                           "dense_reference_bytes_per_launch": B * 55 * 30 * 1200}
        ins = kern.get("gather_rows_in_step")
        if ins and ins["avg_ms"] > 0:
            b_in = uniq_per_launch * (E0 * 4 * 2 + 4)
            g_in = b_in / (ins["avg_ms"] * 1e-3) / 1e9
            g_rd = uniq_per_launch * E0 * 4 / (ins["avg_ms"] * 1e-3) / 1e9        # the row READS only
            roofline_gather.update({"achieved": round(g_in, 1), "frac": round(g_in / PEAK_HBM_GBS, 4), "avg_launch_ms": round(ins["avg_ms"], 5),
                                    "algorithmic_bytes_per_launch": int(b_in),
                                    "note": "an 11 MB, ~14 us launch after the token de-duplication (4.5 k distinct rows of 1200 B): latency-bound, "
                                            "far below the HBM roof by construction; the dense reference would move 127 MB here",
                                    "achieved_in_step": round(g_in, 1), "frac_in_step": round(g_in / PEAK_HBM_GBS, 4),
                                    "achieved_reads_only_in_step": round(g_rd, 1),
                                    "frac_reads_only_in_step": round(g_rd / PEAK_HBM_GBS, 4),
                                    "avg_launch_ms_in_step": round(ins["avg_ms"], 5), "launches_in_step": ins["launches"],
                                    "timed_in_step": "HIP events on the prefetch stream inside the timed region (the gather of "
                                                     "batch N+1 runs beside batch N's user-side chain)"})

    extra = {}
    exp_in = kern.get("expand_rows_in_step")
    if dedup and exp_in and exp_in["avg_ms"] > 0:
        # the de-duplicated path's HBM-heavy data movement: X[r] = Xu[inv[r]] for the weight gradient (prefetch stream, in step):
        # reads the distinct rows (L2-resident after the table gather), writes one 1200-B row per token row
        eb = rows_per_launch * E0 * 4 + uniq_per_launch * E0 * 4 + rows_per_launch * 4
        eg = eb / (exp_in["avg_ms"] * 1e-3) / 1e9
        extra["roofline_expand"] = {"kernel": "expand_rows (X = Xu[inv])", "bound": "hbm", "achieved": round(eg, 1), "peak": PEAK_HBM_GBS,
                                    "unit": "GB/s", "frac": round(eg / PEAK_HBM_GBS, 4), "avg_launch_ms": round(exp_in["avg_ms"], 5),
                                    "algorithmic_bytes_per_launch": int(eb), "launches": exp_in["launches"],
                                    "timed": "HIP events on the prefetch stream inside the timed region"}
    if dist_on:                                    # the step's one collective on its own: 20 all-reduces of the gradient buffer
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts.sync_gradients()
        barrier()
        a.record()
        for _ in range(20):
            ts.sync_gradients()
        b.record()
        torch.cuda.synchronize()
        extra["allreduce_ms"] = round(a.elapsed_time(b) / 20, 4)
        extra["allreduce_bytes"] = ts.fp.numel * 4
        ts.fp.grad.zero_()
        seen = torch.ones(1, dtype=torch.int32, device=dev)         # every rank that really ran adds one
        torch.distributed.all_reduce(seen)
        extra["ranks_seen"] = int(seen.item())
    if world_size == 1 and not dist_on and not args.no_secondary and not args.small:
        if args.steps < 2000:                      # the same configuration over a window long enough to average out jitter (and
            n_long = 2000                          # for a once-a-second GPU-busy sampler to see the device at work: ~1.3 s)
            dl, _, _ = timed_steps(ts, n_long, 0, barrier)
            extra["long_run"] = {"steps": n_long, "ms_per_step": round(dl / n_long * 1e3, 4), "value": round(B * n_long / dl, 1),
                                 "unit": "impressions/s"}
        sec = {}
        del ts
        torch.cuda.empty_cache()
        # config 3: NRMS (MHSA news / user encoders), same world, GloVe variant
        other = "nrms" if args.model == "naml" else "naml"
        t2 = make_ts(other, data)
        d2, _, _ = timed_steps(t2, 60, 10, barrier)
        c2 = t2.counter_sum.tolist()
        tm2, r2, _ = tagged_steps(t2, 8, barrier)            # per-kernel table on 8 steps after the timed ones
        k2 = kernel_table(tm2)
        f2 = nrms_flops(r2, D, E0, per_key=tagged_steps.uniq if per_key_of(t2.engine) else None) if other == "nrms" else {}
        price(k2, f2, ())
        if other == "nrms":
            price_hbm(k2, nrms_core_bytes(r2, D))      # the attention core against the HBM roof, not the matrix one
        sec[f"{other}_hidden{D}_bs{B}"] = {
            "workload": f"MIND-small-shaped {other.upper()} hidden={D} bs={B} GloVe, full train step (BASELINE config 3)",
            "steps": 60, "warmup": 10, "ms_per_step": round(d2 / 60 * 1e3, 4), "value": round(B * 60 / d2, 1), "unit": "impressions/s",
            "live_token_rows_per_step": round(c2[0] / 60, 1),
            **({"distinct_tokens_per_step": round(c2[6] / 60, 1),
                "projection": "once per DISTINCT token of the batch, expanded to the sequence rows (LEGO_NRMS_DEDUP=0: row by row)"}
               if other == "nrms" and getattr(t2.engine, "dedup", False) else {}),
            "roofline": dominant(k2, f2, tuple(f2), {}),
            "kernels": {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in k2.items()}}
        del t2
        torch.cuda.empty_cache()
        # config 3, the OTHER embedding variant (config/embed/null.yaml): trainable [V, D] token table -- its row gather (three summed
        # look-ups in one pass, 1 KB rows) sits on the step's critical path and is the HBM-bound launch of this variant
        if other == "nrms":
            tn = make_ts("nrms", data, embed="null")
            dn, _, _ = timed_steps(tn, 60, 10, barrier)
            cn = tn.counter_sum.tolist()
            tmn, rn, _ = tagged_steps(tn, 8, barrier)
            kn = kernel_table(tmn)
            per_key = bool(getattr(tn.engine, "qkv_dedup", False))
            un = tagged_steps.uniq
            fn_ = nrms_flops(rn, D, E0, per_key=un if per_key else None)
            price(kn, fn_, ())
            price_hbm(kn, nrms_core_bytes(rn, D))
            # per-key form (round 4): ONE look-up per distinct key of the batch, the in-projection over those keys, q|k|v expanded to the rows
            gb = (un if per_key else rn) * (D * 4.0 * 2 + 16.0)     # table row read + E row written + the index words
            hb = {"embed_gather_item": gb}
            if per_key:
                hb["qkv_expand_item"] = rn * (3 * D * 4.0 + 4.0) + un * 3 * D * 4.0          # one 3 KB q|k|v row written per sequence row
                hb["qkv_bwd_segsum"] = rn * (3 * D * 4.0 + 8.0) + un * 3 * D * 4.0           # every d(qkv) row read once, one sum row per key
            price_hbm(kn, hb)
            g = kn.get("embed_gather_item", {})
            sec[f"nrms_null_hidden{D}_bs{B}"] = {
                "workload": f"MIND-small-shaped NRMS hidden={D} bs={B}, trainable 400k x {D} token table (embed/null), full train step",
                "steps": 60, "warmup": 10, "ms_per_step": round(dn / 60 * 1e3, 4), "value": round(B * 60 / dn, 1), "unit": "impressions/s",
                "live_token_rows_per_step": round(cn[0] / 60, 1), "distinct_keys_per_step": round(cn[6] / 60, 1) if per_key else None,
                "in_projection": ("once per DISTINCT key of the batch (token / [SEP] / category id), q|k|v expanded to the sequence rows; weight and table "
                                  "gradients from per-key sums of d(qkv) (exact; LEGO_NRMS_QKV_DEDUP=0: row by row)") if per_key else "row by row",
                "roofline_gather": {"kernel": "embed_gather_item (lego_expand_rows: token + [SEP] + category look-ups summed in one pass)",
                                    "bound": "hbm", "achieved": round(g.get("gb_per_s", 0.0), 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": round(g.get("frac_of_hbm_peak", 0.0), 4), "avg_launch_ms": round(g.get("avg_ms", 0.0), 5),
                                    "algorithmic_bytes_per_launch": int(gb), "timed": "HIP events on the main stream, 8 steps after the timed ones"},
                "kernels": {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kn.items()}}
            del tn
            torch.cuda.empty_cache()
        # worst case for the ragged plan: nothing to skip
        dd = DeviceData(dense_world(world), dev, seed=2023)
        t3 = make_ts("naml", dd)
        d3, tm3, _ = timed_steps(t3, 40, 10, barrier, 8, tags={"conv3_fwd"})
        c3 = t3.counter_sum.tolist()
        k3 = kernel_table(tm3)
        r3 = c3[0] / 40
        price(k3, {"conv3_fwd": 2.0 * r3 * D * 3 * D}, ("conv3_fwd",))
        sec["naml_dense_worst_case"] = {
            "workload": f"NAML hidden={D} bs={B}, every history {cfg['S']} clicks and every title {cfg['T']} tokens (no pad row to skip)",
            "steps": 40, "warmup": 10, "ms_per_step": round(d3 / 40 * 1e3, 4), "value": round(B * 40 / d3, 1), "unit": "impressions/s",
            "live_token_rows_per_step": round(r3, 1), "item_instances_per_step": round(c3[1] / 40, 1),
            "conv3_fwd": {k: (round(v, 5) if isinstance(v, float) else v) for k, v in k3.get("conv3_fwd", {}).items()}}
        del t3, dd
        torch.cuda.empty_cache()
        # the row gather where it IS bound by HBM: (a) the dense world with the projection's token de-duplication
        # off -- the step then gathers all 105.6 k token rows of a batch (127 MB read + 127 MB written) on the prefetch stream;
        # (b) the same launch alone, uniform random rows of the 480 MB table, every launch on a cold Infinity Cache (tools/gather_hbm.py)
        prev_dedup = os.environ.get("LEGO_DEDUP")
        os.environ["LEGO_DEDUP"] = "0"
        try:
            dd = DeviceData(dense_world(world), dev, seed=2023)
            t5 = make_ts("naml", dd)
            d5, tm5, _ = timed_steps(t5, 24, 6, barrier, 1, tags={"gather_rows_in_step"})
            c5 = t5.counter_sum.tolist()
            g5 = kernel_table(tm5).get("gather_rows_in_step", {})
            b5 = c5[0] / 24 * (E0 * 4 * 2 + 4)
            gh = {"in_step_dense_dedup_off": {
                "workload": "dense world, LEGO_DEDUP=0: every token row of the batch gathered (prefetch stream, beside the previous step)",
                "rows_per_launch": round(c5[0] / 24, 1), "algorithmic_bytes_per_launch": int(b5), "launches": g5.get("launches"),
                "avg_launch_ms": round(g5.get("avg_ms", 0.0), 5), "ms_per_step": round(d5 / 24 * 1e3, 4),
                "achieved": round(b5 / (g5["avg_ms"] * 1e-3) / 1e9, 1) if g5.get("avg_ms") else None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(b5 / (g5["avg_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if g5.get("avg_ms") else None}}
            del t5, dd
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import gather_hbm
            # cold caches two ways (profiles/r06_gather.txt): behind a 640 MB READ (clean lines: the kernel's own traffic only) and behind a
            # 640 MB WRITE (rounds 4-5's harness: 256 MB of dirty lines that the gather's stores must evict -- their write-back is timed with it)
            gh["alone_cold_cache"] = gather_hbm.measure(dev, table=glove, flush_kind="read")
            gh["alone_cold_dirty_cache"] = gather_hbm.measure(dev, table=glove)
            gh["traffic"] = traffic.get("gather_rows_hbm")
            sec["gather_rows_hbm_bound"] = gh
        except Exception as exc:                   # noqa: BLE001 -- recorded like the other secondaries' failures; the metric line stands
            sec["gather_rows_hbm_bound"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        finally:
            if prev_dedup is None:
                os.environ.pop("LEGO_DEDUP", None)
            else:
                os.environ["LEGO_DEDUP"] = prev_dedup
        # length statistics close to the real MIND-small tables
        dm = DeviceData(mind_like_world(world), dev, seed=2023)
        t4 = make_ts("naml", dm)
        d4, _, _ = timed_steps(t4, 100, 10, barrier)
        c4 = t4.counter_sum.tolist()
        sec["naml_mind_like_lengths"] = {
            "workload": f"NAML hidden={D} bs={B}, titles ~ N(11.5, 3.5) tokens clipped to [3, {cfg['T']}], histories ~ geometric(mean 33) "
                        f"clipped to [1, {cfg['S']}] -- approximate MIND-small statistics (no MIND on disk)",
            "steps": 100, "warmup": 10, "ms_per_step": round(d4 / 100 * 1e3, 4), "value": round(B * 100 / d4, 1), "unit": "impressions/s",
            "live_token_rows_per_step": round(c4[0] / 100, 1), "item_instances_per_step": round(c4[1] / 100, 1)}
        del t4, dm
        torch.cuda.empty_cache()
        # config 5: BERT-base news encoder through the plug-in route (random-init BertConfig() defaults = bert-base-uncased
        # shapes; round 4: the transformer blocks run on the path's own kernels over ragged rows -- legommenders_amd/bert_native.py --
        # as do gather / Linear / pools / dot + CE).  Both modes the reference has: tune_from = 0 (11 blocks train) and cached-layer (tune_from = 9:
        # layer-9 states of every item resident in HBM, 2 blocks train).  Few steps: a step is 0.1-0.6 s.
        if not args.no_bert:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bert_naml_bench
            from legommenders_amd.loader.env import Env
            bsec = {}
            # (six warm-up steps, the MEDIAN of six timed ones, and tools/bert_naml_bench.py synchronises after every step: every ragged batch larger
            # than the ones before it sends the caching allocator for fresh segments -- 300-360 ms for that step against 120 once the pool has
            # grown, and in this process (after the other secondaries' empty_cache) up to five of the first seven steps were such steps)
            for name, tf, st, wu in (("tune_from_0", 0, 6, 6), ("tune_from_9_cached_layer", 9, 10, 2)):
                r = bert_naml_bench.run(batch=B, steps=st, warmup=wu, layers=12, hidden=D, tune_from=tf)
                bsec[name] = {"steps": st, "warmup": wu, "ms_per_step": round(r["s_per_step"] * 1e3, 2), "value": r["impressions_per_s"],
                              "timing": r["timing"], "step_ms": r["step_ms"], "step_ms_max_over_min": r["step_ms_max_over_min"], "live_rows_per_step": r["live_rows_per_step"],
                              "us_per_live_row": r["us_per_live_row"], "us_per_live_row_max_over_min": r["us_per_live_row_max_over_min"],
                              "workspace_arena": r["workspace_arena"], "torch_allocator": r["torch_allocator"], "ms_per_step_mean": round(r["s_per_step_mean"] * 1e3, 2),
                              "unit": "impressions/s", "bert_blocks_run": r["bert_layers_run"], "trainable_params": r["trainable_params"],
                              "layer_cache_s": r["layer_cache_s"], "layer_cache_GB": r["layer_cache_GB"], "final_loss": round(r["loss"], 4),
                              "item_page_size_yaml": r["item_page_size"], "item_page_effective": r["effective_item_page"],
                              "blocks_on": r["blocks_on"], "kernels": r["kernels"]}
                torch.cuda.empty_cache()
            Env.set_lm_cache(False)
            sec["bert_naml_base"] = dict(bsec, workload=f"MIND-small-shaped BERT-NAML (BASELINE config 5): BertConfig() defaults "
                                         f"(768 x 12 blocks x 12 heads, random init), item_page_size 64 in the yaml (raised to one call per batch side: same values, "
                                         f"fuller launches; only the LIVE history slots are encoded), hidden={D} bs={B}, "
                                         f"5 000-item world, full plug-in train step (PluginStep: device sampler ids, fwd, bwd, one lego_adam_step over the flat buffers), fp32")
        # OPT-IN product mode (include/lego_hip.h: lego_set_product_mode; never the headline): split-bf16 operands (bf16 x 3, fp32
        # accumulate) for the large dense products.  Same workloads, same seeds, engines rebuilt in that mode (NAML takes the direct
        # conv: the Winograd kernels have no split form); logits within ~1e-6 of the oracle, gradients 2e-3 (tests/test_split_bf16.py)
        from legommenders_amd import _lib as _L
        _L.set_product_mode(_L.SPLIT_BF16)
        try:
            ssec = {"dtype": "split-bf16 (every operand = hi + lo bf16, three bf16 MFMA terms, fp32 accumulate): NOT the parity mode",
                    "how": "lego_set_product_mode(1) / LEGO_SPLIT_BF16=1; products of >= 2048 rows only, everything else exact f32"}
            for kind, emb in (("naml", "glove"), ("nrms", "glove"), ("nrms", "null")):
                tsx = make_ts(kind, data, embed=emb)
                dx, _, lx = timed_steps(tsx, 100, 20, barrier)
                ssec[f"{kind}_{emb}_hidden{D}_bs{B}"] = {"steps": 100, "warmup": 20, "ms_per_step": round(dx / 100 * 1e3, 4),
                                                      "value": round(B * 100 / dx, 1), "unit": "impressions/s", "final_loss": round(float(lx), 5)}
                del tsx
                torch.cuda.empty_cache()
            if not args.no_bert:
                r = bert_naml_bench.run(batch=B, steps=5, warmup=2, layers=12, hidden=D, tune_from=0)
                ssec["bert_naml_base_tune_from_0"] = {"steps": 5, "warmup": 2, "ms_per_step": round(r["s_per_step"] * 1e3, 2), "value": r["impressions_per_s"],
                                                      "unit": "impressions/s", "final_loss": round(r["loss"], 4),
                                                      "kernels_tflops_equivalent": {k: v["tflops"] for k, v in r["kernels"].items()}}
                Env.set_lm_cache(False)
                torch.cuda.empty_cache()
            sec["split_bf16_opt_in"] = ssec
        finally:
            _L.set_product_mode(_L.EXACT_F32)
        extra["secondary"] = sec

    if world_size == 1 and not dist_on and not args.no_dist_check and not args.small:
        extra["dist_path_check"] = dist_path_check(make_ts, args, data, dev, barrier, B)
    if rank != 0:
        if dist_on:
            torch.distributed.destroy_process_group()
        return
    from legommenders_amd import _lib as _L0
    _lib_mode = _L0.product_mode()                 # 0 unless the whole run was started with LEGO_SPLIT_BF16=1
    out = {
        "metric": "train impressions/sec on MIND-small NAML" if args.model == "naml" else "train impressions/sec on MIND-small NRMS",
        "value": round(B * world_size * args.steps / dt, 1), "unit": "impressions/s",
        "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(host_ms, 4),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if _lib_mode == 0 else "split-bf16 (LEGO_SPLIT_BF16=1: opt-in product mode, NOT the parity mode)", "data": "synthetic",
        "config": {"workload": f"MIND-small-shaped {args.model.upper()} hidden={D} bs={B}/GPU GloVe(300d frozen) "
                               f"K=4 negatives S=50 T=30, full train step (sample+fwd+bwd+allreduce+Adam), dropout 0.1"
                               + (" [SMALL WORLD - not the metric config]" if args.small else ""),
                   "global_batch": B * world_size, "parallelism": f"dp{world_size}",
                   "raggedness": "histories ~ clipped geometric (mean 20 of 50 slots), titles ~ U[5,30] tokens: pad rows are skipped",
                   "live_token_rows_per_step": round(rows_per_launch, 1),
                   "item_instances_per_step": round(inst_per_launch, 1),
                   "distinct_tokens_per_step": round(uniq_per_launch, 1) if dedup else None,
                   "projection": ("once per DISTINCT token of the batch, expanded to the token rows (exact: the frozen-table "
                                  "projection depends on the token id alone); LEGO_DEDUP=0 projects row by row") if dedup else "row by row"},
        "final_loss": round(final_loss, 5),
        "device_wakeup_ms": spin_ms, "prewarm_scratch_steps": prewarm, "time_every_effective": every,
        "roofline": roofline, "roofline_gather": roofline_gather, "roofline_step": step_roofline(args.model, flops, dt / args.steps, rows_per_launch if args.model == "naml" else
                                       {"rows": rows_per_launch, "uniq": uniq_per_launch, "glove": args.embed == "glove", "per_key": per_key_of(eng)}, D, E0, wino),
        "kernels": {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kern.items()},
    }
    out.update(extra)
    if cold is not None:
        out.update({"value_without_prewarm": cold["value_without_prewarm"], "without_prewarm": cold})
    out["fracs_over_one"] = fracs_over_one(out)    # a fraction of a peak above 1 is a pricing bug, not evidence (tests/test_bench_launcher.py)
    out.setdefault("ranks_seen", 1)
    assert out["ranks_seen"] == args.gpus == out["n_gpus"], "a rank is missing: this line would misreport the job"
    if not args.no_cpu_baseline and world_size == 1:
        out["cpu_baseline"] = cpu_baseline(world, B, D, args.cpu_steps)
    else:
        out["cpu_baseline"] = None
    _json_out.write(json.dumps(out) + "\n")
    _json_out.flush()
    if dist_on:
        torch.distributed.destroy_process_group()


def fracs_over_one(obj, path=""):
    """every `*frac*` entry of the line that exceeds 1 (none is the only acceptable answer).  One exemption, stated here: a Winograd launch is
    priced at the DIRECT conv's flops (the tier's rule: algorithmic work over time) and carries `mfma_issue_frac` = the 2/3 of them the
    matrix pipe is really asked for; on a pad-free batch the first can pass 1 while the second, the physical bound, cannot"""
    bad = []
    if isinstance(obj, dict):
        wino_ok = isinstance(obj.get("mfma_issue_frac"), (int, float)) and obj["mfma_issue_frac"] <= 1.0
        for k, v in obj.items():
            if wino_ok and k in ("frac", "frac_of_f32_mfma_peak"):
                continue
            if isinstance(v, (dict, list)):
                bad += fracs_over_one(v, f"{path}.{k}" if path else k)
            elif "frac" in k and isinstance(v, (int, float)) and v > 1.0:
                bad.append(f"{path}.{k}={v}")
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            bad += fracs_over_one(v, f"{path}[{i}]")
    return bad


def dist_path_check(make_ts, args, data, dev, barrier, B):
    """N = 1, after everything that is timed: the data-parallel step's own code path on a ONE-rank RCCL communicator -- communicator
    creation, the flat gradient buffer's all-reduce (timed alone, 20 launches) and 100 training steps with the all-reduce in them
    (`TrainStep(force_allreduce=True)`).  A path check, not a scaling number: one rank moves no bytes over xGMI.  A failure is recorded
    as text and does not take the metric line with it."""
    out = {"backend": "nccl (RCCL)", "world": 1}
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        pg = torch.distributed.group.WORLD
        ts = make_ts(args.model, data, force=True, embed=args.embed, pg=pg)
        for _ in range(10):
            ts.step()
        barrier()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            ts.step()
        barrier()
        dt = time.perf_counter() - t0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts.sync_gradients()
        barrier()
        a.record()
        for _ in range(20):
            ts.sync_gradients()
        b.record()
        torch.cuda.synchronize()
        out.update({"allreduce_ms": round(a.elapsed_time(b) / 20, 4), "allreduce_bytes": ts.fp.numel * 4,
                    "steps_with_allreduce": n, "ms_per_step_with_allreduce": round(dt / n * 1e3, 4),
                    "value_with_allreduce": round(B * n / dt, 1), "ok": True})
        del ts
        torch.distributed.destroy_process_group()
    except Exception as exc:                       # noqa: BLE001 -- recorded, the metric line stands
        out.update({"ok": False, "error": f"{type(exc).__name__}: {exc}"[:300]})
        try:
            if torch.distributed.is_initialized():
                torch.distributed.destroy_process_group()
        except Exception:
            pass
    return out


def step_roofline(model, flops, step_s, rows, D, E0, wino):
    """the whole step against the fp32 matrix peak: ALGORITHMIC flops of its products (every token row projected, the
    direct conv's 2*3*D*D per row) over the measured step time, and beside it the flops the kernels ISSUE (projection once per
    distinct token, Winograd F(2,3) = 2/3 of the direct conv)"""
    if model == "nrms":
        # reference products over the live sequence rows (attention_operator.py:46-59: in-projection, out-projection, Linear, additive
        # hidden layer; embedding_hub.py:95: the GloVe projection), forward + data gradient + weight gradient each (the projection has no
        # data gradient: frozen table); issued: the folded operator runs ONE product of K = D, N = A behind the attention core instead of
        # three, and the per-key forms run the projection / in-projection forward over the distinct keys (DESIGN.md sections 5, 11.4)
        r, u, glove, per_key = rows["rows"], rows["uniq"], rows["glove"], rows["per_key"]
        alg = {"in_proj": 3 * 2.0 * r * D * 3 * D, "out_proj": 3 * 2.0 * r * D * D, "linear": 3 * 2.0 * r * D * D, "additive": 3 * 2.0 * r * D * 256,
               "glove_proj": (2 * 2.0 * r * D * E0) if glove else 0.0}
        issued = {"in_proj": (2.0 * (u if per_key else r) + 2 * 2.0 * r) * D * 3 * D, "folded_tail": 3 * 2.0 * r * D * 256,
                  "glove_proj": (2 * 2.0 * u * D * E0) if glove else 0.0}
        a, i = sum(alg.values()), sum(issued.values())
        return {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F32_MFMA_TFLOPS, "achieved": round(a / step_s / 1e12, 2),
                "frac": round(a / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "algorithmic_flops_per_step": a, "issued_flops_per_step": i,
                "issued_frac": round(i / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "products": sorted(alg),
                "note": "item-side products only (the attention core is priced against HBM in `kernels`; the user side adds < 3 %)"}
    if not flops or model != "naml":
        return None
    alg = dict(flops)
    alg["proj_fwd"] = alg["proj_bwd_weight"] = 2.0 * rows * D * E0
    issued = {k: v * (WINO_ISSUE if k in wino else 1.0) for k, v in flops.items()}
    a, i = sum(alg.values()), sum(issued.values())
    return {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F32_MFMA_TFLOPS,
            "achieved": round(a / step_s / 1e12, 2), "frac": round(a / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "algorithmic_flops_per_step": a, "issued_flops_per_step": i, "issued_frac": round(i / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "products": sorted(alg), "note": "the big products only (the small user-side / category products add < 2 %); frac = algorithmic "
            "flops / step time / peak, issued_frac = what the matrix pipe is asked to do"}


def per_key_of(eng):
    """the engine runs the in-projection once per DISTINCT key of the batch (trainable table: qkv_dedup; GloVe: dropcorr)"""
    return bool(getattr(eng, "qkv_dedup", False) or getattr(eng, "dropcorr", False))


def nrms_flops(rows, D, E0, per_key=None):
    """flops per launch of the NRMS item-side products AS LAUNCHED (rows = live sequence rows incl. SEP / category; `per_key` = distinct
    keys of the batch when the tagged in-projection launch runs over those: its time must not be priced with the row count)"""
    return {"qkv_fwd_item": 2.0 * (rows if per_key is None else per_key) * D * 3 * D, "out_proj_fwd_item": 2.0 * rows * D * D, "linear_fwd_item": 2.0 * rows * D * D,
            "outlin_fwd_item": 2.0 * rows * D * D,       # out-projection and Linear folded into one product (engine.py, fold_linear)
            "additive_fwd_item": 2.0 * rows * D * 256,      # fold level 2: the only product over the rows between the core and the pool
            }


def naml_small_flops(hist_rows, D, A=256):
    """the user-side additive products (rows = clicked-item instances of the batch, ~1.3 k): latency-bound launches, priced all the same"""
    return {"additive_fwd_user": 2.0 * hist_rows * D * A, "additive_bwd_weight_user": 2.0 * hist_rows * D * A}


def naml_stream_bytes(rows, inst, uniq, D):
    """algorithmic HBM bytes of the streaming kernels of the de-duplicated projection: the expansion H[r] = drop(Hu[inv[r]]) reads the
    distinct rows (L2-resident), an index and a keep byte per 4 rows and writes one D-row per token row; the per-token sums read
    every dH row once (through the sort permutation) and write one row per distinct token"""
    return {"proj_expand": uniq * D * 4.0 + rows * (D * 4.0 + 4.0 + D / 4.0),
            "proj_bwd_segsum": rows * (D * 4.0 + 8.0) + uniq * D * 4.0}


def nrms_core_bytes(rows, D, heads=8, Lbar=21):
    """algorithmic HBM bytes per launch of the attention core (it is memory-bound: 2 * L * hd MACs per row and head against 12 KB of
    operands per (segment, head)): forward reads the Q / K / V rows and writes the output rows and the saved probabilities
    (~Lbar per row and head); backward reads Q / K / V, d(out) and the probabilities and writes d(qkv)"""
    qkv, o, pr = rows * 3 * D * 4.0, rows * D * 4.0, rows * heads * Lbar * 4.0
    return {"mhsa_core_fwd_item": qkv + o + pr, "mhsa_core_bwd_item": qkv + o + pr + qkv}


def price_hbm(kern, nbytes):
    for tag, b in nbytes.items():
        if tag in kern and kern[tag]["avg_ms"] > 0:
            gbs = b / (kern[tag]["avg_ms"] * 1e-3) / 1e9
            kern[tag].update({"bound": "hbm", "gb_per_s": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": b})


def price(kern, flops, wino):
    for tag, f in flops.items():
        if tag in kern and kern[tag]["avg_ms"] > 0:
            tf = f / (kern[tag]["avg_ms"] * 1e-3) / 1e12
            kern[tag]["tflops"] = tf
            kern[tag]["frac_of_f32_mfma_peak"] = tf / PEAK_F32_MFMA_TFLOPS
            if tag in wino:
                kern[tag]["mfma_issue_frac"] = tf * WINO_ISSUE / PEAK_F32_MFMA_TFLOPS


def dominant(kern, flops, solo, traffic):
    """the largest GEMM that runs ALONE on the GPU (forward, main stream): the backward products overlap with side-stream
    kernels, so their event brackets include time-sharing; they are listed in `kernels` with "overlapped": true"""
    mf = [(v["avg_ms"], k) for k, v in kern.items() if "tflops" in v and k in solo]
    if not mf:
        return None
    _, dom = max(mf)
    r = {"kernel": dom, "bound": "mfma", "achieved": round(kern[dom]["tflops"], 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": round(kern[dom]["tflops"] / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic.get(dom),
         "avg_launch_ms": round(kern[dom]["avg_ms"], 5), "launches": kern[dom].get("launches"), "timed": kern[dom].get("timed"),
         "algorithmic_flops_per_launch": flops[dom]}
    for extra in ("in_region", "every_kernel_bracketed"):
        if extra in kern[dom]:
            r[extra] = kern[dom][extra]
    if "mfma_issue_frac" in kern[dom]:
        r["mfma_issue_frac"] = round(kern[dom]["mfma_issue_frac"], 4)
        r["note"] = "frac prices the direct conv's 2*3*D*D flops per row; the Winograd F(2,3) kernel issues 2/3 of them (mfma_issue_frac)"
    return r


if __name__ == "__main__":
    main()
