#!/usr/bin/env python3
"""The REFERENCE itself timed on CPU (VERDICT r3 missing #4): `/root/reference`'s own `Legommender`
forward + `loss.backward()` + `torch.optim.Adam.step()` (trainer.py:190-204) on MIND-small-shaped synthetic batches, for the three
CPU-runnable BASELINE configurations, >= 20 steps after 3 warm-ups, `torch.set_num_threads(8)` (this container has 8 cores).

Runs ONLY in the build container (imports /root/reference with the two in-memory stubs of SURVEY.md Appendix B, through
tests/golden/make_golden.py); writes profiles/r04_reference_cpu.json.  The input pipeline (DataSet + Resampler + collate, 0 workers)
is timed separately: the model-only figure is the one comparable with bench.py's `cpu_baseline` (the oracle's port of the same step).

    python tools/time_reference_cpu.py [--steps 20] [--threads 8]
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_train_band as M                                            # noqa: E402  (stubs, duck-typed tables, Manager-order build)
from legommenders_amd.synthetic import MIND_SMALL, make_world           # noqa: E402

CONFIGS = [
    ("config1_naml_hidden64_bs32_glove", "naml", dict(D=64, B=32), True),
    ("config2_naml_hidden256_bs64_glove", "naml", dict(D=256, B=64), True),
    ("config3_nrms_hidden256_bs64_null", "nrms", dict(D=256, B=64), False),       # trainable 400k x 256 token table (SURVEY.md section 6 probe)
    ("config3_nrms_hidden256_bs64_glove", "nrms", dict(D=256, B=64), True),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    M.MG.install_stubs()
    torch.set_num_threads(args.threads)
    from loader.data_set import DataSet
    from loader.env import Env
    from torch.utils.data import DataLoader
    cfg = dict(MIND_SMALL)
    need = 64 * (args.steps + args.warmup + 2)
    cfg["n_rows"] = need                              # the train table only has to hold the rows that are timed
    w = make_world(seed=2023, **cfg)
    w["valid"] = dict(user=w["row_user"][:8], item=w["row_item"][:8], label=np.ones(8, dtype=np.int64))
    out = {"torch": torch.__version__, "threads": args.threads, "host_cores": os.cpu_count(), "steps": args.steps, "warmup": args.warmup,
           "world": {k: v for k, v in cfg.items()}, "data": "synthetic MIND-small-shaped (legommenders_amd.synthetic.make_world, seed 2023)",
           "what": "the reference's own Legommender.forward + backward + Adam (dropout on), DataLoader(shuffle=True, num_workers=0)",
           "configs": {}}
    for name, kind, hyper, pretrained in CONFIGS:
        if args.only and args.only not in name:
            continue
        M.HYPER.update(hyper)
        random.seed(2023); np.random.seed(2023); torch.manual_seed(2023)
        t0 = time.time()
        model, resampler, train_ut, _ = M.build(kind, w, 2023, pretrained=pretrained)
        build_s = time.time() - t0
        B = hyper["B"]
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3)
        n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        loader = DataLoader(DataSet(train_ut, resampler), batch_size=B, shuffle=True)
        model.train(); Env.train()
        it = iter(loader)
        t_load = t_fwd = t_bwd = t_opt = 0.0
        n = 0
        for step in range(args.steps + args.warmup):
            a = time.perf_counter()
            batch = next(it)
            b = time.perf_counter()
            loss = model(batch=batch)
            c = time.perf_counter()
            loss.backward()
            d = time.perf_counter()
            opt.step(); opt.zero_grad()
            e = time.perf_counter()
            if step >= args.warmup:
                t_load += b - a; t_fwd += c - b; t_bwd += d - c; t_opt += e - d
                n += 1
        model_s = (t_fwd + t_bwd + t_opt) / n
        out["configs"][name] = {
            "kind": kind, "hidden": hyper["D"], "batch": B, "embed": "glove (frozen 400k x 300 + Linear)" if pretrained else "null (trainable 400k x hidden)",
            "trainable_params": n_params, "steps_timed": n,
            "model_only": {"impressions_per_s": round(B / model_s, 2), "s_per_step": round(model_s, 4), "fwd_s": round(t_fwd / n, 4),
                           "bwd_s": round(t_bwd / n, 4), "adam_s": round(t_opt / n, 4)},
            "input_pipeline_only": {"impressions_per_s": round(B * n / t_load, 1), "s_per_batch": round(t_load / n, 4)},
            "end_to_end_impressions_per_s": round(B * n / (t_load + t_fwd + t_bwd + t_opt), 2), "final_loss": round(float(loss), 4),
            "build_s": round(build_s, 1)}
        print(name, json.dumps(out["configs"][name]), flush=True)
        del model, opt, loader, it
    path = os.path.join(ROOT, "profiles", "r04_reference_cpu.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
