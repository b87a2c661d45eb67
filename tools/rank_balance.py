"""Predicted straggler cost of data-parallel dealing, from the tables alone (CPU; no GPU needed).

A step of W ranks costs the SLOWEST rank's step; a rank's step time is linear in its live rows (token rows + item
instances; profiles/r02_step_profile.txt: 0.62-0.78 ms over the batches of one epoch).  For every global batch of W*B rows of
the bench world this prints max/mean of the per-rank live rows for the blind `r::W` dealing and for the cost-sorted snake
(`DeviceData(balance=B)`), and the weak-scaling efficiency each implies under `t = t0 + k * rows`.

    python tools/rank_balance.py [--world 8] [--batch 64] [--mind-like]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd.synthetic import MIND_SMALL, make_world          # noqa: E402
from legommenders_amd.train_step import deal_balanced, row_cost       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--mind-like", action="store_true")
ap.add_argument("--fixed-ms", type=float, default=0.26, help="row-independent part of a step (user-side neck, boundaries, Adam)")
ap.add_argument("--ms-per-krow", type=float, default=0.0145, help="step ms per 1000 live rows (fit of r02_step_profile)")
args = ap.parse_args()
W, B = args.world, args.batch
world = make_world(seed=2023, **MIND_SMALL)
if args.mind_like:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    world = bench.mind_like_world(world)
cost = row_cost(world)
n = len(cost) // (W * B) * (W * B)
perm = torch.randperm(len(cost), generator=torch.Generator().manual_seed(2023))[:n]
rs = np.random.RandomState(0)
neg = torch.from_numpy((world["title_len"].astype(np.int64) + 1)[rs.randint(0, world["n_items"], size=(len(cost), 4))].sum(1))
full = cost + neg                                   # what the step really sees: history + positive + 4 sampled negatives


def per_rank_rows(rows_of_rank):
    return torch.stack([full[r].view(-1, B).sum(1) for r in rows_of_rank], 1).double()      # [steps, W]


blind = per_rank_rows([perm[r::W] for r in range(W)])
snake = per_rank_rows([deal_balanced(perm, cost, W, r, B)[0] for r in range(W)])
for name, m in (("r::W", blind), ("snake", snake)):
    ratio = (m.max(1).values / m.mean(1))
    t_rank = args.fixed_ms + args.ms_per_krow * m / 1e3
    eff = (t_rank.mean(1) / t_rank.max(1).values).mean().item()
    print(f"{name:6s} W={W} B={B}: live rows per rank-step mean {m.mean():.0f}, max/mean over ranks: mean {ratio.mean():.4f} "
          f"p95 {ratio.quantile(0.95):.4f} worst {ratio.max():.4f}; predicted step-time efficiency vs a balanced step {eff:.4f}")
