"""Data-parallel contract on CPU with gloo (world_size 2): per-rank shards are disjoint and cover the
permutation, and ONE all-reduce(sum) of the flat gradient buffer followed by the 1/world scale that
`lego_adam_step` applies equals the single-process gradient of the concatenated (global) batch.
The per-shard gradients come from the CPU oracle here (no GPU in this container); the same contract is
checked on the HIP engine itself in tests/test_hip_parity.py::test_dp_contract_on_device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
from tests.conftest import spawn_ranks

from tests.golden_util import load_model_fixture


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    from legommenders_amd.train_step import FlatParams
    from oracle import lego_oracle as O
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=__import__("datetime").timedelta(seconds=180))     # a lost rendezvous / dead peer fails in minutes
    meta, P, G, tables, batch, _, _ = load_model_fixture("naml_glove_d64")
    B = batch["cand"].shape[0]
    sl = slice(rank * B // world, (rank + 1) * B // world)               # equal shards of the global batch
    _, _, g = O.loss_and_grads("naml", P, tables, batch["cand"][sl], batch["hist"][sl], batch["hist_len"][sl])
    fp = FlatParams({k: torch.tensor(v) for k, v in P.items()},
                    frozen=("embedding_vocab_table.glove.embedding.weight",), device="cpu")
    for k, v in g.items():
        fp.G[k].copy_(torch.tensor(v))
    dist.all_reduce(fp.grad)                                              # the ONE collective of a step
    fp.grad.mul_(1.0 / world)                                             # == grad_scale inside lego_adam_step
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), **{k: v.numpy() for k, v in fp.G.items()})
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_global_batch(tmp_path):
    world, port = 2, _free_port()
    spawn_ranks(_worker, (world, port, str(tmp_path)), world)
    got = np.load(os.path.join(str(tmp_path), "dp.npz"))
    _, _, G, _, _, _, _ = load_model_fixture("naml_glove_d64")           # reference gradient of the FULL batch
    for k, g in G.items():
        scale = float(np.abs(g).max())
        assert float(np.abs(got[k] - g).max()) <= 2e-4 * scale + 1e-9, k


def test_shards_are_equal_and_partition_the_epoch_permutation():
    """every rank owns the same number of rows (unequal shards = unequal step counts = collective mismatch);
    the union of the shards is the epoch's permutation cut to a multiple of the world size"""
    from legommenders_amd.synthetic import make_world
    from legommenders_amd.train_step import BatchSchedule, DeviceData
    w = make_world(seed=3, n_items=50, n_users=40, n_rows=101, V=100)
    world = 4
    shards = [DeviceData(w, "cpu", rank=r, world_size=world, seed=11) for r in range(world)]
    assert {s.n_rows for s in shards} == {101 // world}
    assert len({BatchSchedule(s.n_rows, 13, "keep").steps_per_epoch for s in shards}) == 1       # N=101, W=4, B=13 (ADVICE)
    for epoch in (0, 1, 5):
        perm = shards[0].epoch_permutation(epoch)
        assert all(torch.equal(perm, s.epoch_permutation(epoch)) for s in shards)                # same on every rank
        kept = perm[: (101 // world) * world]
        idx = [s.shard_of(epoch) for s in shards]
        inter = torch.stack(idx, 1).reshape(-1)              # rank r holds positions r::world
        assert torch.equal(inter, kept)
        for s, i in zip(shards, idx):
            ru, ri = s.rows(epoch)
            assert ru.tolist() == w["row_user"][i.numpy()].tolist() and ri.tolist() == w["row_item"][i.numpy()].tolist()
    # reshuffle: consecutive epochs visit the rows in different orders, and an epoch covers every kept row once
    p0, p1 = shards[0].epoch_permutation(0), shards[0].epoch_permutation(1)
    assert not torch.equal(p0, p1) and sorted(p0.tolist()) == sorted(p1.tolist()) == list(range(101))
    # the global batch of step s is what ONE device with batch W*B takes: perm[s*W*B:(s+1)*W*B]
    one = DeviceData(w, "cpu", rank=0, world_size=1, seed=11)
    B = 5
    for s_ in range(3):
        glob = one.shard_of(0)[s_ * world * B:(s_ + 1) * world * B]
        parts = torch.stack([s.shard_of(0)[s_ * B:(s_ + 1) * B] for s in shards], 1).reshape(-1)   # position b*W + r
        assert torch.equal(parts, glob)


def test_batch_schedule_keeps_the_short_last_batch():
    """DataLoader(shuffle=True, drop_last=False) (manager.py:374-381): ceil(n/B) batches per epoch, the last one short"""
    from legommenders_amd.train_step import BatchSchedule
    s = BatchSchedule(25, 8, "keep")
    assert s.steps_per_epoch == 4
    assert [s.at(i) for i in range(9)] == [(0, 0, 8), (0, 8, 8), (0, 16, 8), (0, 24, 1), (1, 0, 8), (1, 8, 8), (1, 16, 8),
                                          (1, 24, 1), (2, 0, 8)]
    d = BatchSchedule(25, 8, "drop")
    assert d.steps_per_epoch == 3 and d.at(3) == (1, 0, 8)
    assert BatchSchedule(24, 8, "keep").steps_per_epoch == 3
    assert BatchSchedule(5, 8, "keep").at(0) == (0, 0, 5)


def _sync_worker(rank, world, port, out_dir):
    """TrainStep.sync_gradients itself (not a copy of it) over gloo, single buffer and bucket train"""
    from legommenders_amd.train_step import FlatParams, TrainStep
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=__import__("datetime").timedelta(seconds=180))     # a lost rendezvous / dead peer fails in minutes
    P = {"a": torch.zeros(7, 5), "b": torch.zeros(33), "embedding_vocab_table.glove.weight": torch.zeros(50, 8)}
    for buckets in (1 << 30, 64 * 4):              # one all-reduce; a train of 64-float buckets
        ts = TrainStep.__new__(TrainStep)
        ts.fp = FlatParams(P, (), "cpu", last=("embedding_vocab_table.glove.weight",))
        ts.pg, ts.world, ts.force_allreduce, ts.BUCKET_BYTES = dist.group.WORLD, world, False, buckets
        g = torch.Generator().manual_seed(100 + rank)
        ts.fp.grad.copy_(torch.randn(ts.fp.numel, generator=g))
        ts.sync_gradients()
        want = sum(torch.randn(ts.fp.numel, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
        assert torch.allclose(ts.fp.grad, want, atol=1e-6), (rank, buckets)
        assert ts.fp.split == ts.fp.offsets["embedding_vocab_table.glove.weight"] > 0      # the table is last in the buffer
    # the OVERLAPPED form (engine.grad_hooks): dense part when the dense gradients are final, then one asynchronous all-reduce
    # per table bucket as the backward produces it; sync_gradients only waits.  Must equal the serial exchange.
    ts = TrainStep.__new__(TrainStep)
    ts.fp = FlatParams(P, (), "cpu", last=("embedding_vocab_table.glove.weight",))
    ts.pg, ts.world, ts.force_allreduce, ts.BUCKET_BYTES = dist.group.WORLD, world, False, 16 * 8 * 4      # 16 table rows per bucket
    o = ts.fp.offsets["embedding_vocab_table.glove.weight"]
    ts.table, ts.touched = (o, 50, 8), torch.zeros(50, dtype=torch.uint8)
    ts.touched[rank::3] = 1
    g = torch.Generator().manual_seed(200 + rank)
    full = torch.randn(ts.fp.numel, generator=g)
    dense_ready, bucket_ready, per = ts._exchange_hooks()
    assert per == 16
    ts.fp.grad[:o].copy_(full[:o])                   # the dense gradients are final ...
    dense_ready()
    for lo in range(0, 50, per):                      # ... then the table gradient arrives bucket by bucket
        hi = min(50, lo + per)
        ts.fp.grad[o + lo * 8:o + hi * 8].copy_(full[o + lo * 8:o + hi * 8])
        bucket_ready(lo, hi)
    assert len(ts._works) == 2 + 4
    ts.sync_gradients()
    assert ts._works == []
    want = sum(torch.randn(ts.fp.numel, generator=torch.Generator().manual_seed(200 + r)) for r in range(world))
    assert torch.allclose(ts.fp.grad, want, atol=1e-6)
    flags = torch.zeros(50, dtype=torch.uint8)
    for r in range(world):
        flags[r::3] = 1
    assert torch.equal(ts.touched, flags)
    dist.destroy_process_group()


def test_train_step_sync_gradients_over_gloo():
    world, port = 2, _free_port()
    spawn_ranks(_sync_worker, (world, port, ""), world)


def test_linear_schedule_matches_oracle():
    from legommenders_amd.train_step import TrainStep
    from oracle import lego_oracle as O
    ts = TrainStep.__new__(TrainStep)
    ts.lr, ts.total_steps, ts.warmup = 1e-3, 10, 0
    for step in range(12):
        assert abs(ts.lr_at(step) - 1e-3 * O.linear_schedule_factor(step, 10)) < 1e-12
    ts.total_steps, ts.warmup = 20, 5
    for step in range(22):
        assert abs(ts.lr_at(step) - 1e-3 * O.linear_schedule_factor(step, 20, 5)) < 1e-12


def _eval_shard_worker(rank, world, port, out_dir):
    from legommenders_amd.evaluate import gather_shards, shard_bounds
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=__import__("datetime").timedelta(seconds=180))     # a lost rendezvous / dead peer fails in minutes
    n, D = 11, 3                                                          # 11 rows over 2 ranks: shards of 6 and 5
    full = torch.arange(n * D, dtype=torch.float32).view(n, D)
    lo, hi, per = shard_bounds(n, rank, world)
    local = torch.zeros(per, D)
    local[:hi - lo] = full[lo:hi]                                         # what this rank "encoded"
    got = gather_shards(local, n, dist.group.WORLD, world)
    assert torch.equal(got, full), (rank, got)
    dist.destroy_process_group()


def test_eval_cache_shards_gather_to_the_whole_table():
    """evaluation caches: contiguous per-rank shards + one all_gather == the whole table on every rank"""
    from legommenders_amd.evaluate import shard_bounds
    for n in (0, 1, 7, 64, 65238, 91935):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(hi - lo <= per for lo, hi, per in b) and len({per for _, _, per in b}) == 1
    world, port = 2, _free_port()
    spawn_ranks(_eval_shard_worker, (world, port, ""), world)


def test_optimizer_and_scheduler_state_are_torch_state_dicts():
    """the engine route's checkpoint blobs load into torch.optim.Adam / LambdaLR (base_lego.py:251-253,257-267) and back"""
    from legommenders_amd.train_step import FlatParams, TrainStep
    P = {"w": torch.randn(4, 3), "b": torch.randn(5), "frozen": torch.randn(2, 2)}
    ts = TrainStep.__new__(TrainStep)
    ts.fp = FlatParams(P, ("frozen",), "cpu")
    ts.lr, ts.total_steps, ts.warmup, ts.step_idx, ts.table = 1e-3, 100, 10, 7, None
    ts.fp.m.copy_(torch.randn(ts.fp.numel)); ts.fp.v.copy_(torch.rand(ts.fp.numel))
    order = ["b", "w"]                                        # parameters() order of some model
    sd = ts.optimizer_state(order)
    params = [torch.nn.Parameter(P[k].clone()) for k in order]
    opt = torch.optim.Adam(params, lr=1e-3)
    opt.load_state_dict(sd)                                    # torch accepts it
    for i, k in enumerate(order):
        o, n = ts.fp.offsets[k], P[k].numel()
        assert torch.equal(opt.state[params[i]]["exp_avg"].reshape(-1), ts.fp.m[o:o + n])
        assert int(opt.state[params[i]]["step"]) == 7
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    sched.load_state_dict(ts.scheduler_state())
    assert sched.last_epoch == 7
    # ... and a state_dict written by torch (the reference's checkpoint) loads into the flat buffers
    ref = opt.state_dict()
    ts2 = TrainStep.__new__(TrainStep)
    ts2.fp = FlatParams(P, ("frozen",), "cpu")
    ts2.lr, ts2.total_steps, ts2.warmup, ts2.step_idx, ts2.table = 1e-3, 100, 10, 0, None
    ts2.load_optimizer_state(ref, order)
    ts2.load_scheduler_state(sched.state_dict())
    for k in order:                                            # (the alignment padding between tensors is not state)
        o, n = ts.fp.offsets[k], P[k].numel()
        assert torch.equal(ts2.fp.m[o:o + n], ts.fp.m[o:o + n]) and torch.equal(ts2.fp.v[o:o + n], ts.fp.v[o:o + n])
    assert ts2.step_idx == 7


def test_balanced_dealing_partitions_every_global_batch():
    """DeviceData(balance=B): the W ranks' rows of step s are still exactly the global batch perm[s*W*B:(s+1)*W*B] (so the
    summed gradient is the one-device gradient), B per rank, each row's recorded position is where it sits in that batch,
    and the per-rank live-row cost is closer to equal than under r::W"""
    from legommenders_amd.synthetic import make_world
    from legommenders_amd.train_step import BatchSchedule, DeviceData, row_cost
    w = make_world(seed=3, n_items=80, n_users=60, n_rows=203, V=100)
    W, B = 4, 6
    ranks = [DeviceData(w, "cpu", rank=r, world_size=W, seed=11, balance=B) for r in range(W)]
    blind = [DeviceData(w, "cpu", rank=r, world_size=W, seed=11) for r in range(W)]
    one = DeviceData(w, "cpu", seed=11, balance=B)
    assert one.balance is None and one.positions(0) is None                     # W = 1: nothing to deal
    cost = row_cost(w)
    sched = BatchSchedule(ranks[0].n_rows, B, "keep")
    assert sched.steps_per_epoch == 9 and ranks[0].n_rows == 50                 # 8 full batches + a short one of 2 rows per rank
    spread_bal, spread_blind = [], []
    for epoch in (0, 3):
        perm = one.epoch_permutation(epoch)[: 50 * W]
        for s in range(sched.steps_per_epoch):
            _, start, nb = sched.at(s)
            glob = perm[start * W:(start + nb) * W]
            rows = [d.shard_of(epoch)[start:start + nb] for d in ranks]
            pos = [d.positions(epoch)[start:start + nb].long() for d in ranks]
            assert sorted(torch.cat(rows).tolist()) == sorted(glob.tolist())
            assert sorted(torch.cat(pos).tolist()) == list(range(nb * W))
            for r_, p_ in zip(rows, pos):
                assert torch.equal(glob[p_], r_)
            for d, r_ in zip(ranks, rows):
                ru, ri = d.rows(epoch)
                assert ru[start:start + nb].tolist() == w["row_user"][r_.numpy()].tolist()
                assert ri[start:start + nb].tolist() == w["row_item"][r_.numpy()].tolist()
            if nb == B:
                per = torch.stack([cost[r_].sum() for r_ in rows]).double()
                spread_bal.append(float(per.max() / per.mean()))
                per = torch.stack([cost[d.shard_of(epoch)[start:start + nb]].sum() for d in blind]).double()
                spread_blind.append(float(per.max() / per.mean()))
    assert np.mean(spread_bal) < 1.0 + 0.5 * (np.mean(spread_blind) - 1.0)


def test_rank_seed_fits_a_custom_op_int():
    """the seed goes through torch.ops.lego_hip.* `int seed` arguments (int64) on the plug-in route"""
    from legommenders_amd.train_step import rank_seed
    seeds = [rank_seed(2023, r) for r in range(16)]
    assert all(0 <= s < 2 ** 63 for s in seeds) and len(set(seeds)) == 16 and seeds[0] == 2023
