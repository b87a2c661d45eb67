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
import torch.multiprocessing as mp

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
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
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
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "dp.npz"))
    _, _, G, _, _, _, _ = load_model_fixture("naml_glove_d64")           # reference gradient of the FULL batch
    for k, g in G.items():
        scale = float(np.abs(g).max())
        assert float(np.abs(got[k] - g).max()) <= 2e-4 * scale + 1e-9, k


def test_shards_partition_the_permutation():
    from legommenders_amd.synthetic import make_world
    from legommenders_amd.train_step import DeviceData
    w = make_world(seed=3, n_items=50, n_users=40, n_rows=101, V=100)
    world = 4
    shards = [DeviceData(w, "cpu", rank=r, world_size=world, seed=11) for r in range(world)]
    sizes = [s.n_rows for s in shards]
    assert sum(sizes) == 101 and max(sizes) - min(sizes) <= 1
    pairs = set()
    for s in shards:
        pairs |= set(zip(s.row_user.tolist(), s.row_item.tolist(), range(10**6)))  # multiset via index below
    allrows = sorted(zip(w["row_user"].tolist(), w["row_item"].tolist()))
    got = sorted(sum([list(zip(s.row_user.tolist(), s.row_item.tolist())) for s in shards], []))
    assert got == allrows


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
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
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
    mp.spawn(_eval_shard_worker, args=(world, port, ""), nprocs=world, join=True)
