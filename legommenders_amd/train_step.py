"""One data-parallel training step of the hot path, all on device:

    rows of this rank's shard -> negative sampling + history fetch (device) -> ragged forward ->
    backward into ONE flat fp32 gradient buffer -> (N > 1) a single RCCL all-reduce of that buffer ->
    fused Adam over the flat parameter buffer.

Mirrors the reference loop `loss = legommender(batch); loss.backward(); optimizer.step();
scheduler.step(); optimizer.zero_grad()` (trainer.py:190-204) and its Adam / linear-schedule set-up
(base_lego.py:175-223).  The reference has no distributed layer; the sharding contract here is
"N ranks x per-rank batch B  ==  one device with batch N*B" (CrossEntropy mean over equal shards, so the
averaged per-rank gradients are the global-batch gradient).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from ._lib import call
from .engine import ItemTables, NamlEngine, NrmsEngine, _ptr, _stream, current_stream, shared_stream, stream_handle


class FlatParams:
    """Trainable tensors as views into one flat fp32 buffer (+ matching grad / Adam-moment buffers)."""

    def __init__(self, params: Dict[str, torch.Tensor], frozen=(), device="cuda", last=()):
        """`last`: names placed at the end of the flat buffers -- the gradients that finish last in backward, so that
        everything before `self.split` can be all-reduced while they are still being computed (TrainStep)."""
        self.names = [k for k in params if k not in frozen and k not in last] + [k for k in last if k in params and k not in frozen]
        sizes = [params[k].numel() for k in self.names]
        self.offsets, off = {}, 0
        for k, n in zip(self.names, sizes):
            self.offsets[k] = off
            off += (n + 3) // 4 * 4                 # keep every tensor 16-B aligned inside the buffer
        self.numel = off
        tail = [k for k in last if k in self.offsets]
        self.split = min(self.offsets[k] for k in tail) if tail else 0      # grad[:split] is complete before grad[split:]
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        for k in params:
            if k in frozen:
                self.P[k] = params[k].to(device=device, dtype=torch.float32).contiguous()
                continue
            o, n = self.offsets[k], params[k].numel()
            self.P[k] = self.flat[o:o + n].view(params[k].shape)
            self.P[k].copy_(params[k])
            self.G[k] = self.grad[o:o + n].view(params[k].shape)

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.P.items()}


def rank_seed(seed: int, rank: int) -> int:
    """seed of a rank-local Philox stream family (dropout keep bits: their counters are rank-local row indices, so the rank
    goes into the key; rank 0 keeps `seed`)"""
    # kept below 2^63: the plug-in route passes it as the `int seed` argument of torch.ops.lego_hip.* (an int64 schema)
    return (int(seed) ^ (0x9E3779B97F4A7C15 * int(rank))) & 0x7FFFFFFFFFFFFFFF


def row_cost(world: dict):
    """live rows a train row puts into a step, from the tables alone (host, int64 [n_rows]): token rows + one instance row
    per clicked item of the user's history, plus the positive candidate's (the K negatives are drawn later; their expected
    cost is the same for every row)"""
    import numpy as np
    tl = np.asarray(world["title_len"]).astype(np.int64) + 1                       # + the category row of the instance
    hist, hl = np.asarray(world["user_hist"]), np.asarray(world["user_hist_len"])
    live = np.arange(hist.shape[1])[None, :] < hl[:, None]
    user_cost = (tl[np.clip(hist, 0, len(tl) - 1)] * live).sum(1)
    return torch.from_numpy(user_cost[np.asarray(world["row_user"])] + tl[np.asarray(world["row_item"])])


def deal_balanced(perm: torch.Tensor, cost: torch.Tensor, W: int, r: int, B: int):
    """rows of every global batch (W*B consecutive entries of `perm`; the last one may be shorter, a multiple of W) dealt
    to W ranks by cost: descending order, snake over the ranks, so every rank gets the same COUNT and near-equal cost.
    Returns (rank r's rows in visiting order, the position of each inside its global batch)."""
    n = perm.numel()
    rows_out, pos_out = [], []
    G = W * B
    full = n // G
    snake = torch.cat([torch.arange(W), torch.arange(W - 1, -1, -1)])

    def one(block, width):          # block: [nb, width] row ids
        order = torch.argsort(cost[block], dim=1, descending=True, stable=True)             # positions inside the batch
        owner = snake[torch.arange(width) % (2 * W)]
        mine = order[:, owner == r]                                                          # [nb, width / W]
        return torch.gather(block, 1, mine).reshape(-1), mine.reshape(-1)

    if full:
        a, b = one(perm[: full * G].view(full, G), G)
        rows_out.append(a); pos_out.append(b)
    if n > full * G:
        a, b = one(perm[full * G:].view(1, n - full * G), n - full * G)
        rows_out.append(a); pos_out.append(b)
    return torch.cat(rows_out), torch.cat(pos_out)


class DeviceData:
    """Train rows + user tables resident in HBM (replaces the DataLoader worker pipeline,
    loader/manager.py:374-381, loader/data_set.py:61-85, loader/resampler.py:139-259).

    Row order = `DataLoader(shuffle=True)` (manager.py:374-381) iterated anew every epoch (trainer.py:186-190): epoch e
    walks ONE permutation drawn from (seed, e), identical on every rank; it is cut to a multiple of the world size and rank r
    takes positions r::world, so every rank has the same number of rows (and steps, and schedule length) and the
    global batch of step s is the contiguous slice perm[s*W*B : (s+1)*W*B] a single device with batch W*B would take.
    Two epochs are resident (buffer = epoch % 2): the batch after the last one of an epoch is sampled ahead of time.

    `balance=B` (W > 1): WHICH rows of a global batch a rank takes is free -- the exchanged gradient is a sum over the same
    W*B rows -- so instead of `r::W` the rows of every global batch are dealt by their live-row cost (`row_cost`: token rows
    + item instances of the row's click history and positive, known from the tables): sorted by cost, dealt in a snake
    (0..W-1, W-1..0, ...), B rows per rank.  A step costs the slowest rank's rows; the snake brings max/mean over the ranks
    from ~1.1 to ~1.00 (tools/rank_balance.py).  The sampler then reads each row's position in the global batch from a
    table (`positions`), so the negatives still are what one device with batch W*B draws."""

    def __init__(self, world: dict, device, rank=0, world_size=1, seed=2023, balance: Optional[int] = None):
        i32 = lambda a: torch.as_tensor(a).to(device=device, dtype=torch.int32).contiguous()
        self.device = device
        self.tables = ItemTables(world["title_tok"], world["title_len"], world["cat"], device)
        self.user_hist, self.user_hist_len = i32(world["user_hist"]), i32(world["user_hist_len"])
        self.neg_list, self.neg_len = i32(world["neg_list"]), i32(world["neg_len"])
        self.neg_cap = self.neg_list.shape[1]
        self.S = self.user_hist.shape[1]
        self.n_items = self.tables.n_items
        self.rank, self.world_size, self.seed = int(rank), int(world_size), int(seed)
        self._all_user, self._all_item = i32(world["row_user"]), i32(world["row_item"])
        self.n_total = self._all_user.numel()
        self.n_rows = self.n_total // self.world_size          # per rank, equal on every rank
        self.balance = int(balance) if balance and self.world_size > 1 else None
        if self.balance:
            self._row_cost = row_cost(world)                   # host int64 [n_total]
        self._buf = [dict(epoch=-1, user=None, item=None, pos=None), dict(epoch=-1, user=None, item=None, pos=None)]
        self.ensure_epoch(0)

    def epoch_permutation(self, epoch: int) -> torch.Tensor:
        """host permutation of ALL train rows for `epoch` (same on every rank: seeded by (seed, epoch) only)"""
        g = torch.Generator().manual_seed(self.seed + 1000003 * int(epoch))
        return torch.randperm(self.n_total, generator=g)

    def deal(self, epoch: int):
        """(this rank's row indices for `epoch` in visiting order, their positions in their global batches or None)"""
        W, r = self.world_size, self.rank
        perm = self.epoch_permutation(epoch)[: self.n_rows * W]
        if not self.balance:
            return perm[r::W], None
        return deal_balanced(perm, self._row_cost, W, r, self.balance)

    def shard_of(self, epoch: int) -> torch.Tensor:
        """this rank's row indices for `epoch`, in visiting order"""
        return self.deal(epoch)[0]

    def ensure_epoch(self, epoch: int):
        """make the rows of `epoch` resident (enqueued on the current stream; a no-op when they already are)"""
        b = self._buf[epoch % 2]
        if b["epoch"] != epoch:
            mine, pos = self.deal(epoch)
            mine = mine.to(self.device)
            if b["user"] is None:
                b["user"], b["item"] = self._all_user[mine].contiguous(), self._all_item[mine].contiguous()
                b["pos"] = None if pos is None else pos.to(device=self.device, dtype=torch.int32).contiguous()
            else:                                              # in place: sampling kernels hold these addresses
                torch.index_select(self._all_user, 0, mine, out=b["user"])
                torch.index_select(self._all_item, 0, mine, out=b["item"])
                if pos is not None:
                    b["pos"].copy_(pos.to(torch.int32), non_blocking=False)
            b["epoch"] = epoch
        return b

    def rows(self, epoch: int):
        b = self.ensure_epoch(epoch)
        return b["user"], b["item"]

    def positions(self, epoch: int):
        """per row of this rank's shard: its position in the global batch it belongs to (None: r + b * W, the sampler's default)"""
        return self.ensure_epoch(epoch)["pos"]

    # epoch-0 views (tests, tools): fresh tensors, never the resident double buffers the sampling kernels read
    @property
    def row_user(self):
        return self._all_user[self.shard_of(0).to(self.device)]

    @property
    def row_item(self):
        return self._all_item[self.shard_of(0).to(self.device)]


class BatchSchedule:
    """batch index -> (epoch, first row, rows in the batch).  `tail="keep"`: the last batch of an epoch is the short one the
    reference's DataLoader yields (drop_last=False, manager.py:374-381); `tail="drop"`: whole batches only (bench)."""

    def __init__(self, n_rows: int, B: int, tail: str = "keep"):
        assert tail in ("keep", "drop")
        self.n_rows, self.B, self.tail = int(n_rows), int(B), tail
        full = self.n_rows // self.B
        self.steps_per_epoch = max(1, full + (1 if tail == "keep" and self.n_rows % self.B else 0))

    def at(self, batch_idx: int):
        epoch, k = divmod(int(batch_idx), self.steps_per_epoch)
        start = k * self.B
        return epoch, start, max(1, min(self.B, self.n_rows - start))


class TrainStep:
    BUCKET_BYTES = 64 << 20      # gradient buffers above this are all-reduced as a train of buckets (trainable token table)

    def __init__(self, kind: str, params: Dict[str, torch.Tensor], data: DeviceData, B: int, K: int = 4,
                 lr: float = 1e-3, total_steps: int = 0, warmup: int = 0, seed: int = 2023, heads: int = 8,
                 glove: bool = True, process_group=None, world_size: int = 1, dropout: bool = True,
                 force_allreduce: bool = False, accumulate: int = 1, tail: str = "keep", rank: Optional[int] = None):
        dev = data.tables.title_tok.device
        if data.balance not in (None, int(B)):
            # the dealing and the sampler's position table are built for global batches of world * data.balance rows: with another
            # B the local batches would no longer line up with one global batch (rows sharing sampler streams) -- and nothing would say so
            raise ValueError(f"DeviceData(balance={data.balance}) deals global batches of {data.balance} rows per rank; "
                             f"the step was built with B={B}")
        self.data, self.B, self.C, self.K = data, B, K + 1, K
        self.rank = data.rank if rank is None else int(rank)
        frozen = ("embedding_vocab_table.glove.embedding.weight",) if "embedding_vocab_table.glove.embedding.weight" in params else ()
        # the big trainable token table (embed/null) goes LAST in the flat buffers: its gradient is the last thing backward
        # produces, and everything before `fp.split` can be on the wire while it is still being scattered
        last = tuple(k for k in ("embedding_vocab_table.glove.weight",) if k in params)
        self.fp = FlatParams(params, frozen, dev, last=last)
        pd = 0.1 if dropout else 0.0
        self.dropout = dropout
        # dropout streams are keyed on rank-local row counters, so the rank goes into their seed; the negative sampler is
        # keyed on the row's position in the GLOBAL batch instead (sample_batch) and takes the plain seed
        eseed = rank_seed(seed, self.rank)
        if kind == "naml":
            self.engine = NamlEngine(self.fp.P, data.tables, B, self.C, data.S, seed=eseed, p_proj=pd, p_conv=pd)
            self.engine.bind_grads(self.fp.G)              # fused user tower: forward writes its gradient partials
        elif kind == "nrms":
            self.engine = NrmsEngine(self.fp.P, data.tables, B, self.C, data.S, heads=heads, glove=glove, seed=eseed,
                                     p_proj=pd, p_att=pd)
        else:
            raise ValueError(f"unknown model kind {kind!r} (the HIP path covers naml and nrms)")
        self.engines = [self.engine]
        # trainable token table (embed/null: nn.Embedding(V, D), dense gradient + dense Adam in the reference,
        # loader/embedding_hub.py:325-335): it sits last in the flat buffers; its Adam step skips rows that have never had a
        # gradient -- bit-identical to the dense rule (g = m = v = 0 -> zero update), see lego_adam_step_rows
        self.table = None
        for k in last:
            o, (rows, width) = self.fp.offsets[k], params[k].shape
            assert o == self.fp.split and o + rows * width == self.fp.numel, "the token table must close the flat buffer"
            self.table = (o, rows, width)
            self.touched = torch.zeros(rows, dtype=torch.uint8, device=dev)
            self.engine.touched_rows = self.touched
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        # two batch / plan slots: step N+1 is sampled and planned on `pre` while step N computes
        self._cand = [torch.zeros(B, self.C, **i32) for _ in range(2)]
        self._hist = [torch.zeros(B, data.S, **i32) for _ in range(2)]
        self._hist_len = [torch.zeros(B, **i32) for _ in range(2)]
        self.cand, self.hist, self.hist_len = self._cand[0], self._hist[0], self._hist_len[0]
        self.prefetch = str(dev) != "cpu"
        if self.prefetch:
            self.engine.enable_plan_slots()
            self.pre = shared_stream(dev, "prefetch")
            self._ready = [torch.cuda.Event(), torch.cuda.Event()]
            self._go = torch.cuda.Event()
            self._neck = torch.cuda.Event()
            self._planned_step = -1
            self._ready_joined = -1
        self.lr, self.total_steps, self.warmup = lr, total_steps, warmup
        self.seed, self.step_idx = seed, 0
        self.schedule = BatchSchedule(data.n_rows, B, tail)
        self.steps_per_epoch = self.schedule.steps_per_epoch
        # `accumulate` batches per optimiser step (exp.policy.accumulate_batch, trainer.py:171,197-203): gradients of the
        # batch-mean losses ADD UP over the cycle (no 1/accumulate), then one all-reduce + Adam + scheduler step.
        # batch_idx counts batches (sampling, plan slots, dropout streams), step_idx optimiser steps (Adam bias, lr).
        self.accumulate, self._acc, self.batch_idx = max(1, int(accumulate)), 0, 0
        self.pg, self.world = process_group, world_size
        self.force_allreduce = force_allreduce
        self.overlap_exchange = True               # tests set it to False to exchange the whole buffer after the backward pass
        self.counter_sum = torch.zeros(8, dtype=torch.int64, device=dev)
        self._grad_clean = True                    # FlatParams allocates a zeroed gradient buffer

    def lr_at(self, step: int) -> float:
        """HF get_linear_schedule_with_warmup (base_lego.py:211-223); total_steps == 0 -> constant lr."""
        if self.total_steps <= 0:
            return self.lr
        if step < self.warmup:
            return self.lr * step / max(1, self.warmup)
        return self.lr * max(0.0, (self.total_steps - step) / max(1, self.total_steps - self.warmup))

    def sample_batch(self, batch_idx=None, slot=0, stream=None):
        """Resampler.rebuild on device for batch `batch_idx` into batch slot `slot`; returns the rows in the batch"""
        d = self.data
        batch_idx = self.batch_idx if batch_idx is None else batch_idx
        epoch, start, nb = self.schedule.at(batch_idx)
        row_user, row_item = d.rows(epoch)
        st = _stream() if stream is None else stream_handle(stream)
        ru, ri = _ptr(row_user, start), _ptr(row_item, start)
        pos = d.positions(epoch)
        call("lego_sample_negatives", ru, ri, _ptr(d.neg_list), _ptr(d.neg_len), d.neg_cap, nb, self.K, d.n_items,
             self.seed, batch_idx, d.rank, d.world_size, None if pos is None else _ptr(pos, start), _ptr(self._cand[slot]), st)
        call("lego_gather_history", ru, _ptr(d.user_hist), _ptr(d.user_hist_len), nb, d.S, _ptr(self._hist[slot]),
             _ptr(self._hist_len[slot]), st)
        return nb

    def _prefetch(self, batch_idx, go=None):
        """sample + plan batch `batch_idx` on the side stream `pre` (slot = batch_idx % 2)"""
        slot = batch_idx % 2
        if go is None:
            go = self._go
            go.record(current_stream())
        self.pre.wait_event(go)                    # the slot's previous user (batch_idx - 2) is complete by then
        nb = self.sample_batch(batch_idx, slot, self.pre)
        self.engine.plan_on(self.pre, slot, self._cand[slot], self._hist[slot], self._hist_len[slot], nb)
        if self.dropout and hasattr(self.engine, "prefetch_masks"):
            self.engine.prefetch_masks(self.pre, slot)     # engine.step == batch_idx here (one training forward per batch)
        with torch.cuda.stream(self.pre):          # row statistics of the planned batch (bench.py), off the main stream
            self.counter_sum += self.engine._slots[slot]["counters"]
        self._ready[slot].record(self.pre)
        self._planned_step = batch_idx

    # ---- overlapped exchange of the trainable token table (embed/null, 410 MB): the engine's backward calls these hooks
    # (engine.grad_hooks) -- dense part first, then one table bucket after another, each as an ASYNCHRONOUS all-reduce that
    # RCCL orders behind the work enqueued so far and runs on its own stream, i.e. beside the scatter of the next bucket
    def _exchange_hooks(self):
        if self.table is None or not (self.world > 1 or self.force_allreduce) or not torch.distributed.is_initialized():
            return None                                # (tests emulate ranks in one process and exchange by hand: no hooks)
        o, rows, width = self.table
        per = max(1, self.BUCKET_BYTES // (4 * width))
        self._works = []

        def dense_ready():
            self._works.append(torch.distributed.all_reduce(self.touched, op=torch.distributed.ReduceOp.MAX, group=self.pg, async_op=True))
            if o > 0:
                self._works.append(torch.distributed.all_reduce(self.fp.grad[:o], group=self.pg, async_op=True))

        def bucket_ready(lo, hi):
            self._works.append(torch.distributed.all_reduce(self.fp.grad[o + lo * width:o + hi * width], group=self.pg, async_op=True))
        return dense_ready, bucket_ready, per

    def sync_gradients(self):
        """the ONE gradient exchange of an optimiser step: all-reduce(sum) of the flat buffer (1/world is applied inside
        Adam).  A buffer above BUCKET_BYTES (the 410 MB trainable token table of embed/null) goes out as a train of
        asynchronous bucket all-reduces on RCCL's stream; when the backward ran with the exchange hooks (the step's own path)
        that train was started DURING the backward -- dense part at the join of the side streams, each table bucket behind
        its own scatter -- and only has to be waited for here."""
        if not (self.world > 1 or self.force_allreduce):
            return
        works = getattr(self, "_works", None)
        if works:
            for w in works:
                w.wait()
            self._works = []
            return
        if getattr(self, "table", None) is not None:         # a row touched on ANY rank has a gradient everywhere after the sum
            torch.distributed.all_reduce(self.touched, op=torch.distributed.ReduceOp.MAX, group=self.pg)
        g = self.fp.grad
        per = max(1, self.BUCKET_BYTES // 4)
        if g.numel() <= per:
            torch.distributed.all_reduce(g, group=self.pg)
            return
        works = [torch.distributed.all_reduce(g[o:o + per], group=self.pg, async_op=True) for o in range(0, g.numel(), per)]
        for w in works:
            w.wait()

    def step(self):
        """sample -> forward -> backward -> all-reduce -> Adam.  Returns the device loss tensor (no sync)."""
        loss, last_of_cycle = self.compute_gradients()
        if last_of_cycle:
            self.sync_gradients()
            self.apply_update()
        self._prepare_next()
        return loss

    def params_changed(self):
        """call after writing into the parameter views (`fp.P`) from outside a step: a prologue prepared from the old values is dropped"""
        if hasattr(self.engine, "_pre_step"):
            self.engine._pre_step = None

    def _prepare_next(self):
        """the parameter-dependent prologue of the NEXT batch's forward pass, here, behind Adam on the main stream (NamlEngine.pre_forward):
        the next step then starts without a side chain and without cross-stream waits.  The next batch's plan is complete by now (it was
        started at this step's neck on the prefetch stream); the backward pass's final join has already waited for it (`join_ev`)."""
        eng = self.engine
        if not (self.prefetch and hasattr(eng, "pre_forward")) or self._planned_step != self.batch_idx or eng._serial:
            return
        slot = self.batch_idx % 2
        if self._ready_joined != self.batch_idx:
            current_stream().wait_event(self._ready[slot])
            self._ready_joined = self.batch_idx
        eng.pre_forward(slot)

    def compute_gradients(self):
        """one batch: sample -> forward -> backward into the flat gradient buffer; returns (loss, optimiser step due)"""
        slot = self.batch_idx % 2 if self.prefetch else 0
        self.cand, self.hist, self.hist_len = self._cand[slot], self._hist[slot], self._hist_len[slot]
        last_of_cycle = self._acc + 1 == self.accumulate
        epoch, _, nb = self.schedule.at(self.batch_idx)
        self.data.ensure_epoch(epoch)
        if self.prefetch:
            self.data.ensure_epoch(self.schedule.at(self.batch_idx + 1)[0])   # on this stream, before the events below
            if self._planned_step != self.batch_idx:
                self._prefetch(self.batch_idx)
            if self._ready_joined != self.batch_idx:   # (else: the previous step's final join covered this batch's plan)
                current_stream().wait_event(self._ready[slot])
                self._ready_joined = self.batch_idx
            self.engine.use_slot(slot)
        else:
            self.sample_batch()
        self.engine.set_batch(nb)
        if self._acc == 0 and not self._grad_clean:
            self.fp.grad.zero_()
        go = neck = None
        if self.prefetch:
            go, neck = self._go, self._neck
            if getattr(self.engine, "_pre_step", None) != (self.engine.step, slot):
                go.record(current_stream())     # everything before this step's forward (a prepared forward pass forks nothing: no event)
            else:
                go = None
        _, loss = self.engine.forward(self.cand, self.hist, self.hist_len, training=True, planned=self.prefetch,
                                      fork_ev=go, neck_ev=neck)
        self.engine.grad_hooks = self._exchange_hooks() if (last_of_cycle and self.overlap_exchange) else None
        if self.prefetch and hasattr(self.engine, "pre_forward") and not self.engine._serial:
            # host order (round 6): the backward pass's launches first, THEN the next batch's prefetch chain (its ~23 small launches start on the
            # device where this step's item tower ended -- `neck` -- whenever the host gets to them), then the backward pass's final join of the
            # side stream, which also covers the next batch's plan: ONE wait on the main stream for both
            self.engine.backward(self.fp.G, defer_join=True)
            self._prefetch(self.batch_idx + 1, neck)
            self.engine.finish_backward(join_ev=self._ready[(self.batch_idx + 1) % 2])
            self._ready_joined = self.batch_idx + 1
        else:
            if self.prefetch:
                # next batch: starts where this step's item tower ends (at the head of this step's forward pass instead: no faster)
                self._prefetch(self.batch_idx + 1, neck)
            self.engine.backward(self.fp.G)
        self.engine.grad_hooks = None
        self.batch_idx += 1
        if not self.prefetch:
            self.counter_sum += self.engine.counters
        self._grad_clean = False
        self._acc = 0 if last_of_cycle else self._acc + 1      # else: gradients stay in the buffer for the cycle's next batch
        return loss, last_of_cycle

    def apply_update(self):
        """Adam + linear schedule over the flat buffers (gradient scaled by 1/world, then cleared)"""
        self.step_idx += 1
        fp, lr, st = self.fp, self.lr_at(self.step_idx - 1), _stream()
        dense = fp.numel if self.table is None else self.table[0]
        call("lego_adam_step", _ptr(fp.flat), _ptr(fp.grad), _ptr(fp.m), _ptr(fp.v), dense, lr, 0.9, 0.999, 1e-8,
             self.step_idx, 1.0 / self.world, 1, st)
        if self.table is not None:
            o, rows, width = self.table
            call("lego_adam_step_rows", _ptr(fp.flat, o), _ptr(fp.grad, o), _ptr(fp.m, o), _ptr(fp.v, o), rows, width,
                 _ptr(self.touched), lr, 0.9, 0.999, 1e-8, self.step_idx, 1.0 / self.world, 1, st)
        self._grad_clean = True                    # Adam cleared the gradient buffer as it consumed it

    # ---- optimiser / scheduler state as the reference checkpoints them (base_lego.py:257-267): a torch.optim.Adam state_dict and
    # a LambdaLR state_dict, so that `exp.load.model_only: false` works in both directions.  `order` = the trainable parameter
    # names in the order `filter(requires_grad, legommender.parameters())` yields them (base_lego.py:201-204: one parameter group)
    def optimizer_state(self, order=None):
        order = [k for k in (order or self.fp.names) if k in self.fp.offsets]
        fp = self.fp
        state = {}
        for i, k in enumerate(order):
            o, n, shape = fp.offsets[k], fp.P[k].numel(), fp.P[k].shape
            state[i] = {"step": torch.tensor(float(self.step_idx)), "exp_avg": fp.m[o:o + n].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": fp.v[o:o + n].view(shape).detach().cpu().clone()}
        group = {"lr": self.lr_at(self.step_idx), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "initial_lr": self.lr, "params": list(range(len(order)))}
        return {"state": state if self.step_idx > 0 else {}, "param_groups": [group], "lego_param_names": order}

    def load_optimizer_state(self, st, order=None):
        """a torch.optim.Adam state_dict (ours or the reference's).  The parameter order comes from the file when we wrote it,
        else from `order` (the model's trainable parameters in `parameters()` order)."""
        if "state" not in st or "param_groups" not in st:
            raise ValueError("optimizer state is not a torch.optim.Adam state_dict")
        names = st.get("lego_param_names") or [k for k in (order or self.fp.names) if k in self.fp.offsets]
        ids = [i for g in st["param_groups"] for i in g["params"]]
        if len(ids) != len(names):
            raise ValueError(f"optimizer state holds {len(ids)} parameters, the model trains {len(names)}")
        fp = self.fp
        fp.m.zero_(); fp.v.zero_()
        step = 0
        for i, k in zip(ids, names):
            e = st["state"].get(i)
            if e is None:
                continue
            o, n = fp.offsets[k], fp.P[k].numel()
            if e["exp_avg"].numel() != n:
                raise ValueError(f"optimizer state of {k}: {tuple(e['exp_avg'].shape)} does not match {tuple(fp.P[k].shape)}")
            fp.m[o:o + n].copy_(e["exp_avg"].reshape(-1)); fp.v[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(e["step"])))
        self.step_idx = step
        if self.table is not None:                  # rows with a non-zero moment have had a gradient
            o, rows, width = self.table
            mv = (fp.m[o:].view(rows, width) != 0).any(1) | (fp.v[o:].view(rows, width) != 0).any(1)
            self.touched.copy_(mv.to(torch.uint8))

    def scheduler_state(self):
        """torch.optim.lr_scheduler.LambdaLR.state_dict() of HF's get_linear_schedule_with_warmup (base_lego.py:211-223)"""
        return {"base_lrs": [self.lr], "last_epoch": self.step_idx, "_step_count": self.step_idx + 1, "verbose": False,
                "_get_lr_called_within_step": False, "_last_lr": [self.lr_at(self.step_idx)], "lr_lambdas": [None],
                "_is_initial": False}

    def load_scheduler_state(self, st):
        self.step_idx = int(st["last_epoch"])
