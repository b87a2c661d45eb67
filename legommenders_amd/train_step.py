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
from .engine import ItemTables, NamlEngine, NrmsEngine, _ptr, _stream


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


class DeviceData:
    """Train rows + user tables resident in HBM (replaces the DataLoader worker pipeline,
    loader/manager.py:374-381, loader/data_set.py:61-85, loader/resampler.py:139-259)."""

    def __init__(self, world: dict, device, rank=0, world_size=1, seed=2023):
        i32 = lambda a: torch.as_tensor(a).to(device=device, dtype=torch.int32).contiguous()
        self.tables = ItemTables(world["title_tok"], world["title_len"], world["cat"], device)
        self.user_hist, self.user_hist_len = i32(world["user_hist"]), i32(world["user_hist_len"])
        self.neg_list, self.neg_len = i32(world["neg_list"]), i32(world["neg_len"])
        self.neg_cap = self.neg_list.shape[1]
        self.S = self.user_hist.shape[1]
        self.n_items = self.tables.n_items
        # one shared seeded permutation per epoch; rank r takes rows r::world
        g = torch.Generator().manual_seed(seed)
        perm = torch.randperm(len(world["row_user"]), generator=g)
        mine = perm[rank::world_size]
        self.row_user = i32(world["row_user"])[mine.to(device)].contiguous()
        self.row_item = i32(world["row_item"])[mine.to(device)].contiguous()
        self.n_rows = self.row_user.numel()


class TrainStep:
    def __init__(self, kind: str, params: Dict[str, torch.Tensor], data: DeviceData, B: int, K: int = 4,
                 lr: float = 1e-3, total_steps: int = 0, warmup: int = 0, seed: int = 2023, heads: int = 8,
                 glove: bool = True, process_group=None, world_size: int = 1, dropout: bool = True, micro: int = 1,
                 force_allreduce: bool = False, accumulate: int = 1):
        dev = data.tables.title_tok.device
        self.data, self.B, self.C, self.K = data, B, K + 1, K
        frozen = ("embedding_vocab_table.glove.embedding.weight",) if "embedding_vocab_table.glove.embedding.weight" in params else ()
        # the projection weight gradient is the last kernel of backward: with LEGO_AR_BUCKETS=2 it is all-reduced on its
        # own after the rest (which then overlaps that GEMM).  Off by default: at world size 1 under torchrun the second
        # collective's stream hops cost more than the overlap can save (0.851 vs 0.830 ms/step); to be re-measured on 8 GPUs.
        last = ("embedding_vocab_table.glove.linear.weight",) if (kind == "naml" and micro == 1) else ()
        self.fp = FlatParams(params, frozen, dev, last=last)
        pd = 0.1 if dropout else 0.0
        self.dropout = dropout
        # `micro` > 1: the batch is processed as `micro` equal micro-batches on their own HIP streams.  Their kernel
        # chains are independent, so one chain's prologue / epilogue / last-wave tail overlaps the other's MFMA main
        # loops; gradients of all micro-batches accumulate (atomics) into the same flat buffer == the full-batch mean.
        assert B % micro == 0, "batch must be divisible by the number of micro-batches"
        self.micro, Bm = micro, B // micro
        self.engines = []
        for i in range(micro):
            if kind == "naml":
                e = NamlEngine(self.fp.P, data.tables, Bm, self.C, data.S, seed=seed + 7919 * i, p_proj=pd, p_conv=pd)
            elif kind == "nrms":
                e = NrmsEngine(self.fp.P, data.tables, Bm, self.C, data.S, heads=heads, glove=glove, seed=seed + 7919 * i,
                               p_proj=pd, p_att=pd)
            else:
                raise ValueError(f"unknown model kind {kind!r} (the HIP path covers naml and nrms)")
            self.engines.append(e)
        if micro == 1 and kind == "naml":
            self.engines[0].bind_grads(self.fp.G)          # fused user tower: forward writes its gradient partials
        self.engine = self.engines[0]
        self.streams = [torch.cuda.Stream(dev) for _ in range(micro)] if micro > 1 else [None]
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        # two batch / plan slots: step N+1 is sampled and planned on `pre` while step N computes
        self._cand = [torch.zeros(B, self.C, **i32) for _ in range(2)]
        self._hist = [torch.zeros(B, data.S, **i32) for _ in range(2)]
        self._hist_len = [torch.zeros(B, **i32) for _ in range(2)]
        self.cand, self.hist, self.hist_len = self._cand[0], self._hist[0], self._hist_len[0]
        self.prefetch = micro == 1 and str(dev) != "cpu"
        if self.prefetch:
            self.engine.enable_plan_slots()
            self.pre = torch.cuda.Stream(dev)
            self._ready = [torch.cuda.Event(), torch.cuda.Event()]
            self._go = torch.cuda.Event()
            self._neck = torch.cuda.Event()
            self._planned_step = -1
        self.lr, self.total_steps, self.warmup = lr, total_steps, warmup
        self.seed, self.step_idx = seed, 0
        # `accumulate` batches per optimiser step (exp.policy.accumulate_batch, trainer.py:171,197-203): gradients of the
        # batch-mean losses ADD UP over the cycle (no 1/accumulate), then one all-reduce + Adam + scheduler step.
        # batch_idx counts batches (sampling, plan slots, dropout streams), step_idx optimiser steps (Adam bias, lr).
        self.accumulate, self._acc, self.batch_idx = max(1, int(accumulate)), 0, 0
        self.pg, self.world = process_group, world_size
        self.force_allreduce = force_allreduce
        self.counter_sum = torch.zeros(8, dtype=torch.int64, device=dev)
        self._grad_clean = True                    # FlatParams allocates a zeroed gradient buffer

    def lr_at(self, step: int) -> float:
        """HF get_linear_schedule_with_warmup (base_lego.py:211-223); total_steps == 0 -> constant lr."""
        if self.total_steps <= 0:
            return self.lr
        if step < self.warmup:
            return self.lr * step / max(1, self.warmup)
        return self.lr * max(0.0, (self.total_steps - step) / max(1, self.total_steps - self.warmup))

    def sample_batch(self, step_idx=None, slot=0, stream=None):
        """Resampler.rebuild on device for training step `step_idx` into batch slot `slot`"""
        d, B = self.data, self.B
        step_idx = self.batch_idx if step_idx is None else step_idx
        start = (step_idx * B) % max(1, d.n_rows - B + 1)
        st = _stream() if stream is None else ctypes.c_void_p(stream.cuda_stream)
        ru, ri = _ptr(d.row_user, start), _ptr(d.row_item, start)
        call("lego_sample_negatives", ru, ri, _ptr(d.neg_list), _ptr(d.neg_len), d.neg_cap, B, self.K, d.n_items,
             self.seed, step_idx, _ptr(self._cand[slot]), st)
        call("lego_gather_history", ru, _ptr(d.user_hist), _ptr(d.user_hist_len), B, d.S, _ptr(self._hist[slot]),
             _ptr(self._hist_len[slot]), st)

    def _prefetch(self, step_idx, go=None):
        """sample + plan training step `step_idx` on the side stream `pre` (slot = step_idx % 2)"""
        slot = step_idx % 2
        if go is None:
            go = self._go
            go.record(torch.cuda.current_stream())
        self.pre.wait_event(go)                    # the slot's previous user (step_idx - 2) is complete by then
        self.sample_batch(step_idx, slot, self.pre)
        self.engine.plan_on(self.pre, slot, self._cand[slot], self._hist[slot], self._hist_len[slot])
        if self.dropout and hasattr(self.engine, "prefetch_masks"):
            self.engine.prefetch_masks(self.pre, slot)     # engine.step == step_idx here (one training forward per step)
        with torch.cuda.stream(self.pre):          # row statistics of the planned batch (bench.py), off the main stream
            self.counter_sum += self.engine._slots[slot]["counters"]
        self._ready[slot].record(self.pre)
        self._planned_step = step_idx

    def step(self):
        """sample -> forward -> backward -> all-reduce -> Adam.  Returns the device loss tensor (no sync)."""
        slot = self.batch_idx % 2 if self.prefetch else 0
        self.cand, self.hist, self.hist_len = self._cand[slot], self._hist[slot], self._hist_len[slot]
        last_of_cycle = self._acc + 1 == self.accumulate
        if self.prefetch:
            if self._planned_step != self.batch_idx:
                self._prefetch(self.batch_idx)
            torch.cuda.current_stream().wait_event(self._ready[slot])
            self.engine.use_slot(slot)
        else:
            self.sample_batch()
        if self._acc == 0 and not self._grad_clean:
            self.fp.grad.zero_()
        if self.micro == 1:
            go = neck = None
            if self.prefetch:
                go, neck = self._go, self._neck
                go.record(torch.cuda.current_stream())         # everything before this step's forward
            _, loss = self.engine.forward(self.cand, self.hist, self.hist_len, training=True, planned=self.prefetch,
                                          fork_ev=go, neck_ev=neck)
            if self.prefetch:
                self._prefetch(self.batch_idx + 1, neck)       # next batch: starts where this step's item tower ends
            dist_on = (self.world > 1 or self.force_allreduce) and last_of_cycle
            early = None
            if dist_on and self.fp.split > 0 and os.environ.get("LEGO_AR_BUCKETS", "1") == "2":
                def early():       # everything but the projection weight gradient: overlaps the last backward GEMM
                    self._work = torch.distributed.all_reduce(self.fp.grad[:self.fp.split], group=self.pg, async_op=True)
            self._work = None
            if early is not None:
                self.engine.backward(self.fp.G, before_last=early)
            else:
                self.engine.backward(self.fp.G)
        else:
            main = torch.cuda.current_stream()
            Bm = self.B // self.micro
            for e, st in zip(self.engines, self.streams):
                st.wait_stream(main)
            for phase in ("fwd", "bwd"):                       # enqueue all forwards first so every stream has work early
                for i, (e, st) in enumerate(zip(self.engines, self.streams)):
                    with torch.cuda.stream(st):
                        sl = slice(i * Bm, (i + 1) * Bm)
                        if phase == "fwd":
                            e.forward(self.cand[sl], self.hist[sl], self.hist_len[sl], training=True)
                        else:
                            e.backward(self.fp.G, gloss=1.0 / self.micro)
            for st in self.streams:
                main.wait_stream(st)
            loss = self.loss
            torch.mean(torch.stack([e.loss for e in self.engines]), dim=0, out=self.loss)
        self.batch_idx += 1
        if not self.prefetch:
            for e in self.engines:
                self.counter_sum += e.counters
        if not last_of_cycle:                          # gradients stay in the flat buffer for the next batch of the cycle
            self._acc += 1
            return loss
        self._acc = 0
        if self.world > 1 or self.force_allreduce:
            if getattr(self, "_work", None) is not None:
                torch.distributed.all_reduce(self.fp.grad[self.fp.split:], group=self.pg)   # the late tail (same RCCL stream:
                self._work.wait()                                                           # ordered after the early part)
                self._work = None
            else:
                torch.distributed.all_reduce(self.fp.grad, group=self.pg)      # one RCCL all-reduce per step
        self.step_idx += 1
        call("lego_adam_step", _ptr(self.fp.flat), _ptr(self.fp.grad), _ptr(self.fp.m), _ptr(self.fp.v),
             self.fp.numel, self.lr_at(self.step_idx - 1), 0.9, 0.999, 1e-8, self.step_idx, 1.0 / self.world, 1, _stream())
        self._grad_clean = True                    # Adam cleared the gradient buffer as it consumed it
        return loss
