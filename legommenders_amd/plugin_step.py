"""Training / evaluation through the plug-in route -- `Legommender.forward` operator by operator on id-only batches -- for
item / user operators that the fused ragged engines (engine.py) do not cover, e.g. the BERT news encoder
(SURVEY.md section 8f-2).  Same step as the reference's `Trainer.train` (trainer.py:184-204): sample -> forward ->
backward -> (all-reduce) -> Adam + linear schedule, with the batch sampled on the device by the path's own kernels
(`lego_sample_negatives`, `lego_gather_history`) instead of the DataLoader workers."""
from __future__ import annotations

import ctypes

import torch

from ._lib import call
from .engine import _ptr, _stream
from .train_step import BatchSchedule, rank_seed


class PluginStep:
    def __init__(self, model, data, B: int, K: int = 4, lr: float = 1e-3, total_steps: int = 0, warmup: int = 0,
                 seed: int = 2023, process_group=None, world_size: int = 1, accumulate: int = 1, item_lr=None,
                 tail: str = "keep"):
        from legommenders_amd.loader.env import Env
        from legommenders_amd import functional
        if data.balance not in (None, int(B)):                               # see TrainStep.__init__: the dealing is built for one B
            raise ValueError(f"DeviceData(balance={data.balance}) deals global batches of {data.balance} rows per rank; "
                             f"the step was built with B={B}")
        functional.seed_streams(rank_seed(seed, data.rank))                 # dropout streams differ per rank
        self.schedule = BatchSchedule(data.n_rows, B, tail)
        self.steps_per_epoch = self.schedule.steps_per_epoch
        self.model, self.data, self.B, self.K, self.C = model, data, B, K, K + 1
        self.accumulate, self._acc, self.batch_idx = max(1, int(accumulate)), 0, 0     # trainer.py:171,197-203
        dev = data.tables.title_tok.device
        i32 = dict(dtype=torch.int32, device=dev)
        self.cand = torch.zeros(B, self.C, **i32)
        self.hist = torch.zeros(B, data.S, **i32)
        self.hist_len = torch.zeros(B, **i32)
        self.ar = torch.arange(data.S, device=dev)[None]
        self.params = [p for p in model.parameters() if p.requires_grad]
        if model.config.use_item_content and item_lr:                        # base_lego.py:183-197: pretrained encoder vs the rest
            pretrained, other = model.get_parameters()
            self.opt = torch.optim.Adam([{"params": pretrained, "lr": float(item_lr)}, {"params": other, "lr": lr}])
        else:
            self.opt = torch.optim.Adam(self.params, lr=lr)                 # base_lego.py:201-204 (defaults)
        self.total_steps, self.warmup = total_steps, warmup

        def factor(step):                                                    # HF get_linear_schedule_with_warmup
            if total_steps <= 0:
                return 1.0
            if step < warmup:
                return step / max(1, warmup)
            return max(0.0, (total_steps - step) / max(1, total_steps - warmup))
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, factor)
        self.seed, self.step_idx = seed, 0
        self.pg, self.world = process_group, world_size
        self.Env = Env

    def sample_batch(self):
        d = self.data
        epoch, start, nb = self.schedule.at(self.batch_idx)
        row_user, row_item = d.rows(epoch)                                  # per-epoch reshuffle (train_step.DeviceData)
        ru, ri = _ptr(row_user, start), _ptr(row_item, start)
        pos = d.positions(epoch)
        call("lego_sample_negatives", ru, ri, _ptr(d.neg_list), _ptr(d.neg_len), d.neg_cap, nb, self.K, d.n_items,
             self.seed, self.batch_idx, d.rank, d.world_size, None if pos is None else _ptr(pos, start), _ptr(self.cand), _stream())
        call("lego_gather_history", ru, _ptr(d.user_hist), _ptr(d.user_hist_len), nb, d.S, _ptr(self.hist),
             _ptr(self.hist_len), _stream())
        return nb

    def step(self):
        cm = self.model.cm
        nb = self.sample_batch()                                            # nb < B: the short last batch of an epoch
        batch = {cm.item_col: self.cand[:nb].long(), cm.history_col: self.hist[:nb].long(),
                 cm.mask_col: (self.ar < self.hist_len[:nb, None]).long()}
        self.Env.train()
        self.model.train()
        if self._acc == 0:
            self.opt.zero_grad(set_to_none=True)
        loss = self.model(batch=batch)
        loss.backward()
        self.batch_idx += 1
        self._acc += 1
        if self._acc < self.accumulate:                                      # gradients add up over the cycle
            return loss.detach().reshape(1)
        self._acc = 0
        if self.world > 1:                                                   # one all-reduce of the flattened gradients
            grads = [p.grad for p in self.params if p.grad is not None]
            flat = torch._utils._flatten_dense_tensors(grads)
            torch.distributed.all_reduce(flat, group=self.pg)
            flat.mul_(1.0 / self.world)
            for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
                g.copy_(f)
        self.opt.step()
        self.sched.step()
        self.step_idx += 1
        return loss.detach().reshape(1)

    # ---- checkpoint state, as the reference saves it (base_lego.py:257-267: optimizer.state_dict() + scheduler.state_dict())
    def optimizer_state(self):
        return self.opt.state_dict()

    def load_optimizer_state(self, st):
        self.opt.load_state_dict(st)

    def scheduler_state(self):
        return self.sched.state_dict()

    def load_scheduler_state(self, st):
        self.sched.load_state_dict(st)
        self.step_idx = int(st.get("last_epoch", self.step_idx))


class PluginEvaluator:
    """The reference's fast-eval path on the model itself: item representation cache over all items (pages of
    `item_page`), user cache from `item_repr[history]`, then score = <user, item> per evaluation row
    (loader/cacher/*, model/legommender.py:153-157,202-214,282)."""

    def __init__(self, model, data, item_page: int = 512, user_page: int = 512, process_group=None, rank=0, world_size=1):
        self.model, self.data, self.item_page, self.user_page = model, data, item_page, user_page
        self.pg, self.rank, self.world = process_group, rank, world_size
        self.item_repr = self.user_repr = None

    @torch.no_grad()
    def build_caches(self):
        from legommenders_amd.loader.env import Env
        m, d = self.model, self.data
        dev = d.tables.title_tok.device
        cm = m.cm
        Env.test()
        m.eval()
        m.item_repr = m.user_repr = None
        from legommenders_amd.evaluate import gather_shards, shard_bounds

        def sharded(n, page, encode):                               # this rank's contiguous shard, then one all_gather
            lo, hi, per = shard_bounds(n, self.rank, self.world)
            outs = [encode(s, min(s + page, hi)) for s in range(lo, hi, page)]
            local = torch.cat(outs, 0) if outs else None
            pad = torch.zeros(per, int(m.config.hidden_size), dtype=torch.float32, device=dev)
            if local is not None:
                pad[:hi - lo] = local
            return gather_shards(pad, n, self.pg, self.world).contiguous()

        self.item_repr = sharded(d.n_items, self.item_page, lambda s, e: m.get_item_content(
            {cm.item_col: torch.arange(s, e, device=dev)[:, None]}, cm.item_col)[:, 0])
        m.item_repr = self.item_repr
        S = d.S
        ar = torch.arange(S, device=dev)[None]
        n_users = d.user_hist.shape[0]
        self.user_repr = sharded(n_users, self.user_page, lambda s, e: m.get_user_content(
            {cm.history_col: d.user_hist[s:e].long(), cm.mask_col: (ar < d.user_hist_len[s:e, None]).long()}))
        m.item_repr = None                                                    # training must not see a stale cache
        return self.item_repr, self.user_repr

    @torch.no_grad()
    def scores(self, users, items) -> torch.Tensor:
        """score[r] = <user_repr[users[r]], item_repr[items[r]]> (model/legommender.py:153-157,282 on cached vectors)"""
        dev = self.item_repr.device
        u = torch.as_tensor(users).to(dev, torch.int32).contiguous()
        it = torch.as_tensor(items).to(dev, torch.int32).contiguous()
        n, D = u.numel(), self.item_repr.shape[1]
        gu = torch.empty(n, D, dtype=torch.float32, device=dev)
        gi = torch.empty(n, D, dtype=torch.float32, device=dev)
        out = torch.empty(n, dtype=torch.float32, device=dev)
        st = _stream()
        call("lego_gather_rows", _ptr(self.user_repr), D, D, _ptr(u), n, None, _ptr(gu), D, 0, st)
        call("lego_gather_rows", _ptr(self.item_repr), D, D, _ptr(it), n, None, _ptr(gi), D, 0, st)
        call("lego_rowdot_fwd", _ptr(gu), D, _ptr(gi), D, n, D, _ptr(out), st)
        return out

    def evaluate(self, users, items, labels, groups=None, metrics=("GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10")):
        from legommenders_amd import metrics as M
        import numpy as np
        self.build_caches()
        if self.rank != 0:
            return {}, None
        s = self.scores(users, items)
        g = np.asarray(users if groups is None else groups)
        return M.calculate_device(s, np.asarray(labels), g, list(metrics)), s.cpu().numpy()
