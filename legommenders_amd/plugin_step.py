"""Training / evaluation through the plug-in route -- `Legommender.forward` operator by operator on id-only batches -- for
item / user operators that the fused ragged engines (engine.py) do not cover, e.g. the BERT news encoder
(SURVEY.md section 8f-2).  Same step as the reference's `Trainer.train` (trainer.py:184-204): sample -> forward ->
backward -> (all-reduce) -> Adam + linear schedule, with the batch sampled on the device by the path's own kernels
(`lego_sample_negatives`, `lego_gather_history`) instead of the DataLoader workers."""
from __future__ import annotations

import torch

from ._lib import call
from .engine import _ptr, _stream
from .train_step import BatchSchedule, rank_seed


class _Groups:
    """what callers read of `torch.optim.Adam`: `.param_groups` (params, lr, initial_lr, betas, eps)"""

    def __init__(self, groups):
        self.param_groups = groups


class PluginStep:
    """Parameters, gradients and Adam moments of the model's trainable tensors live in ONE flat fp32 buffer each (as in TrainStep): the
    module tree keeps its tensors -- every `nn.Parameter.data` / `.grad` becomes a view -- so state_dict keys, hooks and `model.parameters()`
    are untouched, and a step is: forward, backward (the BERT blocks add their gradients straight into the views: bert_native.DIRECT_GRADS),
    ONE all-reduce of the gradient buffer, ONE `lego_adam_step` launch per learning-rate group (it clears the gradients it consumes).
    Same update rule as `torch.optim.Adam` + HF linear schedule (base_lego.py:175-223) on every step in which every trainable tensor takes
    part in the loss -- the case for every model of the path.  (torch skips a parameter whose `.grad` is None; the flat rule applies a zero
    gradient, which changes nothing while that parameter's moments are zero.)"""

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
        if model.config.use_item_content and item_lr:                        # base_lego.py:183-197: pretrained encoder vs the rest
            pretrained, other = model.get_parameters()
            groups = [(list(pretrained), float(item_lr)), (list(other), float(lr))]
        else:
            groups = [([p for p in model.parameters() if p.requires_grad], float(lr))]           # base_lego.py:201-204 (defaults)
        self.params = [p for ps, _ in groups for p in ps]                    # torch.optim.Adam's parameter order (state_dict indices)
        self._flatten(groups, dev)
        self.total_steps, self.warmup = total_steps, warmup
        self.seed, self.step_idx = seed, 0
        self.pg, self.world = process_group, world_size
        self.Env = Env

    # ------------------------------------------------------------------ flat buffers
    def _flatten(self, groups, dev):
        """lay the trainable tensors out group by group; inside a group, the q / k / v tensors of an attention block back to back (weights,
        then biases), so that the BERT blocks' stacked [3H, H] in-projection -- weight, bias and both gradients -- are views, not copies"""
        names = {id(p): k for k, p in self.model.named_parameters()}
        self.ranges, self.offsets, off = [], {}, 0
        for ps, lr in groups:
            start, seen, order = off, set(), []
            by_name = {names.get(id(p), ""): p for p in ps}
            for p in ps:
                if id(p) in seen:
                    continue
                k = names.get(id(p), "")
                trio = None
                for kind in ("weight", "bias"):
                    if k.endswith("attention.self.query." + kind):
                        stem = k[: -len("query." + kind)]
                        trio = [by_name.get(stem + n + "." + kind) for n in ("query", "key", "value")]
                if trio and all(t is not None and t.dtype == torch.float32 and t.numel() % 4 == 0 for t in trio):
                    order += [t for t in trio if id(t) not in seen]
                    seen.update(id(t) for t in trio)
                else:
                    order.append(p)
                    seen.add(id(p))
            for p in order:
                if p.dtype != torch.float32:
                    raise TypeError("the plug-in step trains fp32 parameters")
                self.offsets[id(p)] = off
                off += (p.numel() + 3) // 4 * 4                             # every tensor 16-B aligned inside the buffers
            self.ranges.append((start, off, lr))
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gflat, self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        with torch.no_grad():
            for p in self.params:
                o, n = self.offsets[id(p)], p.numel()
                view = self.flat[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.gflat[o:o + n].view(p.shape)
        self.opt = _Groups([{"params": list(ps), "lr": lr, "initial_lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0}
                            for ps, lr in groups])

    def factor(self, step: int) -> float:
        """HF get_linear_schedule_with_warmup (base_lego.py:211-223); total_steps == 0 -> constant"""
        if self.total_steps <= 0:
            return 1.0
        if step < self.warmup:
            return step / max(1, self.warmup)
        return max(0.0, (self.total_steps - step) / max(1, self.total_steps - self.warmup))

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
        from legommenders_amd import bert_native
        cm = self.model.cm
        nb = self.sample_batch()                                            # nb < B: the short last batch of an epoch
        batch = {cm.item_col: self.cand[:nb].long(), cm.history_col: self.hist[:nb].long(),
                 cm.mask_col: (self.ar < self.hist_len[:nb, None]).long()}
        self.Env.train()
        self.model.train()
        loss = self.model(batch=batch)                                      # (the gradient buffer is clean: Adam cleared what it consumed)
        prev, bert_native.DIRECT_GRADS = bert_native.DIRECT_GRADS, True
        try:
            loss.backward()
        finally:
            bert_native.DIRECT_GRADS = prev
        self.batch_idx += 1
        self._acc += 1
        if self._acc < self.accumulate:                                      # gradients add up over the cycle
            return loss.detach().reshape(1)
        self._acc = 0
        if self.world > 1:                                                   # ONE all-reduce of the flat gradient buffer (1 / world inside Adam)
            torch.distributed.all_reduce(self.gflat, group=self.pg)
        self.apply_update()
        return loss.detach().reshape(1)

    def apply_update(self):
        self.step_idx += 1
        f, st = self.factor(self.step_idx - 1), _stream()
        for (lo, hi, lr), g in zip(self.ranges, self.opt.param_groups):
            g["lr"] = lr * f
            if hi > lo:
                call("lego_adam_step", _ptr(self.flat, lo), _ptr(self.gflat, lo), _ptr(self.m, lo), _ptr(self.v, lo), hi - lo, lr * f, 0.9, 0.999,
                     1e-8, self.step_idx, 1.0 / self.world, 1, st)

    # ---- checkpoint state, as the reference saves it (base_lego.py:257-267: optimizer.state_dict() + scheduler.state_dict()): a
    # torch.optim.Adam state_dict (parameter indices in group order) and a LambdaLR state_dict, loadable in both directions
    def optimizer_state(self):
        state, idx = {}, 0
        groups = []
        for (lo, hi, lr), g in zip(self.ranges, self.opt.param_groups):
            ids = []
            for p in g["params"]:
                o, n = self.offsets[id(p)], p.numel()
                if self.step_idx > 0:
                    state[idx] = {"step": torch.tensor(float(self.step_idx)), "exp_avg": self.m[o:o + n].view(p.shape).detach().cpu().clone(),
                                  "exp_avg_sq": self.v[o:o + n].view(p.shape).detach().cpu().clone()}
                ids.append(idx)
                idx += 1
            groups.append({"lr": lr * self.factor(self.step_idx), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                           "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                           "decoupled_weight_decay": False, "initial_lr": lr, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_optimizer_state(self, st):
        if "state" not in st or "param_groups" not in st:
            raise ValueError("optimizer state is not a torch.optim.Adam state_dict")
        ids = [i for g in st["param_groups"] for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError(f"optimizer state holds {len(ids)} parameters, the model trains {len(self.params)}")
        self.m.zero_(); self.v.zero_()
        step = 0
        for i, p in zip(ids, self.params):
            e = st["state"].get(i)
            if e is None:
                continue
            o, n = self.offsets[id(p)], p.numel()
            if e["exp_avg"].numel() != n:
                raise ValueError(f"optimizer state of parameter {i}: {tuple(e['exp_avg'].shape)} does not match {tuple(p.shape)}")
            self.m[o:o + n].copy_(e["exp_avg"].reshape(-1)); self.v[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(e["step"])))
        self.step_idx = step

    def scheduler_state(self):
        base = [lr for _, _, lr in self.ranges]
        return {"base_lrs": base, "last_epoch": self.step_idx, "_step_count": self.step_idx + 1, "verbose": False,
                "_get_lr_called_within_step": False, "_last_lr": [lr * self.factor(self.step_idx) for lr in base],
                "lr_lambdas": [None] * len(base), "_is_initial": False}

    def load_scheduler_state(self, st):
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
