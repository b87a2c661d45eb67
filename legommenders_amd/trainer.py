"""`trainer.py` CLI of the MI355X path (mirror of the reference's trainer.py:299-322 / base_lego.py:82-142):

    python -m legommenders_amd.trainer --data config/data/synthetic.yaml --model config/model/naml.yaml \
        --embed config/embed/glove.yaml --batch_size 64 --lr 0.001 --hidden_size 256 --cuda 0
    torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 8 -m legommenders_amd.trainer ...   # data parallel

Same flags and config schema (`--data --model --embed --exp --batch_size --lr --hidden_size --item_hidden_size
--cuda --seed --epoch --epoch_batch --interval --patience --metric --simple_dev --load_sign`); train -> dev
evaluation with early stopping (`Monitor`, utils/monitor.py:43-73) -> best checkpoint -> test, checkpoints at
`checkpoints/<data>/<model>/<signature>.pt` with the reference's `state_dict` keys.  `--cuda -1` is refused:
there is no CPU path here (run the reference for that)."""
from __future__ import annotations

import base64
import hashlib
import json
import os
import random
import time
from typing import Dict

import numpy as np
import torch

from legommenders_amd._lib import LegoHipError
from legommenders_amd.config_init import CommandInit, Obj


def seeding(seed=2023):
    """utils/function.py:58-75"""
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)


def get_signature(data, embed, model, exp) -> str:
    """utils/function.py:146-186: 8 chars of the url-safe base64 md5 of the canonical config json"""
    s = json.dumps({"data": data, "embed": embed, "model": model, "exp": exp}, sort_keys=True, ensure_ascii=False)
    return base64.urlsafe_b64encode(hashlib.md5(s.encode("utf-8")).digest()).decode("utf-8").rstrip("=")[:8]


class Monitor:
    """early stopping (utils/monitor.py:43-73), including its first-push quirk (best_index stays 0)."""

    def __init__(self, minimize: bool, patience: int = 2):
        self.patience, self.minimize = patience, minimize
        self.best_value, self.best_index, self.current_index = None, 0, -1

    def push(self, value: float) -> str:
        self.current_index += 1
        if self.best_value is None:
            self.best_value = value
            return "best"
        if self.minimize ^ (value > self.best_value):
            self.best_value, self.best_index = value, self.current_index
            return "best"
        if self.current_index - self.best_index >= self.patience:
            return "stop"
        return "skip"


def load_world(data_cfg: Obj, seed: int) -> dict:
    """MIND tables: the synthetic MIND-small-shaped world, or npz files under data.base_dir."""
    from legommenders_amd.synthetic import MIND_SMALL, make_world
    base = data_cfg.base_dir
    if base == "synthetic":
        cfg = dict(MIND_SMALL)
        if data_cfg.scale == "small":
            cfg.update(n_items=3000, n_users=2000, n_rows=8000, V=5000)
        w = make_world(seed=seed, **cfg)
        rs = np.random.RandomState(seed + 1)
        for split, n_u in (("valid", min(2000, cfg["n_users"])), ("test", min(4000, cfg["n_users"]))):
            users = np.repeat(rs.choice(cfg["n_users"], size=n_u, replace=False), 10)
            items = rs.randint(0, cfg["n_items"], size=users.size)
            labels = np.zeros(users.size, dtype=np.int64)
            labels[::10] = 1
            w[split] = dict(user=users, item=items, label=labels)
        return w
    need = ["items", "users", "train", "valid", "test"]
    z = {n: np.load(os.path.join(base, n + ".npz")) for n in need}
    w = dict(title_tok=z["items"]["title_tok"], title_len=z["items"]["title_len"], cat=z["items"]["cat"],
             user_hist=z["users"]["user_hist"], user_hist_len=z["users"]["user_hist_len"],
             neg_list=z["users"]["neg_list"], neg_len=z["users"]["neg_len"],
             row_user=z["train"]["row_user"], row_item=z["train"]["row_item"])
    w.update(n_items=len(w["cat"]), n_users=len(w["user_hist_len"]), n_rows=len(w["row_user"]),
             V=int(z["items"]["vocab_size"]), n_cat=int(w["cat"].max()) + 1, T=w["title_tok"].shape[1],
             S=w["user_hist"].shape[1], neg_cap=w["neg_list"].shape[1])
    for split in ("valid", "test"):
        w[split] = dict(user=z[split]["user"], item=z[split]["item"], label=z[split]["label"])
    return w


def build_model(config: Obj, world: dict, device):
    """Manager.load_model_configs + load_embeddings + Legommender (loader/manager.py:271-326) on our tables."""
    from legommenders_amd.loader.class_hub import ClassHub
    from legommenders_amd.loader.column_map import ColumnMap
    from legommenders_amd.loader.embedding_hub import EmbeddingHub
    from legommenders_amd.loader.env import Env
    from legommenders_amd.loader.tables import Feature, Table, Vocab
    from legommenders_amd.model.lego_config import LegoConfig
    from legommenders_amd.model.legommender import Legommender
    from legommenders_amd.synthetic import glove_like

    model, embed, data = config.model, config.embed, config.data
    Env.data_name = data.name
    first = (data.item.inputs() or [{"title@glove": world["T"]}])[0]          # data yaml: item.inputs[0] = {col: max_len} | col
    tcol = next(iter(first)) if isinstance(first, dict) else str(first)
    tvocab = tcol.split("@", 1)[1] if "@" in tcol else tcol                   # `title@glove` -> vocab `glove` (UniTok naming)
    glove_v, cat_v = Vocab(tvocab, world["V"]), Vocab("category", world["n_cat"])
    item_v, user_v = Vocab("item_id", world["n_items"]), Vocab("user_id", world["n_users"])
    item_ut = Table([Feature("item_id", item_v), Feature(tcol, glove_v, world["T"]), Feature("category", cat_v)],
                    {"item_id": np.arange(world["n_items"]), tcol: (world["title_tok"], world["title_len"]),
                     "category": world["cat"]}, "item_id")
    user_ut = Table([Feature("user_id", user_v), Feature("history", item_v, world["S"])],
                    {"user_id": np.arange(world["n_users"]), "history": (world["user_hist"], world["user_hist_len"])}, "user_id")
    ops, preds = ClassHub.operators(), ClassHub.predictors()
    lc = LegoConfig(**model.config())
    lc.set_component_classes(ops[model.meta.item], ops[model.meta.user], preds[model.meta.predictor])
    lc.set_item_ut(item_ut, [tcol, "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(**data.column_map()))
    eh = EmbeddingHub(embedding_dim=lc.item_hidden_size, transformation=embed.transformation,
                      transformation_dropout=embed.transformation_dropout)
    for info in (embed.embeddings() or []):
        if os.path.exists(info["path"]):
            arr = np.load(info["path"])
        elif data.base_dir == "synthetic":      # GloVe-like 300-d table, or (other vocabularies, e.g. bert) one of the model's width
            width = 300 if info.get("vocab_name", "glove") == "glove" else int(lc.item_hidden_size)
            arr = glove_like(world["V"], width, seed=2024, device=device)
        else:
            raise ValueError(f"pre-trained embedding {info['path']} not found")
        eh.load_pretrained_embedding(info["path"], vocab_name=info.get("vocab_name"), col_name=info.get("col_name"),
                                     frozen=info.get("frozen", True), array=arr)
    eh.register_ut(item_ut, [tcol, "category"])
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    legommender = Legommender(lc).to(device)
    kind = {"CNNOperator": "naml", "AttentionOperator": "nrms"}.get(type(lc.item_operator).__name__)
    if type(lc.predictor).__name__ != "DotPredictor":
        raise LegoHipError("the MI355X training path covers the Dot predictor")
    if kind is None or type(lc.user_operator).__name__ not in ("AdaOperator", "AttentionOperator") \
            or (kind == "naml") != (type(lc.user_operator).__name__ == "AdaOperator"):
        kind = "plugin"          # e.g. BertBase + Ada: trained operator by operator (plugin_step.py), not by a fused engine
    if any(not info.get("frozen", True) for info in (embed.embeddings() or [])):
        kind = "plugin"          # an un-frozen pre-trained table: the fused engines keep the pre-trained table frozen
    if kind == "plugin":
        from legommenders_amd.engine import ItemTables
        legommender.attach_item_table(ItemTables(world["title_tok"], world["title_len"], world["cat"], device))
    return legommender, kind


class Trainer:
    def __init__(self, config: Obj):
        from legommenders_amd.evaluate import Evaluator
        from legommenders_amd.loader.env import Env
        from legommenders_amd.train_step import BatchSchedule, DeviceData, TrainStep
        self.config, self.exp = config, config.exp
        config.seed = int(config.seed or 2023)
        seeding(config.seed)
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        cuda = config.cuda if config.cuda is not None else local
        self.device = Env.set_device(int(cuda) if self.world_size == 1 else local)     # -1 -> LegoHipError
        torch.cuda.set_device(self.device)
        self.pg = None
        if self.world_size > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("nccl", device_id=self.device)
            self.pg = torch.distributed.group.WORLD
        self.signature = get_signature(config.data(), config.embed(), config.model(), config.exp())
        self.ckpt_dir = os.path.join("checkpoints", config.data.name, config.model.name)
        if self.rank == 0:
            os.makedirs(self.ckpt_dir, exist_ok=True)
            json.dump(config(), open(os.path.join(self.ckpt_dir, self.signature + ".json"), "w"), default=str)
        self.world = load_world(config.data, config.seed)
        self.legommender, self.kind = build_model(config, self.world, self.device)
        self.data = DeviceData(self.world, self.device, rank=self.rank, world_size=self.world_size, seed=config.seed,
                               balance=int(self.exp.policy.batch_size))
        pol = self.exp.policy
        self.B = int(pol.batch_size)
        # every row of the rank's shard once per epoch, the short last batch included (DataLoader drop_last=False,
        # manager.py:374-381); shards are equal on all ranks, so steps and schedule length agree everywhere
        self.steps_per_epoch = BatchSchedule(self.data.n_rows, self.B, "keep").steps_per_epoch
        # schedule length as the reference counts it: len(train_set) // batch_size * epoch (base_lego.py:218-222)
        sched_steps = max(1, self.data.n_rows // self.B) * int(pol.epoch)
        accumulate = int(pol.accumulate_batch or 1)                       # trainer.py:171
        Env.simple_dev = bool(pol.simple_dev)                            # base_lego.py:121
        params = {k: v.detach() for k, v in self.legommender.state_dict().items()}
        glove = any(k.endswith("glove.embedding.weight") for k in params)
        heads = getattr(self.legommender.item_op.config, "num_attention_heads", 8)
        if self.kind == "plugin":
            from legommenders_amd.plugin_step import PluginEvaluator, PluginStep
            self.ts = PluginStep(self.legommender, self.data, self.B, K=self.legommender.neg_count, lr=float(pol.lr),
                                 total_steps=sched_steps, warmup=int(pol.n_warmup or 0),
                                 seed=config.seed, process_group=self.pg, world_size=self.world_size, accumulate=accumulate,
                                 item_lr=pol.item_lr)
            self.evaluator = PluginEvaluator(self.legommender, self.data,
                                             item_page=int(self.legommender.config.cache_page_size or 512),
                                             process_group=self.pg, rank=self.rank, world_size=self.world_size)
        else:
            self.ts = TrainStep(self.kind, params, self.data, self.B, K=self.legommender.neg_count, lr=float(pol.lr),
                                total_steps=sched_steps, warmup=int(pol.n_warmup or 0),
                                seed=config.seed, heads=heads, glove=glove, process_group=self.pg, world_size=self.world_size,
                                accumulate=accumulate)
            self._heads, self._glove = heads, glove
            self.evaluator = Evaluator(self.kind, self.ts.fp.P, self.data, heads=heads, glove=glove,
                                       process_group=self.pg, rank=self.rank, world_size=self.world_size)
        if self.exp.load.sign:
            self.load(str(self.exp.load.sign).replace("@", ""))

    def log(self, *a):
        if self.rank == 0:
            print("[%s]" % time.strftime("%H:%M:%S"), *a, flush=True)
            with open(os.path.join(self.ckpt_dir, self.signature + ".log"), "a") as f:
                f.write(" ".join(str(x) for x in a) + "\n")

    def save(self):
        """{model, optimizer, scheduler} as base_lego.py:257-267: `model` with the reference's state_dict keys, `optimizer` a
        torch.optim.Adam state_dict over the trainable parameters in `legommender.parameters()` order (the engine route builds
        it from its flat moment buffers), `scheduler` a LambdaLR state_dict -- loadable by the reference with
        `model_only: false`, and the reference's by us."""
        path = os.path.join(self.ckpt_dir, self.signature + ".pt")
        model = self.legommender.state_dict() if self.kind == "plugin" else self.ts.fp.P
        opt = self.ts.optimizer_state() if self.kind == "plugin" else self.ts.optimizer_state(self._trainable_order())
        torch.save({"model": {k: v.detach().cpu() for k, v in model.items()}, "optimizer": opt,
                    "scheduler": self.ts.scheduler_state()}, path)
        self.log("save model to", path)

    def _trainable_order(self):
        """trainable parameter names in `parameters()` order (what torch.optim.Adam indexes its state by)"""
        return [n for n, p in self.legommender.named_parameters() if p.requires_grad]

    def load(self, sign, model_only=None):
        """base_lego.py:240-253: model always; optimizer + scheduler unless `exp.load.model_only`."""
        path = os.path.join(self.ckpt_dir, sign + ".pt")
        state = torch.load(path, map_location=self.device, weights_only=False)
        if self.kind == "plugin":
            self.legommender.load_state_dict(state["model"], strict=bool(self.exp.load.strict))
        else:
            for k, v in state["model"].items():
                if k in self.ts.fp.P:
                    self.ts.fp.P[k].copy_(v)
                elif self.exp.load.strict:
                    raise KeyError(k)
            self.ts.params_changed()
        model_only = bool(self.exp.load.model_only) if model_only is None else model_only
        if not model_only:
            if "optimizer" not in state or "scheduler" not in state:
                raise KeyError(f"{path} holds no optimizer / scheduler state (set exp.load.model_only: true)")
            if self.kind == "plugin":
                self.ts.load_optimizer_state(state["optimizer"])
            else:
                self.ts.load_optimizer_state(state["optimizer"], self._trainable_order())
            self.ts.load_scheduler_state(state["scheduler"])
        self.log("load model from", path, "(model only)" if model_only else "(model + optimizer + scheduler)")

    def evaluate(self, split, metrics):
        rows = self.world[split]
        res, _ = self.evaluator.evaluate(rows["user"], rows["item"], rows["label"], metrics=metrics)
        return res

    def simple_evaluate(self):
        """`--simple_dev true` (trainer.py:126-140, manager.py:331-346, resampler.py:159-171): the dev rows with label 1,
        K sampled negatives each, the training loss in eval mode (no dropout); mean of the batch means, a short last
        batch included.  Runs on rank 0 (no collective inside)."""
        import ctypes
        from legommenders_amd._lib import call
        from legommenders_amd.engine import NamlEngine, NrmsEngine, _ptr, _stream
        from legommenders_amd.loader.env import Env
        rows = self.world["valid"]
        pos = np.asarray(rows["label"]) == 1
        dev, d = self.device, self.data
        users = torch.as_tensor(np.asarray(rows["user"])[pos]).to(dev, torch.int32).contiguous()
        items = torch.as_tensor(np.asarray(rows["item"])[pos]).to(dev, torch.int32).contiguous()
        K, C, S = self.legommender.neg_count, self.legommender.neg_count + 1, d.S
        if not hasattr(self, "_dev_engines"):
            self._dev_engines = {}
        total = torch.zeros((), dtype=torch.float32, device=dev)
        n_batches = 0
        for bi, s in enumerate(range(0, users.numel(), self.B)):
            b = min(self.B, users.numel() - s)
            cand = torch.empty(b, C, dtype=torch.int32, device=dev)
            hist = torch.empty(b, S, dtype=torch.int32, device=dev)
            hist_len = torch.empty(b, dtype=torch.int32, device=dev)
            call("lego_sample_negatives", _ptr(users, s), _ptr(items, s), _ptr(d.neg_list), _ptr(d.neg_len), d.neg_cap, b, K,
                 d.n_items, int(self.config.seed) + 1, bi, 0, 1, None, _ptr(cand), _stream())
            call("lego_gather_history", _ptr(users, s), _ptr(d.user_hist), _ptr(d.user_hist_len), b, S, _ptr(hist),
                 _ptr(hist_len), _stream())
            if self.kind == "plugin":
                cm = self.legommender.cm
                Env.dev()
                self.legommender.eval()
                with torch.no_grad():
                    loss = self.legommender(batch={cm.item_col: cand.long(), cm.history_col: hist.long(),
                                                   cm.mask_col: (torch.arange(S, device=dev)[None] < hist_len[:, None]).long()})
            else:
                if b not in self._dev_engines:
                    P = self.ts.fp.P
                    self._dev_engines[b] = NamlEngine(P, d.tables, b, C, S, p_proj=0.0, p_conv=0.0) if self.kind == "naml" else \
                        NrmsEngine(P, d.tables, b, C, S, heads=self._heads, glove=self._glove, p_proj=0.0, p_att=0.0)
                _, loss = self._dev_engines[b].forward(cand, hist, hist_len, training=False)
            total += loss.reshape(())
            n_batches += 1
        return {"loss": float(total) / max(n_batches, 1)}

    def train(self):
        from legommenders_amd import metrics as M
        from legommenders_amd.loader.env import Env
        pol, store = self.exp.policy, self.exp.store
        simple = bool(Env.simple_dev)
        monitor = Monitor(patience=int(store.patience), minimize=simple or M.is_minimize(store.metric))
        interval = int(pol.check_interval or 0)
        if interval < 0:
            interval = max(self.steps_per_epoch // (-interval), 1)
        for epoch in range(int(pol.epoch)):
            t0 = time.time()
            run, n = None, 0
            for step in range(self.steps_per_epoch):
                loss = self.ts.step()
                run = loss.clone() if run is None else run + loss      # no host sync inside the epoch
                n += 1
                if interval and (step + 1) % interval == 0:
                    self.log(f"[epoch {epoch}] step {step + 1} / {self.steps_per_epoch}, loss {float(run) / n:.4f}")
                if pol.epoch_batch and step > (pol.epoch_batch if pol.epoch_batch > 0
                                               else max(self.steps_per_epoch // (-pol.epoch_batch), 1)):
                    break
            torch.cuda.synchronize()
            dt = time.time() - t0
            self.log(f"[epoch {epoch}] train loss {float(run) / max(n, 1):.4f}  "
                     f"{n * self.B * self.world_size / dt:.0f} impressions/s ({self.world_size} GPU)")
            action = "skip"
            if simple:
                res = self.simple_evaluate() if self.rank == 0 else {}
            else:
                res = self.evaluate("valid", [store.metric])   # every rank: the cache build is sharded (evaluate.py)
            if self.rank == 0:
                self.log(f"[epoch {epoch}] " + " ".join(f"{k} {v:.4f}" for k, v in res.items()))
                action = monitor.push(res["loss" if simple else store.metric])
                if action == "best":
                    self.save()
            if self.world_size > 1:
                flag = torch.tensor([{"skip": 0, "best": 1, "stop": 2}[action]], device=self.device)
                torch.distributed.broadcast(flag, 0)
                action = ["skip", "best", "stop"][int(flag.item())]
            if action == "stop":
                self.log("Early stop triggered.")
                break
        self.log("Training Ended")

    def test(self):
        res = self.evaluate("test", list(self.exp.metrics() or ["GAUC"]))
        if self.rank != 0:
            return {}
        self.log("[test] " + " ".join(f"{k} {v:.4f}" for k, v in res.items()))
        with open(os.path.join(self.ckpt_dir, self.signature + ".csv"), "w") as f:
            f.write(",".join(res.keys()) + "\n" + ",".join(f"{v:.6f}" for v in res.values()) + "\n")
        return res

    def run(self):
        self.train()
        if self.world_size > 1:
            torch.distributed.barrier()                        # rank 0 may still be writing the best checkpoint
        if os.path.exists(os.path.join(self.ckpt_dir, self.signature + ".pt")):
            self.load(self.signature, model_only=True)         # every rank: the test caches are built from all shards
        out = self.test()
        if self.world_size > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return out


def get_configurations(kwargs=None) -> Obj:
    return CommandInit(
        required_args=["data", "model"],
        default_args=dict(embed="config/embed/null.yaml", exp="config/exp/default.yaml", hidden_size=256,
                          item_hidden_size="${hidden_size}$", item_page_size=64),
    ).parse(kwargs=kwargs)


if __name__ == "__main__":
    Trainer(config=get_configurations()).run()
