"""Throughput of the BERT-NAML plug-in route (BERT-base news encoder, item_hidden = 768) on one
MI355X: the product's own training step (`plugin_step.PluginStep`: device sampler, `Legommender.forward`, backward, the flat-buffer Adam launch)
on a MIND-small-shaped synthetic world, random-init BERT-base
(no pretrained weights offline; tune_from = 0 -> 11 of the 12 blocks run, as in the reference).

    python tools/bert_naml_bench.py [--batch 64] [--steps 5] [--layers 12] [--hidden 256] [--tune_from 9]
The transformer runs through PyTorch-ROCm; the table gather, Linear(768 -> D), additive pools, dot + CE are the path's kernels."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(batch=64, steps=5, warmup=2, layers=12, hidden=256, tune_from=0, n_items=5000, item_page_size=64):
    """build the model, run `warmup` + `steps` training steps; returns the result record (also used by bench.py's
    `secondary.bert_naml_base`)"""
    a = argparse.Namespace(batch=batch, steps=steps, warmup=warmup, layers=layers, hidden=hidden, tune_from=tune_from)
    from legommenders_amd.engine import ItemTables
    from legommenders_amd.loader.class_hub import ClassHub
    from legommenders_amd.loader.column_map import ColumnMap
    from legommenders_amd.loader.embedding_hub import EmbeddingHub
    from legommenders_amd.loader.env import Env
    from legommenders_amd.loader.tables import Feature, Table, Vocab
    from legommenders_amd.model.lego_config import LegoConfig
    from legommenders_amd.model.legommender import Legommender
    from legommenders_amd.synthetic import MIND_SMALL, make_world
    dev = torch.device("cuda:0")
    Env.set_device(dev)
    cfg = dict(MIND_SMALL); cfg.update(n_items=n_items, n_users=4000, n_rows=20000, V=30522)
    w = make_world(seed=2023, **cfg)
    H, D, V = 768, a.hidden, cfg["V"]
    n_items = w["title_tok"].shape[0]
    tok_v, cat_v, item_v = Vocab("bert", V), Vocab("category", cfg["n_cat"]), Vocab("item_id", n_items)
    user_v = Vocab("user_id", w["user_hist"].shape[0])
    item_ut = Table([Feature("item_id", item_v), Feature("title@bert", tok_v, 30), Feature("category", cat_v)],
                    {"item_id": np.arange(n_items), "title@bert": (w["title_tok"], w["title_len"]), "category": w["cat"]}, "item_id")
    user_ut = Table([Feature("user_id", user_v), Feature("history", item_v, 50)],
                    {"user_id": np.arange(user_v.size), "history": (w["user_hist"], w["user_hist_len"])}, "user_id")
    bert = dict(vocab_size=V, hidden_size=H, num_hidden_layers=a.layers, num_attention_heads=12, intermediate_size=3072,
                max_position_embeddings=512)
    ops, preds = ClassHub.operators(), ClassHub.predictors()
    lc = LegoConfig(hidden_size=D, item_hidden_size=H, neg_count=4, item_page_size=item_page_size,     # bert-naml.yaml: 64
                    user_config={"inputer_config": {"use_cls_token": False, "use_sep_token": False}},
                    item_config={"tune_from": a.tune_from, "use_lora": False, "lora_r": None, "lora_alpha": None,
                                 "inputer_config": {"use_cls_token": False, "use_sep_token": False}, "transformer_config": bert})
    lc.set_component_classes(ops["BertBase"], ops["Ada"], preds["Dot"])
    lc.set_item_ut(item_ut, ["title@bert", "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(item_col="item_id", user_col="user_id", history_col="history", neg_col="neg",
                                label_col="click", group_col="user_id"))
    eh = EmbeddingHub(embedding_dim=H, transformation="auto", transformation_dropout=0.1)
    eh.load_pretrained_embedding(None, vocab_name="bert", frozen=True,
                                 array=(np.random.RandomState(1).standard_normal((V, H)) * 0.02).astype(np.float32))
    eh.register_ut(item_ut, ["title@bert", "category"])
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    os.environ.setdefault("LEGO_LAYER_CACHE_SAVE", "0")
    model = Legommender(lc).to(dev)
    t_cache = time.perf_counter()
    model.attach_item_table(ItemTables(w["title_tok"], w["title_len"], w["cat"], dev))
    torch.cuda.synchronize()
    t_cache = time.perf_counter() - t_cache
    from legommenders_amd.arena import arena_of
    from legommenders_amd.plugin_step import PluginStep
    from legommenders_amd.train_step import DeviceData
    B = a.batch
    ps = PluginStep(model, DeviceData(w, dev, seed=2023), B, K=4, lr=1e-4, seed=2023, tail="drop")
    step = ps.step
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    # one device sync per step (a step is ~125 ms: the sync costs nothing): un-synchronised, the host runs several steps ahead, each step's
    # ~15 GB of saved activations is requested before the previous step's are back in the caching allocator, and the timed steps pay for
    # fresh hipMalloc segments -- 165-213 ms per step inside bench.py's process against 122-128 ms for the same steps taken one at a time
    step_ms, step_rows = [], []
    from legommenders_amd import bert_native
    for _ in range(a.steps):
        bert_native.ROWS_SEEN = []
        t1 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        step_ms.append(round((time.perf_counter() - t1) * 1e3, 1))
        step_rows.append(int(sum(bert_native.ROWS_SEEN)))
    bert_native.ROWS_SEEN = None
    us_per_row = [round(m * 1e3 / max(1, r), 3) for m, r in zip(step_ms, step_rows)]
    # the MEDIAN step: a step whose ragged batch is larger than any before it makes the caching allocator fetch new segments (hipMalloc:
    # 120 -> 290-360 ms for that step, seen for two of five steps in bench.py's process); the mean and the list are kept beside it
    dt = sorted(step_ms)[len(step_ms) // 2] / 1e3
    ar = arena_of(dev)
    if os.environ.get("LEGO_BERT_STEP_TIMES") == "1":
        print("bert_naml_bench step times (ms):", step_ms, "allocated GB", round(torch.cuda.memory_allocated() / 2**30, 2),
              "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2), file=sys.stderr, flush=True)
    n_par = sum(p.numel() for p in model.parameters() if p.requires_grad)
    # per-product table of the native blocks: two more steps with every product launch bracketed by HIP events (outside the timing above)
    kernels = None
    if getattr(model.item_op, "native", False) and model.item_op._native_ok(32):
        bert_native.TIMERS = {}
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        tm, bert_native.TIMERS = bert_native.TIMERS, None
        kernels = {}
        for tag, evs in tm.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            fl = sum(f for _, _, f in evs)
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            kernels[tag] = {"ms_per_step": round(ms / 2, 3), "launches_per_step": len(evs) // 2, "tflops": round(tf, 1),
                            "frac_of_f32_mfma_peak": round(tf / 157.3, 4)}

    return ({"model": "BERT-NAML plug-in route", "batch": B, "bert_layers_run": len(model.item_op.transformer.encoder.layer),
           "trainable_params": n_par, "s_per_step": round(dt, 4), "impressions_per_s": round(B / dt, 1),
           "timing": "median of the per-step wall times (one device sync per step)", "step_ms": step_ms, "s_per_step_mean": round(sum(step_ms) / 1e3 / len(step_ms), 4), "loss": float(loss.detach()),
           "item_page_size": item_page_size, "effective_item_page": model._item_page(10 ** 9), "tune_from": a.tune_from, "layer_cache_s": round(t_cache, 3) if a.tune_from else None,
           "layer_cache_GB": round(model.item_op.hidden_weights.numel() * 4 / 1e9, 3) if a.tune_from else None,
           "step_ms_max_over_min": round(max(step_ms) / max(1e-9, min(step_ms)), 3),
           "live_rows_per_step": step_rows, "us_per_live_row": us_per_row,
           "us_per_live_row_max_over_min": round(max(us_per_row) / max(1e-9, min(us_per_row)), 3),     # batches are ragged: a step's time follows its live rows
           "workspace_arena": {"chunk_allocations_total": ar.allocations, "peak_GB": round(ar.peak / 2**30, 2), "reserved_GB": round(sum(c.numel() for c in ar.chunks) / 2**30, 2)},
           "torch_allocator": {"allocated_GB": round(torch.cuda.memory_allocated() / 2**30, 2), "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2),
                               "num_device_alloc": torch.cuda.memory_stats().get("num_device_alloc", None)},
           "blocks_on": "the path's kernels over ragged rows (legommenders_amd/bert_native.py)" if kernels is not None
                        else "transformers modules on PyTorch-ROCm (LEGO_BERT_NATIVE=0 or an uncovered configuration)",
           "kernels": kernels})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--tune_from", type=int, default=0, help="k > 0: cached-layer mode, blocks [k+1:] train on the HBM-resident layer-k cache")
    ap.add_argument("--item_page_size", type=int, default=64, help="items per transformer call (config/model/bert-naml.yaml: 64; 0 = all at once)")
    a = ap.parse_args()
    print(run(a.batch, a.steps, a.warmup, a.layers, a.hidden, a.tune_from, item_page_size=a.item_page_size))


if __name__ == "__main__":
    main()
