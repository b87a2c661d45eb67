"""trainer.py CLI surface: argument typing, YAML template grammar, early-stopping monitor, signature,
the refusal of the CPU path -- and (gpu) a short end-to-end train -> dev -> checkpoint -> test run."""
import os

import numpy as np
import pytest

from legommenders_amd.config_init import CommandInit, argparse, typed
from legommenders_amd.trainer import Monitor, get_configurations, get_signature

REF = "/root/reference"


def test_cli_typing_rules():
    kw = argparse(["--data", "x.yaml", "--batch_size", "64", "--lr", "0.001", "--cuda", "-1", "--fast_eval", "false",
                   "--load_sign", "null", "--metric", "GAUC"])
    assert kw == {"data": "x.yaml", "batch_size": 64, "lr": 0.001, "cuda": -1, "fast_eval": False, "load_sign": None,
                  "metric": "GAUC"}
    assert typed("-2") == -2 and typed("1e-3") == 1e-3 and typed("True") is True


def test_template_grammar_on_own_configs():
    c = get_configurations(dict(data="config/data/synthetic.yaml", model="config/model/nrms.yaml", batch_size=32, lr=0.01,
                                hidden_size=64, num_item_heads=4))
    m = c.model.config()
    assert m["hidden_size"] == 64 and m["item_hidden_size"] == 64
    assert m["item_config"]["num_attention_heads"] == 4 and m["user_config"]["num_attention_heads"] == 8
    assert c.model.meta.item == "Attention" and c.model.meta.user == "Attention" and c.model.meta.predictor == "Dot"
    assert c.exp.policy.batch_size == 32 and c.exp.policy.lr == 0.01 and c.exp.store.metric == "GAUC"
    assert c.exp.policy.check_interval == -2 and c.exp.load.sign is None
    assert c.embed.name is None and c.embed.embeddings() == []


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")
def test_reference_yaml_files_load_unchanged():
    """The reference's own config files resolve through our grammar (drop-in for the listed configs)."""
    c = CommandInit(["data", "model"], dict(embed=f"{REF}/config/embed/null.yaml", exp=f"{REF}/config/exp/default.yaml",
                                            hidden_size=256, item_hidden_size="${hidden_size}$", item_page_size=64)
                    ).parse(dict(data=f"{REF}/config/data/mind.yaml", model=f"{REF}/config/model/naml.yaml",
                                 embed=f"{REF}/config/embed/glove.yaml", batch_size=64, lr=0.001))
    assert c.model.meta.item == "CNN" and c.model.meta.user == "Ada" and c.model.meta.predictor == "Dot"
    assert c.model.config()["item_config"] == {"dropout": 0.1, "kernel_size": 3}
    assert c.data.item.ut == "data/mind/items" and c.data.column_map.group_col == "user_id"
    assert c.embed.embeddings()[0]["vocab_name"] == "glove" and c.embed.transformation_dropout == 0.1
    assert c.exp.policy.epoch == 50 and c.exp.store.patience == 5 and c.exp.metrics()[0] == "GAUC"
    n = CommandInit(["data", "model"], dict(embed=f"{REF}/config/embed/null.yaml", exp=f"{REF}/config/exp/default.yaml",
                                            hidden_size=256, item_hidden_size="${hidden_size}$")
                    ).parse(dict(data=f"{REF}/config/data/mind.yaml", model=f"{REF}/config/model/nrms.yaml", batch_size=64))
    assert n.model.config()["item_config"]["inputer_config"] == {"use_cls_token": False, "use_sep_token": True}


def test_monitor_early_stopping_sequence():
    m = Monitor(minimize=False, patience=2)
    assert [m.push(v) for v in (0.5, 0.6, 0.55, 0.58, 0.59)] == ["best", "best", "skip", "stop", "stop"]
    lo = Monitor(minimize=True, patience=1)
    assert [lo.push(v) for v in (1.0, 0.9, 0.95)] == ["best", "best", "stop"]


def test_signature_is_stable_and_short():
    a = get_signature({"a": 1}, {}, {"m": [1, 2]}, {"lr": 0.1})
    assert a == get_signature({"a": 1}, {}, {"m": [1, 2]}, {"lr": 0.1}) and len(a) == 8
    assert a != get_signature({"a": 2}, {}, {"m": [1, 2]}, {"lr": 0.1})


def test_cpu_device_request_is_refused():
    from legommenders_amd._lib import LegoHipError
    from legommenders_amd.trainer import Trainer
    cfg = get_configurations(dict(data="config/data/synthetic.yaml", model="config/model/naml.yaml",
                                  embed="config/embed/glove.yaml", batch_size=8, hidden_size=64, cuda=-1, world="small"))
    with pytest.raises(LegoHipError):
        Trainer(cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("model,embed", [("naml", "glove"), ("nrms", "null")])
def test_short_training_run_end_to_end(model, embed, tmp_path, monkeypatch):
    from legommenders_amd.trainer import Trainer
    monkeypatch.chdir(tmp_path)
    cfg = get_configurations(dict(data="config/data/synthetic.yaml", model=f"config/model/{model}.yaml",
                                  embed=f"config/embed/{embed}.yaml", batch_size=32, hidden_size=64, lr=0.001, cuda=0,
                                  world="small", epoch=2, patience=2, interval=0))
    tr = Trainer(cfg)
    res = tr.run()
    assert set(res) == {"GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10"} and all(np.isfinite(v) for v in res.values())
    assert 0.3 < res["GAUC"] < 0.7                      # random labels: the metric must sit near chance
    ck = os.path.join("checkpoints", "synthetic", cfg.model.name, tr.signature + ".pt")
    import torch
    state = torch.load(ck)
    assert set(state["model"]) == set(tr.legommender.state_dict())      # checkpoint carries the reference's keys


@pytest.mark.gpu
def test_resume_restores_optimizer_and_scheduler(tmp_path, monkeypatch):
    """`--load_sign <sig>` with `exp.load.model_only: false` (base_lego.py:240-253): parameters, Adam moments, step counter and the
    schedule position come back, and the optimizer blob is a torch.optim.Adam state_dict over `parameters()` order (so the
    reference can read it: loaded into a torch Adam over the model's own parameters here)."""
    import torch
    from legommenders_amd.trainer import Trainer
    monkeypatch.chdir(tmp_path)
    kw = dict(data="config/data/synthetic.yaml", model="config/model/naml.yaml", embed="config/embed/glove.yaml", batch_size=32,
              hidden_size=64, lr=0.001, cuda=0, world="small", epoch=1, patience=2, interval=0)
    tr = Trainer(get_configurations(dict(kw)))
    for _ in range(7):
        tr.ts.step()
    tr.save()
    ck = torch.load(os.path.join("checkpoints", "synthetic", tr.config.model.name, tr.signature + ".pt"), weights_only=False)
    params = [p for p in tr.legommender.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=0.001)
    opt.load_state_dict(ck["optimizer"])                                   # what the reference does (base_lego.py:252)
    assert len(ck["optimizer"]["state"]) == len(params) and ck["scheduler"]["last_epoch"] == 7
    cfg2 = get_configurations(dict(kw, load_sign=tr.signature))
    cfg2.exp.load.model_only = False
    tr2 = Trainer(cfg2)
    assert tr2.ts.step_idx == 7
    for k in tr.ts.fp.names:
        o, n = tr.ts.fp.offsets[k], tr.ts.fp.P[k].numel()
        assert torch.equal(tr2.ts.fp.m[o:o + n], tr.ts.fp.m[o:o + n]) and torch.equal(tr2.ts.fp.v[o:o + n], tr.ts.fp.v[o:o + n])
        assert torch.equal(tr2.ts.fp.P[k], tr.ts.fp.P[k])
    assert abs(tr2.ts.lr_at(tr2.ts.step_idx) - tr.ts.lr_at(7)) < 1e-12


def test_bert_naml_yaml_resolves():
    c = get_configurations(dict(data="config/data/synthetic-bert.yaml", model="config/model/bert-naml.yaml",
                                embed="config/embed/bertbase.yaml", batch_size=8, hidden_size=64))
    m = c.model.config()
    assert c.model.meta.item == "BertBase" and c.model.meta.user == "Ada" and c.model.meta.predictor == "Dot"
    assert m["item_hidden_size"] == 768 and m["hidden_size"] == 64 and m["item_page_size"] == 64
    assert m["item_config"]["tune_from"] == 0 and m["item_config"]["use_lora"] is False
    assert c.embed.embeddings()[0]["vocab_name"] == "bert"


@pytest.mark.gpu
@pytest.mark.parametrize("tune_from", [0, 1])
def test_bert_naml_cli_trains_through_the_plugin_route(tune_from, tmp_path, monkeypatch):
    """`trainer.py --model config/model/bert-naml.yaml [--tune_from 1]` end to end: a tiny LOCAL BERT checkpoint (width 768
    as the yaml demands, 3 blocks -> tune_from = 0 keeps 2; tune_from = 1 caches layer 1 of every item in HBM and keeps 1),
    the plug-in training step, cached evaluation, checkpoint, test."""
    import torch
    from transformers import BertConfig, BertModel
    from legommenders_amd.trainer import Trainer
    monkeypatch.chdir(tmp_path)
    ck = str(tmp_path / "tiny-bert")
    torch.manual_seed(0)
    monkeypatch.setenv("LEGO_LAYER_CACHE_SAVE", "0")
    BertModel(BertConfig(vocab_size=5000, hidden_size=768, num_hidden_layers=3, num_attention_heads=12, intermediate_size=128,
                         max_position_embeddings=64)).save_pretrained(ck)
    monkeypatch.setenv("LEGO_MODEL_BERTBASE", ck)
    cfg = get_configurations(dict(data="config/data/synthetic-bert.yaml", model="config/model/bert-naml.yaml",
                                  embed="config/embed/bertbase.yaml", batch_size=16, hidden_size=64, lr=0.0005, cuda=0,
                                  world="small", epoch=1, patience=2, interval=0, epoch_batch=6, tune_from=tune_from,
                                  item_lr=0.00001))
    tr = Trainer(cfg)
    groups = tr.ts.opt.param_groups                                  # base_lego.py:183-197: transformer at item_lr, rest at lr
    assert [g["initial_lr"] for g in groups] == [0.00001, 0.0005]
    assert sum(p.numel() for p in groups[0]["params"]) == sum(p.numel() for p in tr.legommender.item_op.transformer.parameters())
    assert tr.kind == "plugin" and type(tr.legommender.item_op).__name__ == "BertBaseOperator"
    assert len(tr.legommender.item_op.transformer.encoder.layer) == 3 - 1 - tune_from
    if tune_from:
        assert tuple(tr.legommender.item_op.hidden_weights.shape) == (3000, 31, 768)
        assert tr.legommender.item_op.hidden_weights.is_cuda
    assert "embedding_vocab_table.bert.weight" in tr.legommender.state_dict()      # frozen table, no projection (768 == 768)
    res = tr.run()
    assert set(res) == {"GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10"} and all(np.isfinite(v) for v in res.values())
    assert 0.3 < res["GAUC"] < 0.7
    state = torch.load(os.path.join("checkpoints", "synthetic", cfg.model.name, tr.signature + ".pt"))
    assert set(state["model"]) == set(tr.legommender.state_dict())


def test_sizer_counts_trainable_parameters_only(capsys):
    """sizer.py:49-66; SURVEY.md 8a13 pins NAML-GloVe-256 at 476 416 trainable parameters (frozen GloVe not counted)"""
    from legommenders_amd import sizer
    cfg = sizer.get_configurations(dict(data="config/data/synthetic.yaml", model="config/model/naml.yaml",
                                        embed="config/embed/glove.yaml", hidden_size=256, world="small"))
    sz = sizer.Sizer(cfg)
    assert sz.run() == 476416
    out = capsys.readouterr().out
    assert "item_op.cnn.weight (256, 256, 3)" in out and "Number of parameters: 0.48M" in out
    assert "glove.embedding.weight" not in out
    # embed null: the token table becomes a trainable [V, D] parameter
    cfg = sizer.get_configurations(dict(data="config/data/synthetic.yaml", model="config/model/nrms.yaml", hidden_size=64,
                                        world="small"))
    sz = sizer.Sizer(cfg)
    names = dict(sz.named_trainable())
    assert tuple(names["embedding_vocab_table.glove.weight"].shape) == (sz.world["V"], 64)
    assert tuple(names["item_op.multi_head_attention.in_proj_weight"].shape) == (192, 64)


def test_status_timer_stops_after_total_count():
    from legommenders_amd.tester import StatusTimer
    st = StatusTimer(total_count=3)
    with pytest.raises(StopIteration):
        for _ in range(10):
            st.run()
            st.run()
    assert st.count == 3 and st.avgms() >= 0.0 and not st.timing


@pytest.mark.gpu
def test_tester_restores_checkpoint_and_times_scoring(tmp_path, monkeypatch):
    """tester.py: `--load_sign` restores the trainer's checkpoint, `test` reproduces the trainer's test metrics and writes
    the result file, `--latency` stops after num_batches timed scoring steps."""
    from legommenders_amd import tester
    from legommenders_amd.trainer import Trainer
    monkeypatch.chdir(tmp_path)
    common = dict(data="config/data/synthetic.yaml", model="config/model/naml.yaml", embed="config/embed/glove.yaml",
                  batch_size=32, hidden_size=64, lr=0.001, cuda=0, world="small")
    tr = Trainer(get_configurations(dict(common, epoch=1, patience=2, interval=0)))
    ref = tr.run()
    te = tester.Tester(tester.get_configurations(dict(common, load_sign=tr.signature)))
    res = te.run()
    assert res.keys() == ref.keys() and all(abs(res[k] - ref[k]) < 1e-6 for k in ref)
    lines = open(os.path.join(te.ckpt_dir, te.signature + ".result")).read().splitlines()
    assert lines[0].startswith("GAUC: ") and len(lines) == len(res)
    te = tester.Tester(tester.get_configurations(dict(common, load_sign=tr.signature, latency=True, num_batches=20)))
    st = te.run()
    assert st.count == 20 and 0.0 < st.avgms() < 50.0


@pytest.mark.gpu
@pytest.mark.parametrize("model,embed", [("naml", "glove"), ("nrms", "null")])
def test_simple_dev_and_accumulate_batch_flags(model, embed, tmp_path, monkeypatch):
    """`--simple_dev true` monitors the eval-mode training loss of the positive dev rows (minimised) instead of a ranking
    metric; `accumulate_batch 2` takes one optimiser step per two batches (trainer.py:126-140,164-171,197-203)."""
    from legommenders_amd.trainer import Trainer
    monkeypatch.chdir(tmp_path)
    cfg = get_configurations(dict(data="config/data/synthetic.yaml", model=f"config/model/{model}.yaml",
                                  embed=f"config/embed/{embed}.yaml", batch_size=32, hidden_size=64, lr=0.001, cuda=0,
                                  world="small", epoch=2, patience=2, interval=0, simple_dev=True, accumulate_batch=2))
    tr = Trainer(cfg)
    assert tr.ts.accumulate == 2
    first = tr.simple_evaluate()["loss"]
    assert 1.2 < first < 2.2                                     # ~log(5) for an untrained model
    res = tr.run()
    assert tr.ts.batch_idx == 2 * tr.steps_per_epoch and tr.ts.step_idx == tr.steps_per_epoch
    log = open(os.path.join("checkpoints", "synthetic", cfg.model.name, tr.signature + ".log")).read()
    assert "[epoch 0] loss " in log and any(l.startswith("[test] GAUC") for l in log.splitlines())
    assert set(res) == {"GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10"}
