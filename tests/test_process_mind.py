"""Raw-MIND converter on a tiny hand-written corpus: table shapes, filtering and split semantics of
processor/mind_processor.py, and that the trainer's loader reads the result."""
import os

import numpy as np


def _write(tmp):
    os.makedirs(tmp / "mind" / "train")
    os.makedirs(tmp / "mind" / "dev")
    news_tr = ["N1\tsports\tnba\tLakers win the game!\tabs\turl\t[]\t[]", "N2\tnews\tus\tBig storm hits coast\t\turl\t[]\t[]",
               "N3\tsports\tnfl\tUnknownword zzz\tabs\turl\t[]\t[]"]
    news_dev = ["N2\tnews\tus\tBig storm hits coast\t\turl\t[]\t[]", "N4\tfinance\tmarkets\tStocks fall\tabs\turl\t[]\t[]"]
    (tmp / "mind" / "train" / "news.tsv").write_text("\n".join(news_tr) + "\n")
    (tmp / "mind" / "dev" / "news.tsv").write_text("\n".join(news_dev) + "\n")
    beh_tr = [f"{i}\tU{i % 12}\t11/11/2019\tN1 N2 N9\tN3-1 N1-0 N2-0" for i in range(24)] + ["99\tU99\tt\t\tN1-1"]
    beh_dev = ["1\tU1\tt\tN1 N2\tN4-1 N3-0", "2\tU77\tt\tN4\tN1-0 N2-1"]
    (tmp / "mind" / "train" / "behaviors.tsv").write_text("\n".join(beh_tr) + "\n")
    (tmp / "mind" / "dev" / "behaviors.tsv").write_text("\n".join(beh_dev) + "\n")
    words = ["the", "lakers", "win", "game", "!", "big", "storm", "hits", "coast", "stocks", "fall"]
    rs = np.random.RandomState(0)
    (tmp / "glove.txt").write_text("\n".join(w + " " + " ".join(f"{x:.4f}" for x in rs.randn(300)) for w in words) + "\n")


def test_converter_semantics(tmp_path):
    from legommenders_amd.process_mind import build
    _write(tmp_path)
    out = tmp_path / "data" / "mind"
    stats = build(str(tmp_path / "mind"), str(tmp_path / "glove.txt"), str(out), seed=1)
    assert stats["items"] == 4 and stats["categories"] == 3
    assert stats["users"] == 13                      # U0..U11 + U77; U99 has an empty history and is dropped
    it = np.load(out / "items.npz")
    assert it["title_tok"].shape == (4, 30) and it["title_len"].tolist() == [5, 4, 0, 2]   # OOV words are dropped
    assert (it["title_tok"][0, :5] == [1, 2, 0, 3, 4]).all() and it["title_tok"][0, 5] == -1
    us = np.load(out / "users.npz")
    assert us["user_hist_len"][0] == 2               # N9 is not a known news id
    tr, va, te = (np.load(out / f"{n}.npz") for n in ("train", "valid", "test"))
    n_valid_users = len(set(va["user"].tolist()))
    assert n_valid_users == 1                        # 10 % of the 12 train users
    assert set(tr["row_user"].tolist()).isdisjoint(set(va["user"].tolist()))
    assert (tr["row_item"] == 2).all()               # only the positive (N3-1) rows are training samples
    assert us["neg_len"].max() <= 100 and us["neg_len"][tr["row_user"][0]] > 0
    assert te["label"].tolist() == [1, 0, 0, 1]
    assert np.load(tmp_path / "data" / "embeddings" / "glove.npy").shape == (11, 300)


def test_tables_match_the_reference_processor(tmp_path):
    """SURVEY.md 8f-3 pin: the reference's own MINDProcessor.load() was run on tests/golden/mind_raw (make_golden_mind.py,
    unitok replaced by recording stand-ins) -- item / user order after compression, histories, negative lists, the seeded
    10 % validation split and the three interaction tables must come out the same here."""
    import json
    from legommenders_amd.process_mind import build, build_tables
    here = os.path.dirname(os.path.abspath(__file__))
    gold = json.load(open(os.path.join(here, "golden", "mind_tables.json")))
    raw = os.path.join(here, "golden", "mind_raw")
    t = build_tables(raw, seed=gold["seed"])
    assert list(t["news"]) == gold["items"]["nid"]                                   # N19 (never referenced) is gone
    assert [c for c, _ in t["news"].values()] == gold["items"]["category"]
    assert [ti for _, ti in t["news"].values()] == gold["items"]["title"]
    assert list(t["users"]) == gold["users"]["uid"]                                  # U4, U5, U19 dropped
    assert list(t["users"].values()) == gold["users"]["history"]
    assert [t["neg"][u] for u in t["users"]] == gold["users"]["neg"]
    for name in ("train", "valid", "test"):
        g = gold[name]
        assert t[name] == list(zip(g["uid"], g["nid"], g["click"])), name
    # ... and the npz tables carry exactly that, as indices into the compressed tables
    words = ["the", "storm", "stocks"]
    (tmp_path / "glove.txt").write_text("\n".join(w + " " + " ".join(["0.1"] * 300) for w in words) + "\n")
    out = tmp_path / "data" / "mind"
    build(raw, str(tmp_path / "glove.txt"), str(out), seed=gold["seed"])
    it, us = np.load(out / "items.npz"), np.load(out / "users.npz")
    nid, uid = it["nid"].tolist(), us["uid"].tolist()
    assert nid == gold["items"]["nid"] and uid == gold["users"]["uid"]
    assert [it["category"][c] for c in it["cat"]] == gold["items"]["category"]
    for r, (h, ng) in enumerate(zip(gold["users"]["history"], gold["users"]["neg"])):
        assert [nid[i] for i in us["user_hist"][r, : us["user_hist_len"][r]]] == h[:50]
        assert [nid[i] for i in us["neg_list"][r, : us["neg_len"][r]]] == ng
    tr = np.load(out / "train.npz")
    pos = [(u, n) for u, n, c in zip(*(gold["train"][k] for k in ("uid", "nid", "click"))) if c == 1]
    assert [(uid[u], nid[n]) for u, n in zip(tr["row_user"], tr["row_item"])] == pos
    for name in ("valid", "test"):
        z, g = np.load(out / f"{name}.npz"), gold[name]
        assert [uid[u] for u in z["user"]] == g["uid"] and [nid[n] for n in z["item"]] == g["nid"] and z["label"].tolist() == g["click"]


def test_trainer_loader_reads_converted_tables(tmp_path, monkeypatch):
    from legommenders_amd.config_init import Obj
    from legommenders_amd.process_mind import build
    from legommenders_amd.trainer import load_world
    _write(tmp_path)
    build(str(tmp_path / "mind"), str(tmp_path / "glove.txt"), str(tmp_path / "data" / "mind"), seed=1)
    w = load_world(Obj({"base_dir": str(tmp_path / "data" / "mind"), "name": "mind"}), seed=1)
    assert w["n_items"] == 4 and w["V"] == 11 and w["T"] == 30 and w["S"] == 50 and w["neg_cap"] == 100
    assert set(w["valid"]) == {"user", "item", "label"}
