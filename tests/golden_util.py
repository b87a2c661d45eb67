"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def glove_table(seed, V, E0=300):
    """Same generator as tests/golden/make_golden.py (legacy RandomState is frozen)."""
    return (np.random.RandomState(seed).standard_normal((V, E0)) * 0.4).astype(np.float32)


def load_model_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    P = {k[len("param::"):]: z[k] for k in z.files if k.startswith("param::")}
    G = {k[len("grad::"):]: z[k] for k in z.files if k.startswith("grad::")}
    if meta["embed"] == "glove":
        P["embedding_vocab_table.glove.embedding.weight"] = glove_table(meta["table_seed"], meta["V"])
    tables = {k: z[k] for k in ("title_tok", "title_len", "cat", "user_hist", "user_hist_len")}
    batch = {k: z[k] for k in ("cand", "hist", "hist_len")}
    return meta, P, G, tables, batch, z["logits"], float(z["loss"])


MODEL_FIXTURES = ["naml_glove_d64", "nrms_null_d64", "nrms_glove_d64", "naml_glove_cfg1", "naml_glove_d256"]
BERT_FIXTURES = ["bert_naml_small", "bert_naml_tune1"]          # SURVEY.md 8(f)-2, tests/golden/make_golden_bert.py
