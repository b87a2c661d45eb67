#!/usr/bin/env python3
"""Pin for SURVEY.md section 8f-3 (table preprocessing): run the REFERENCE's `MINDProcessor.load(regenerate=True)`
(processor/mind_processor.py:30-227 + processor/base_processor.py:197-373) on the small raw MIND sample under
tests/golden/mind_raw/ and store what it hands to its tokenisers -- the compressed item / user tables and the three
interaction splits -- as tests/golden/mind_tables.json.  `legommenders_amd.process_mind` is held to that file
(tests/test_process_mind.py).

Runs ONLY in the build container (needs /root/reference, read-only).  The reference's table format library `unitok` is
not vendored, so the loaders, the 10 % user split, the negative-list merge and the unused-user / unused-item compression run
from the reference's own code (pure pandas), while `unitok` is replaced by in-memory stand-ins that RECORD the data frames
given to `UniTok.tokenize` instead of serialising them.  Two behaviours of unitok are assumed, both stated here because the
fixture depends on them: `Vocab.counter.trim(min_count)` returns the indices seen at least `min_count` times after
`activate()`, and `Feature.get_slice(n)` is the first-n slice for n > 0.  Title tokenisation (GloVeTokenizer -> nltk) is
not exercised: token ids stay unpinned, everything else in the tables is.

    python tests/golden/make_golden_mind.py        # rewrites tests/golden/mind_tables.json
"""
from __future__ import annotations

import json
import os
import random
import sys
import tempfile
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
RAW = os.path.join(HERE, "mind_raw")
RECORDED = []


def install_stubs():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    class Counter:
        def __init__(self, vocab):
            self.vocab, self.active, self.counts = vocab, False, {}

        def activate(self):
            self.active = True

        def initialize(self):
            self.counts = {}

        def trim(self, min_count=1):
            return [i for i in range(len(self.vocab.toks)) if self.counts.get(i, 0) >= min_count]

    class Vocab:
        def __init__(self, name):
            self.name, self.toks, self.index = name, [], {}
            self.counter = Counter(self)

        def append(self, tok):
            if tok not in self.index:
                self.index[tok] = len(self.toks)
                self.toks.append(tok)
            i = self.index[tok]
            if self.counter.active:
                self.counter.counts[i] = self.counter.counts.get(i, 0) + 1
            return i

        def extend(self, toks):
            return [self.append(t) for t in toks]

        def __getitem__(self, i):
            return self.toks[i]

        def __len__(self):
            return len(self.toks)

    class Tok:
        def __init__(self, vocab=None, **kw):
            self.vocab = Vocab(vocab) if isinstance(vocab, str) else vocab
            self.kw = kw

    class UniTok:
        def __init__(self):
            self.features, self.df = [], None

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def add_feature(self, **kw):
            self.features.append(kw)

        def add_index_feature(self):
            pass

        def tokenize(self, df):
            self.df = df.copy()
            RECORDED.append(self)
            return self

        def save(self, path):
            return self

        def __len__(self):
            return len(self.df)

    class Feature:
        @staticmethod
        def get_slice(n):
            return slice(None) if n == 0 else (slice(0, n) if n > 0 else slice(n, None))

    class Symbol:
        def __init__(self, name):
            self.name = name

    unitok = types.ModuleType("unitok")
    for k, v in dict(Vocab=Vocab, UniTok=UniTok, Feature=Feature, Symbol=Symbol, BaseTokenizer=Tok, EntityTokenizer=Tok,
                     EntitiesTokenizer=Tok, DigitTokenizer=Tok, BertTokenizer=Tok, TransformersTokenizer=Tok,
                     VocabularyHub=types.SimpleNamespace(add=lambda v: None)).items():
        setattr(unitok, k, v)
    sys.modules["unitok"] = unitok
    tk = types.ModuleType("unitok.tokenizer")
    gt = types.ModuleType("unitok.tokenizer.glove_tokenizer")
    gt.GloVeTokenizer = Tok
    sys.modules["unitok.tokenizer"], sys.modules["unitok.tokenizer.glove_tokenizer"] = tk, gt
    pig = types.ModuleType("pigmento")
    pig.pnt = lambda *a, **k: None
    sys.modules["pigmento"] = pig
    ge = types.ModuleType("embedder.glove_embedder")
    ge.GloVeEmbedder = types.SimpleNamespace(get_glove_vocab=lambda: Vocab("glove"))
    sys.modules["embedder"] = types.ModuleType("embedder")
    sys.modules["embedder.glove_embedder"] = ge
    ci = types.ModuleType("utils.config_init")
    ci.ModelInit = types.SimpleNamespace(get=lambda name: "")
    sys.modules["utils.config_init"] = ci


def main():
    install_stubs()
    from processor.mind_processor import MINDProcessor          # the reference's own class
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                                            # the processor writes data/mind/*.parquet relative to cwd
        try:
            random.seed(2023)                                    # utils/function.py:58-75 seeds `random`; the split shuffles with it
            proc = MINDProcessor(data_dir=RAW)
            proc.load(regenerate=True)
        finally:
            os.chdir(cwd)
    items, users, train, valid, test = (r.df for r in RECORDED)
    slicer = slice(0, MINDProcessor.NEG_TRUNCATE)
    out = {
        "seed": 2023,
        "items": {"nid": items["nid"].tolist(), "category": items["category"].tolist(), "title": items["title"].tolist()},
        "users": {"uid": users["uid"].tolist(), "history": [list(h) for h in users["history"]],
                  "neg": [list(n)[slicer] for n in users["neg"]]},
        "item_features": [f.get("name") or f.get("column") for f in RECORDED[0].features],
        "user_features": [(f.get("name") or f.get("column"), f.get("truncate")) for f in RECORDED[1].features],
    }
    for name, df in (("train", train), ("valid", valid), ("test", test)):
        out[name] = {"uid": df["uid"].tolist(), "nid": df["nid"].tolist(), "click": [int(x) for x in df["click"]]}
    with open(os.path.join(HERE, "mind_tables.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: (len(v["uid"]) if isinstance(v, dict) and "uid" in v else None) for k, v in out.items()})
    print("items", out["items"]["nid"])
    print("valid users", sorted(set(out["valid"]["uid"])))


if __name__ == "__main__":
    main()
