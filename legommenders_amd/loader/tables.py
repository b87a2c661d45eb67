"""Minimal table containers standing in for the third-party `unitok` objects the reference loads
(`LegoUT`, loader/ut/lego_ut.py; UniTok's on-disk layout is not vendored -- SURVEY.md section 8f-3).
They expose exactly the attributes the hot path reads: `len`, `[i] -> row dict`,
`.meta.features[col].{name,max_len,tokenizer.vocab.{name,size}}`, `.key_feature`, plus flat int32
column arrays for the device-resident path."""
from __future__ import annotations

import types
from typing import Dict, List, Optional

import numpy as np


class Vocab:
    def __init__(self, name: str, size: int = 0):
        self.name, self._size, self._toks = name, size, []

    def append(self, tok):
        self._toks.append(tok)
        return len(self._toks) - 1

    @property
    def size(self):
        return max(self._size, len(self._toks))


class Feature:
    def __init__(self, name: str, vocab: Vocab, max_len: Optional[int] = None):
        self.name, self.max_len = name, max_len
        self.tokenizer = types.SimpleNamespace(vocab=vocab)


class Table:
    """Column store: scalar columns are int arrays [n]; sequence columns are ([n,max_len] int array with -1
    pads, [n] lengths)."""

    def __init__(self, features: List[Feature], columns: Dict[str, object], key: str):
        self.meta = types.SimpleNamespace(features={f.name: f for f in features})
        self.columns = columns
        self.key_feature = key
        first = columns[key]
        self._n = len(first[0] if isinstance(first, tuple) else first)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        row = {}
        for name, col in self.columns.items():
            if isinstance(col, tuple):
                vals, lens = col
                row[name] = [int(v) for v in vals[i, : lens[i]]]
            else:
                row[name] = int(col[i])
        return row

    def seq(self, col):
        vals, lens = self.columns[col]
        return np.asarray(vals), np.asarray(lens)

    def scalar(self, col):
        return np.asarray(self.columns[col])
