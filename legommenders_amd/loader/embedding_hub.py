"""EmbeddingHub / Transformation (mirror of the reference's loader/embedding_hub.py:73-385): one module per
vocabulary / feature, frozen pre-trained tables wrapped in Dropout(Linear(Embedding)) when the dimension
differs or the policy is `linear`, fresh trainable `nn.Embedding(vocab.size, embedding_dim)` otherwise.
Same `state_dict` keys (`<vocab>.embedding.weight`, `<vocab>.linear.{weight,bias}` / `<vocab>.weight`);
the look-up + projection + dropout arithmetic runs in the HIP kernels (functional.glove_project / embedding)."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import numpy as np
import torch
from torch import nn

from legommenders_amd import functional as F_hip
from legommenders_amd.loader.env import Env


class HipEmbedding(nn.Embedding):
    """nn.Embedding whose forward is the HIP row gather (pad id -1 -> zero row, dense gradient)."""

    def forward(self, indexes):
        if self.weight.requires_grad:
            return F_hip.embedding(indexes, self.weight)
        return F_hip.embedding(indexes, self.weight.detach())


class Transformation(nn.Module):
    """y = Dropout(Linear(Embedding(x)))  (embedding_hub.py:73-96)."""

    def __init__(self, embedding: nn.Embedding, to_dimension: int, transformation_dropout: float):
        super().__init__()
        self.embedding = embedding
        self.linear = nn.Linear(embedding.weight.data.shape[1], to_dimension)
        self.dropout = nn.Dropout(transformation_dropout)

    def forward(self, indexes):
        # frozen=False (embedding_hub.py:171,262): the table's weight requires grad and the op's autograd formula adds the dense
        # table gradient (glove_project_bwd_table); such a model trains through the plug-in route (trainer.build_model)
        return F_hip.glove_project(indexes, self.embedding.weight, self.linear.weight, self.linear.bias,
                                   p=self.dropout.p, training=self.training)


class PretrainedEmbedding:
    def __init__(self, embedder, transformation, transformation_dropout, frozen):
        self.embedder, self.transformation = embedder, transformation
        self.transformation_dropout, self.frozen = transformation_dropout, frozen


class EmbeddingHub:
    LINEAR, AUTO, DEFAULT = "linear", "auto", "default"
    global_types = {LINEAR, AUTO}
    pretrained_types = {DEFAULT, LINEAR, AUTO}

    def __init__(self, embedding_dim: int, transformation: str, transformation_dropout: float):
        if transformation not in self.global_types:
            raise ValueError(f"invalid transformation type {transformation}, expected {self.global_types}")
        self.embedding_dim = embedding_dim
        self.transformation = transformation
        self.transformation_dropout = transformation_dropout
        self._vocab_size: Dict[str, int] = {}
        self.vocab_table = nn.ModuleDict()
        self.feature_table = nn.ModuleDict()
        self._pretrained_vocab_embeddings: Dict[str, PretrainedEmbedding] = {}
        self._pretrained_feature_embeddings: Dict[str, PretrainedEmbedding] = {}

    def load_pretrained_embedding(self, path, *, vocab_name=None, col_name=None, transformation=DEFAULT,
                                  transformation_dropout=None, frozen=True, array: Optional[np.ndarray] = None):
        if vocab_name is None and col_name is None:
            raise ValueError("vocab_name or col_name must be specified")
        if vocab_name is not None and col_name is not None:
            raise ValueError("only one of vocab_name and col_name can be specified")
        name = vocab_name or col_name
        arr = array if array is not None else np.load(path)
        weight = arr if isinstance(arr, torch.Tensor) else torch.tensor(arr, dtype=torch.float32)
        embedding = HipEmbedding(weight.shape[0], weight.shape[1], _weight=weight.float(), _freeze=True)
        if name == "<vocab_name>":
            raise ValueError("please specify the vocab name for the pretrained embedding in the config")
        if transformation not in self.pretrained_types:
            raise ValueError(f"invalid transformation type {transformation}, expected {self.pretrained_types}")
        if transformation == self.DEFAULT:
            transformation = self.transformation
        if transformation_dropout is None:
            transformation_dropout = self.transformation_dropout
        target = self._pretrained_vocab_embeddings if vocab_name is not None else self._pretrained_feature_embeddings
        target[name] = PretrainedEmbedding(embedding, transformation, transformation_dropout, frozen)

    def _process_pretrained_embedding(self, name, size, pe: PretrainedEmbedding):
        if int(pe.embedder.weight.shape[0]) != size:
            raise ValueError(f"{name} does not match the expected vocab size {size}")
        pe.embedder.weight.requires_grad = not pe.frozen
        embedding_size = int(pe.embedder.weight.data.shape[1])
        if embedding_size != self.embedding_dim or self.transformation == self.LINEAR:
            pe.embedder = Transformation(pe.embedder, self.embedding_dim, pe.transformation_dropout)

    def build_feature_embedding(self, feature) -> bool:
        if feature.name in self.feature_table or feature.name not in self._pretrained_feature_embeddings:
            return False
        pe = self._pretrained_feature_embeddings[feature.name]
        self._process_pretrained_embedding(feature.name, feature.tokenizer.vocab.size, pe)
        self.feature_table.add_module(feature.name, pe.embedder.to(Env.device))
        return True

    def build_vocab_embedding(self, vocab) -> None:
        if vocab.name in self.vocab_table:
            return
        if vocab.name not in self._pretrained_vocab_embeddings:
            self.vocab_table.add_module(vocab.name, HipEmbedding(vocab.size, self.embedding_dim).to(Env.device))
            return
        pe = self._pretrained_vocab_embeddings[vocab.name]
        self._process_pretrained_embedding(vocab.name, vocab.size, pe)
        self.vocab_table.add_module(vocab.name, pe.embedder.to(Env.device))

    def register_vocab(self, vocab) -> None:
        if vocab.name in self._vocab_size:
            if self._vocab_size[vocab.name] != vocab.size:
                raise ValueError(f"conflict in vocab {vocab.name}: {self._vocab_size[vocab.name]} vs {vocab.size}")
            return
        self._vocab_size[vocab.name] = vocab.size
        self.build_vocab_embedding(vocab)

    def register_ut(self, ut, used_cols: Iterable[str]) -> None:
        for col in used_cols:
            feature = ut.meta.features[col]
            self.build_feature_embedding(feature)
            self.register_vocab(feature.tokenizer.vocab)

    def __call__(self, vocab_name: str, col_name: Optional[str] = None) -> nn.Module:
        if col_name and col_name in self.feature_table:
            return self.feature_table[col_name]
        return self.vocab_table[vocab_name]
