"""ConcatInputer (mirror of the reference's model/inputer/concat_inputer.py:24-114): all columns of an item
in ONE compact sequence `[CLS?] col_1 [SEP?] col_2 [SEP?] ... [PAD...]`, special vocab [PAD]=0,[CLS]=1,[SEP]=2;
embeddings = sum of the per-column masked look-ups."""
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from legommenders_amd.loader.env import Env
from legommenders_amd.loader.tables import Vocab
from legommenders_amd.model.inputer.base_inputer import BaseInputer


class ConcatInputer(BaseInputer):
    output_single_sequence = True

    vocab = Vocab(name="__cat_inputer_special_ids")
    PAD = vocab.append("[PAD]")
    CLS = vocab.append("[CLS]")
    SEP = vocab.append("[SEP]")

    def __init__(self, use_cls_token, use_sep_token, **kwargs):
        super().__init__(**kwargs)
        self.use_cls_token = use_cls_token
        self.use_sep_token = use_sep_token
        self.vocab_activated = self.use_sep_token or self.use_cls_token
        self.max_content_len = sum((self.ut.meta.features[c].max_len or 1) for c in self.inputs)
        self.max_sequence_len = self.max_content_len + int(use_cls_token) + int(use_sep_token) * len(self.inputs)

    def get_vocabs(self) -> Optional[List]:
        return [self.vocab] if self.vocab_activated else []

    def get_empty_input(self):
        return torch.ones(self.max_sequence_len, dtype=torch.long) * Env.UNSET

    def sample_rebuilder(self, sample):
        pos = 0
        input_ids = OrderedDict()
        special_ids = self.get_empty_input()
        if self.use_cls_token:
            special_ids[pos] = self.CLS
            pos += 1
        for col in self.inputs:
            value = sample[col]
            if not isinstance(value, list):
                value = [value]
            ids = self.get_empty_input()
            ids[pos: pos + len(value)] = torch.tensor(value, dtype=torch.long)
            pos += len(value)
            input_ids[col] = ids
            if self.use_sep_token:
                special_ids[pos] = self.SEP
                pos += 1
        if self.vocab_activated:
            special_ids[pos:] = self.PAD
            input_ids[self.vocab.name] = special_ids
        attention_mask = torch.tensor([1] * pos + [0] * (self.max_sequence_len - pos), dtype=torch.long)
        return dict(input_ids=input_ids, attention_mask=attention_mask)

    def get_mask(self, batched_samples: Dict[str, torch.Tensor]):
        return batched_samples["attention_mask"].to(Env.device)

    def get_embeddings(self, batched_samples: Dict[str, torch.Tensor]):
        input_ids = batched_samples["input_ids"]
        total = None
        for col in input_ids:
            vocab = col if col == self.vocab.name else self.ut.meta.features[col].tokenizer.vocab.name
            seq = input_ids[col].to(Env.device)          # -1 where the column is absent: zero rows from the gather kernel
            emb = self.eh(vocab)(seq)
            total = emb if total is None else total + emb
        return total
