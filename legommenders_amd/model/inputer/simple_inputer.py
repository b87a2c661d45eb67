"""SimpleInputer (mirror of the reference's model/inputer/simple_inputer.py:11-66): one padded id sequence
per column (pad = Env.UNSET = -1, scalar columns become length-1 sequences), per-column embedding look-up
through the EmbeddingHub, masked positions zeroed."""
from collections import OrderedDict
from typing import Dict

import torch

from legommenders_amd.loader.env import Env
from legommenders_amd.model.inputer.base_inputer import BaseInputer


class SimpleInputer(BaseInputer):
    output_single_sequence = False

    @classmethod
    def pad(cls, l: list, max_len: int):
        return l + [Env.UNSET] * (max_len - len(l)), [1] * len(l) + [0] * (max_len - len(l))

    def sample_rebuilder(self, sample: dict):
        input_ids, attention_mask = dict(), dict()
        for col in self.inputs:
            max_len = self.ut.meta.features[col].max_len
            value = sample[col]
            if not max_len:
                value, max_len = [value], 1
            ids, mask = self.pad(list(value), max_len)
            input_ids[col] = torch.tensor(ids)
            attention_mask[col] = torch.tensor(mask)
        return dict(input_ids=input_ids, attention_mask=attention_mask)

    def get_mask(self, batched_samples: Dict[str, torch.Tensor]):
        return OrderedDict((k, v.to(Env.device)) for k, v in batched_samples["attention_mask"].items())

    def get_embeddings(self, batched_samples: Dict[str, torch.Tensor]):
        """The reference rewrites pad ids to 0, looks row 0 up and zeroes it afterwards
        (simple_inputer.py:58-63); here pads stay -1 and the gather kernel emits the zero row directly."""
        out = OrderedDict()
        input_ids, attention_mask = batched_samples["input_ids"], batched_samples["attention_mask"]
        for col in input_ids:
            vocab = self.ut.meta.features[col].tokenizer.vocab.name
            seq = input_ids[col].to(Env.device)
            mask = attention_mask[col].to(Env.device)
            seq = torch.where(mask > 0, seq, torch.full_like(seq, Env.UNSET))
            emb = self.eh(vocab, col_name=col)(seq)
            out[col] = emb
        return out
