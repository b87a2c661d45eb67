"""Inputer contract (mirror of the reference's model/inputer/base_inputer.py:10-41)."""
from typing import Dict, List, Optional

import torch


class BaseInputer:
    output_single_sequence = True

    def __init__(self, ut, inputs, eh, **kwargs):
        self.ut = ut
        self.inputs: list = inputs
        self.eh = eh

    def get_vocabs(self) -> Optional[List]:
        return []

    def sample_rebuilder(self, sample: dict):
        raise NotImplementedError

    def get_mask(self, batched_samples: Dict[str, torch.Tensor]):
        raise NotImplementedError

    def get_embeddings(self, batched_samples: Dict[str, torch.Tensor]):
        raise NotImplementedError

    def __call__(self, sample: dict):
        return self.sample_rebuilder(sample)
