"""`sizer.py` of the MI355X path (mirror of the reference's sizer.py:42-92): list every trainable parameter of the
configured model with its shape and print the total in millions.  Frozen tables (pre-trained GloVe / BERT word
pieces) are not counted, as in the reference (`requires_grad` filter, sizer.py:56-58).

    python -m legommenders_amd.sizer --data config/data/synthetic.yaml --model config/model/naml.yaml \
        --embed config/embed/glove.yaml --hidden_size 256"""
from __future__ import annotations

import torch

from legommenders_amd.config_init import CommandInit, Obj
from legommenders_amd.trainer import build_model, load_world, seeding


class Sizer:
    """Builds the model only (no device kernels run, so this one tool works without a GPU)."""

    def __init__(self, config: Obj):
        self.config = config
        config.seed = int(config.seed or 2023)
        seeding(config.seed)
        self.world = load_world(config.data, config.seed)
        self.legommender, self.kind = build_model(config, self.world, torch.device("cpu"))

    def log(self, *a):
        print(*a, flush=True)

    def named_trainable(self):
        return [(k, p) for k, p in self.legommender.named_parameters() if p.requires_grad]

    def run(self):
        named = self.named_trainable()
        for name, p in named:
            self.log(name, tuple(p.shape))
        total = sum(p.numel() for _, p in named)
        self.log(f"Number of parameters: {total / 1e6:.2f}M")
        return total


def get_configurations(kwargs=None) -> Obj:
    return CommandInit(
        required_args=["data", "model"],
        default_args=dict(embed="config/embed/null.yaml", exp="config/exp/default.yaml", hidden_size=256,
                          item_hidden_size="${hidden_size}$", item_page_size=64, batch_size=64),
    ).parse(kwargs=kwargs)


if __name__ == "__main__":
    Sizer(config=get_configurations()).run()
