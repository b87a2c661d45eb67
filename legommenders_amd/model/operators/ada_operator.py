"""AdaOperator -- NAML user encoder (mirror of the reference's model/operators/ada_operator.py:8-38)."""
from legommenders_amd.loader.env import Env
from legommenders_amd.model.common.attention import AdditiveAttention
from legommenders_amd.model.inputer.concat_inputer import ConcatInputer
from legommenders_amd.model.operators.base_operator import BaseOperator, BaseOperatorConfig


class AdaOperatorConfig(BaseOperatorConfig):
    def __init__(self, additive_hidden_size: int = 256, **kwargs):
        super().__init__(**kwargs)
        self.additive_hidden_size = additive_hidden_size


class AdaOperator(BaseOperator):
    config_class = AdaOperatorConfig
    inputer_class = ConcatInputer
    config: AdaOperatorConfig

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.additive_attention = AdditiveAttention(embed_dim=self.config.input_dim,
                                                    hidden_size=self.config.additive_hidden_size)

    def forward(self, embeddings, mask=None, **kwargs):
        return self.additive_attention(embeddings, mask.to(Env.device))

    @property
    def output_dim(self):
        return self.config.input_dim
