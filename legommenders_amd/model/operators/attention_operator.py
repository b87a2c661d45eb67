"""AttentionOperator -- NRMS news / user encoder (mirror of the reference's
model/operators/attention_operator.py:9-59): nn.MultiheadAttention parameters, Linear, additive pool;
in/out projections on the MFMA GEMM core, per-head softmax core in `lego_mhsa_core_*`."""
from torch import nn

from legommenders_amd import functional as F_hip
from legommenders_amd.loader.env import Env
from legommenders_amd.model.common.attention import AdditiveAttention
from legommenders_amd.model.inputer.concat_inputer import ConcatInputer
from legommenders_amd.model.operators.base_operator import BaseOperator, BaseOperatorConfig


class AttentionOperatorConfig(BaseOperatorConfig):
    def __init__(self, num_attention_heads: int = 8, attention_dropout: float = 0.1,
                 additive_hidden_size: int = 256, **kwargs):
        super().__init__(**kwargs)
        self.num_attention_heads = num_attention_heads
        self.attention_dropout = attention_dropout
        self.additive_hidden_size = additive_hidden_size


class AttentionOperator(BaseOperator):
    config_class = AttentionOperatorConfig
    inputer_class = ConcatInputer
    config: AttentionOperatorConfig

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.multi_head_attention = nn.MultiheadAttention(
            embed_dim=self.config.input_dim, num_heads=self.config.num_attention_heads,
            dropout=self.config.attention_dropout, batch_first=True)
        self.linear = nn.Linear(self.config.input_dim, self.config.hidden_size)
        self.additive_attention = AdditiveAttention(embed_dim=self.config.hidden_size,
                                                    hidden_size=self.config.additive_hidden_size)

    def forward(self, embeddings, mask=None, **kwargs):
        mask = mask.to(Env.device)
        mha = self.multi_head_attention
        outputs = F_hip.multi_head_self_attention(
            embeddings, mask, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
            self.config.num_attention_heads, p=self.config.attention_dropout, training=self.training)
        linear_outputs = F_hip.linear(outputs, self.linear.weight, self.linear.bias)
        return self.additive_attention(linear_outputs, mask)
