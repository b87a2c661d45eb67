"""CNNOperator -- NAML news encoder (mirror of the reference's model/operators/cnn_operator.py:9-67).
Same parameters (`cnn.{weight,bias}`, `linear.*`, `additive_attention.encoder.*`); Conv1d+ReLU+mask+Dropout,
the length-1-column Linear and the additive pool run on the HIP kernels."""
import torch
from torch import nn

from legommenders_amd import functional as F_hip
from legommenders_amd.loader.env import Env
from legommenders_amd.model.common.attention import AdditiveAttention
from legommenders_amd.model.inputer.simple_inputer import SimpleInputer
from legommenders_amd.model.operators.base_operator import BaseOperator, BaseOperatorConfig


class CNNOperatorConfig(BaseOperatorConfig):
    def __init__(self, kernel_size: int = 3, dropout: float = 0.1, additive_hidden_size: int = 256, **kwargs):
        super().__init__(**kwargs)
        self.kernel_size = kernel_size
        self.dropout = dropout
        self.additive_hidden_size = additive_hidden_size


class CNNOperator(BaseOperator):
    config_class = CNNOperatorConfig
    config: CNNOperatorConfig
    inputer_class = SimpleInputer

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if self.config.kernel_size != 3:
            raise ValueError("the MI355X conv kernel implements kernel_size=3 (config/model/naml.yaml:14)")
        self.cnn = nn.Conv1d(in_channels=self.config.input_dim, out_channels=self.config.hidden_size,
                             kernel_size=self.config.kernel_size, padding="same")
        self.linear = nn.Linear(self.config.input_dim, self.config.hidden_size)
        self.activation = nn.ReLU()
        self.dropout = nn.Dropout(self.config.dropout)
        self.additive_attention = AdditiveAttention(embed_dim=self.config.hidden_size,
                                                    hidden_size=self.config.additive_hidden_size)

    def forward(self, embeddings: dict, mask=None, **kwargs):
        output_list, output_mask = [], []
        for col in embeddings:
            embedding = embeddings[col]
            if embedding.size()[1] > 1:
                output = F_hip.conv3_relu_mask(embedding, mask[col].to(Env.device), self.cnn.weight, self.cnn.bias,
                                               p=self.dropout.p, training=self.training)
            else:
                output = F_hip.linear(embedding, self.linear.weight, self.linear.bias)
            output_list.append(output)
            output_mask.append(mask[col].to(Env.device))
        outputs = torch.cat(output_list, dim=1)
        mask = torch.cat(output_mask, dim=1)
        return self.additive_attention(outputs, mask)
