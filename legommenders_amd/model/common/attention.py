"""AdditiveAttention (mirror of the reference's model/common/attention.py:10-38): same parameters
(`encoder.0.{weight,bias}`, `encoder.2.weight`), forward on the HIP tanh-GEMM + pooling kernels."""
import torch
from torch import nn

from legommenders_amd import functional as F_hip


class AdditiveAttention(nn.Module):
    def __init__(self, embed_dim, hidden_size):
        super().__init__()
        self.embed_dim = embed_dim
        self.hidden_size = hidden_size
        self.encoder = nn.Sequential(
            nn.Linear(self.embed_dim, self.hidden_size),
            nn.Tanh(),
            nn.Linear(self.hidden_size, 1, bias=False),
        )

    def forward(self, inputs: torch.Tensor, attention_mask: torch.Tensor = None) -> torch.Tensor:
        """inputs [B,L,D], attention_mask [B,L] -> [B,D]; exp() is un-stabilised and eps sits in the
        denominator only, exactly as the reference (attention.py:31-38)."""
        if attention_mask is None:
            attention_mask = torch.ones(inputs.shape[:2], dtype=torch.int32, device=inputs.device)
        return F_hip.additive_attention(inputs, attention_mask, self.encoder[0].weight, self.encoder[0].bias,
                                        self.encoder[2].weight)
