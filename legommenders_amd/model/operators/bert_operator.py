"""BertOperator -- BERT news encoder of config/model/bert-naml.yaml (mirror of the reference's
model/operators/bert_operator.py:10-52 on top of once_operator.py:23-193 and lm_operator.py:8-23).

What runs where: the word-piece table look-up (frozen, no projection: the table width equals the transformer width,
loader/embedding_hub.py:269-271) is the path's HIP row gather; the transformer blocks run through PyTorch-ROCm
(`transformers.BertModel(inputs_embeds=..., word_embeddings=None)`, once_operator.py:156-170 -- rocBLAS / hipBLASLt GEMMs);
`Linear(768 -> D)` and the additive pool after it (once_operator.py:190-193) are the path's own MFMA / pooling kernels.

Reproduced quirk: with the yaml default `tune_from: 0` the reference still slices `encoder.layer[1:]` (and pre-caches the
layer-0 states on disk, which nothing reads) while `forward` takes the whole-transformer branch because `not 0` is true
(once_operator.py:128-134,173-180) -- so the model that trains is BERT without its first block.  `tune_from > 0` (cached
hidden states gathered on the CPU per batch, once_operator.py:182-188) and LoRA (needs `peft`) are not built and say so."""
import abc
import os

from torch import nn

from legommenders_amd import functional as F_hip
from legommenders_amd.loader.env import Env
from legommenders_amd.model.common.attention import AdditiveAttention
from legommenders_amd.model.inputer.concat_inputer import ConcatInputer
from legommenders_amd.model.operators.ada_operator import AdaOperatorConfig
from legommenders_amd.model.operators.base_operator import BaseOperator


class OnceOperatorConfig(AdaOperatorConfig):
    def __init__(self, tune_from: int = 0, use_lora=True, lora_alpha=128, lora_r=32, lora_dropout=0.1,
                 transformer_config: dict = None, **kwargs):
        """`transformer_config` (not in the reference): a BertConfig dict for a randomly initialised transformer, used when no
        pretrained checkpoint is reachable (tests; the reference always loads `ModelInit.get(name)`)."""
        super().__init__(**kwargs)
        self.tune_from = tune_from
        self.use_lora = use_lora
        self.lora_alpha = lora_alpha
        self.lora_r = lora_r
        self.lora_dropout = lora_dropout
        self.transformer_config = transformer_config


class LMOperator(BaseOperator):
    """lm_operator.py:8-23"""

    @property
    def operator_name(self):
        return self.__class__.__name__.replace("Operator", "").lower()

    def use_lm_cache(self):
        raise NotImplementedError

    def get_layer_nums(self):
        raise NotImplementedError


class BertOperator(LMOperator, abc.ABC):
    config_class = OnceOperatorConfig
    inputer_class = ConcatInputer
    inputer: ConcatInputer
    config: OnceOperatorConfig

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.transformer = self._load_transformer()
        self.num_hidden_layers = self.get_layer_nums()
        if self.config.tune_from and self.config.tune_from < 0:
            self.config.tune_from = self.num_hidden_layers + self.config.tune_from
        self.linear = nn.Linear(self.config.input_dim, self.config.hidden_size)
        self.additive_attention = AdditiveAttention(embed_dim=self.config.hidden_size,
                                                    hidden_size=self.config.additive_hidden_size)
        self.transformer.embeddings.word_embeddings = None            # bert_operator.py:16: inputs_embeds only
        self._prepare_network()
        if self.transformer.config.hidden_size != self.config.input_dim:
            raise ValueError(f"In {self.classname}, hidden_size of transformer ({self.transformer.config.hidden_size}) "
                             f"does not match input_dim ({self.config.input_dim})")

    # ---- construction
    def _load_transformer(self):
        from transformers import AutoModel, BertConfig, BertModel     # deferred: the import takes ~20 s
        if self.config.transformer_config is not None:
            return BertModel(BertConfig(**self.config.transformer_config))
        from legommenders_amd.config_init import ModelInit
        key = ModelInit.get(self.operator_name)
        if key is None or not os.path.exists(str(key)):
            raise ValueError(f"{self.classname}: no local checkpoint for '{self.operator_name}' ({key!r}); this build has no "
                             f"network -- put `{self.operator_name} = <path>` into .model / set LEGO_MODEL_{self.operator_name.upper()}, "
                             f"or pass item_config.transformer_config for a random-init transformer")
        return AutoModel.from_pretrained(key)

    def use_lm_cache(self):
        return bool(self.config.tune_from)

    def get_layer_nums(self):
        return self.transformer.config.num_hidden_layers

    def _slice_transformer_layers(self):
        self.transformer.encoder.layer = self.transformer.encoder.layer[self.config.tune_from + 1:]

    def _prepare_network(self):
        if self.config.tune_from is not None:
            if self.config.tune_from > self.num_hidden_layers:
                raise ValueError(f"tune_from should be less than {self.num_hidden_layers}")
            if self.config.tune_from:
                raise NotImplementedError("tune_from > 0 trains on hidden states cached in cache/<data>/<name>/layer_k.npy "
                                          "(once_operator.py:99-126,182-188); only the whole-transformer branch is built")
            self._slice_transformer_layers()                          # also for tune_from == 0: see the module docstring
        if (self.config.tune_from is None or self.config.tune_from < self.num_hidden_layers - 1) and self.config.use_lora:
            raise NotImplementedError("use_lora needs `peft`, which this build does not ship; set item_config.use_lora: false "
                                      "(the bert-naml.yaml default)")

    def get_pretrained_parameter_names(self):
        return ["transformer"]

    # ---- forward (once_operator.py:173-193, tune_from falsy)
    def forward(self, embeddings, mask=None, **kwargs):
        mask = mask.to(Env.device)
        outputs = self.transformer(inputs_embeds=embeddings.float(), attention_mask=mask.float(),
                                   return_dict=True).last_hidden_state
        outputs = F_hip.linear(outputs.float().contiguous(), self.linear.weight, self.linear.bias)
        return self.additive_attention(outputs, mask)


class BertBaseOperator(BertOperator):
    pass


class BertLargeOperator(BertOperator):
    pass
