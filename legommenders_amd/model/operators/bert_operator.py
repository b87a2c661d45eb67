"""BertOperator -- BERT news encoder of config/model/bert-naml.yaml (mirror of the reference's
model/operators/bert_operator.py:10-52 on top of once_operator.py:23-193 and lm_operator.py:8-23).

What runs where: the word-piece table look-up (frozen, no projection: the table width equals the transformer width,
loader/embedding_hub.py:269-271) is the path's HIP row gather; the transformer blocks run through PyTorch-ROCm
(`transformers.BertModel(inputs_embeds=..., word_embeddings=None)`, once_operator.py:156-170 -- rocBLAS / hipBLASLt GEMMs);
`Linear(768 -> D)` and the additive pool after it (once_operator.py:190-193) are the path's own MFMA / pooling kernels.

Reproduced quirk: with the yaml default `tune_from: 0` the reference still slices `encoder.layer[1:]` (and pre-caches the
layer-0 states on disk, which nothing reads) while `forward` takes the whole-transformer branch because `not 0` is true
(once_operator.py:128-134,173-180) -- so the model that trains is BERT without its first block.

`tune_from = k > 0` (cached-layer mode, once_operator.py:99-134,182-188): hidden_states[k] of the checkpoint transformer
is computed once for every item (`build_layer_cache`, pages of `item_page_size`), the blocks `[k + 1:]` stay (block k is
skipped, as upstream) and training batches index the cache by item id (`Env.lm_cache`).  The reference keeps that cache as
a CPU tensor and copies the batch's rows to the device every step; here it lives in HBM ([n_items, L, H] fp32: 6.6 GB for
MIND-small at BERT-base width, of 288 GB) and the per-batch look-up is a device gather.  The on-disk layout
`cache/<data>/<operator>/layer_k.npy` + `mask.npy` is read when present and written after a build
(`LEGO_LAYER_CACHE_SAVE=0` skips the write), so caches made by the reference's splitter.py are usable.
LoRA (`use_lora: true`, once_operator.py:27-38,137-151 through `peft.get_peft_model(self.transformer.encoder, LoraConfig(r, lora_alpha,
lora_dropout))`, bert_operator.py:26-28): `peft` is not in this image, so its published algorithm is restated natively
(`LoraLinear`, `_LoraEncoder`): peft's default targets for model_type "bert" are the `query` and `value` projections of every kept
block; each becomes y = base(x) + (alpha / r) * B(A(dropout(x))) with A ~ kaiming_uniform(a = sqrt(5)), B = 0, every other encoder
parameter frozen; the module tree (and so the state_dict keys) is peft's: `encoder.base_model.model.layer.N.attention.self.query.
{base_layer.weight, base_layer.bias, lora_A.default.weight, lora_B.default.weight}`.  Parity for this piece is held against the
oracle's restatement only (no peft to generate a fixture from: stated in DESIGN.md)."""
import abc
import os

import numpy as np
import torch
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


class LoraLinear(nn.Module):
    """peft.tuners.lora.Linear restated: the frozen base layer plus a rank-r update, scaled by alpha / r."""

    def __init__(self, base: nn.Linear, r: int, alpha: int, dropout: float):
        super().__init__()
        import math
        self.base_layer = base
        self.r, self.scaling = int(r), float(alpha) / float(r)
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()})
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, self.r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(self.r, base.out_features, bias=False)})
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B["default"].weight)
        for p in base.parameters():
            p.requires_grad = False

    def forward(self, x):
        # the two rank-r products on the path's own MFMA kernels (torch.ops.lego_hip.linear, autograd registered)
        a = F_hip.linear(self.lora_dropout["default"](x).float().contiguous(), self.lora_A["default"].weight)
        return self.base_layer(x) + F_hip.linear(a, self.lora_B["default"].weight) * self.scaling


class _LoraModel(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model


class _LoraEncoder(nn.Module):
    """what `get_peft_model(encoder, LoraConfig(...))` returns, as far as the operator uses it: `.base_model.model` is the encoder
    (so parameters are named `base_model.model.layer...`), calls are forwarded to it."""

    def __init__(self, encoder, r, alpha, dropout, targets=("query", "value")):
        super().__init__()
        for p in encoder.parameters():
            p.requires_grad = False
        for block in encoder.layer:
            att = block.attention.self
            for t in targets:
                setattr(att, t, LoraLinear(getattr(att, t), r, alpha, dropout))
        self.base_model = _LoraModel(encoder)

    @property
    def layer(self):
        return self.base_model.model.layer

    def forward(self, *args, **kwargs):
        return self.base_model.model(*args, **kwargs)

    def trainable_parameters(self):
        t = sum(p.numel() for p in self.parameters() if p.requires_grad)
        return t, sum(p.numel() for p in self.parameters())


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
        self.hidden_weights = None                                    # [n_items, L, H] layer cache (device)
        self.attention_mask = None                                    # [n_items, L]
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
            if not self.config.tune_from:
                self._slice_transformer_layers()                      # also for tune_from == 0: see the module docstring
            # tune_from > 0: the cache needs every block and the item table -- `build_layer_cache` slices afterwards
        if not self.config.tune_from:
            self._lora_encoder()                 # tune_from > 0: after build_layer_cache has sliced the blocks

    def _lora_encoder(self):
        """once_operator.py:137-151 + bert_operator.py:26-28 (peft restated natively, see the module docstring)"""
        c = self.config
        if not ((c.tune_from is None or c.tune_from < self.num_hidden_layers - 1) and c.use_lora):
            return
        if not isinstance(c.lora_r, int):
            raise ValueError("lora_r should be an integer")
        if not isinstance(c.lora_alpha, int):
            raise ValueError("lora_alpha should be an integer")
        if not isinstance(c.lora_dropout, float):
            raise ValueError("lora_dropout should be a float")
        self.transformer.encoder = _LoraEncoder(self.transformer.encoder, c.lora_r, c.lora_alpha, c.lora_dropout)

    def get_pretrained_parameter_names(self):
        return ["transformer"]

    # ---- cached-layer mode (once_operator.py:72-126)
    @property
    def _cache_base_dir(self):
        return os.path.join("cache", str(getattr(Env, "data_name", None) or "data"), self.operator_name)

    def _get_cache_path(self, layer):
        return os.path.join(self._cache_base_dir, f"layer_{layer}.npy")

    def _get_mask_path(self):
        return os.path.join(self._cache_base_dir, "mask.npy")

    @torch.no_grad()
    def build_layer_cache(self, legommender):
        """hidden_states[tune_from] of every item, then `encoder.layer = layer[tune_from + 1:]` (once_operator.py:99-134).
        Called by `Legommender.attach_item_table`, i.e. before any optimiser collects the parameters."""
        k = int(self.config.tune_from)
        if not k or self.hidden_weights is not None:
            return
        dev = Env.device
        n_items = int(legommender.item_table.title_tok.shape[0])
        # files are trusted only in single-process runs or when they predate this launch on every rank; with WORLD_SIZE > 1 a
        # missing file on ANY rank's first look means some rank may be writing: then nobody reads, everybody computes
        multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
        have = os.path.exists(self._get_cache_path(k)) and os.path.exists(self._get_mask_path())
        if multi and torch.distributed.is_available() and torch.distributed.is_initialized():
            flag = torch.tensor([int(have)], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            have = bool(flag.item())
        if have:
            mask = torch.from_numpy(np.load(self._get_mask_path())).to(dev)
            hidden = torch.from_numpy(np.load(self._get_cache_path(k))).to(dev).float()
            hidden = hidden.view(*mask.shape[:2], hidden.shape[-1])
            if mask.shape[0] != n_items:
                raise ValueError(f"{self._get_cache_path(k)} holds {mask.shape[0]} items, the item table {n_items}")
        else:
            from legommenders_amd.model.legommender import _flatten
            was_training = self.transformer.training
            self.transformer.eval()                                   # from_pretrained hands the reference an eval-mode module
            page = int(self.lego_config.item_page_size or 512)
            hs, ms = [], []
            for s in range(0, n_items, page):
                ids = torch.arange(s, min(s + page, n_items), device=dev)[:, None]
                content, _, _ = _flatten(legommender.expand_item_ids(ids))
                m = self.inputer.get_mask(content).to(dev)
                e = self.inputer.get_embeddings(content)
                out = self.transformer(inputs_embeds=e.float(), attention_mask=m.float(), output_hidden_states=True,
                                       return_dict=True)
                hs.append(out.hidden_states[k].float())
                ms.append(m.long())
            hidden, mask = torch.cat(hs, 0).contiguous(), torch.cat(ms, 0).contiguous()
            self.transformer.train(was_training)
            # under torchrun every rank computes its own HBM-resident cache (identical checkpoint -> identical values); only
            # rank 0 writes the files, through a temporary name + os.replace, so no reader ever sees a half-written array
            if os.environ.get("LEGO_LAYER_CACHE_SAVE", "1") != "0" and int(os.environ.get("RANK", "0")) == 0:
                os.makedirs(self._cache_base_dir, exist_ok=True)
                for path, arr in ((self._get_cache_path(k), hidden), (self._get_mask_path(), mask)):
                    tmp = f"{path}.tmp{os.getpid()}.npy"
                    np.save(tmp, arr.cpu().numpy())
                    os.replace(tmp, path)
        nan_rows = torch.isnan(hidden).any(-1)                        # once_operator.py:116-124
        if bool(nan_rows.any()):
            hidden[nan_rows] = torch.rand_like(hidden[nan_rows])
            bad = nan_rows.any(-1)
            template = torch.zeros_like(mask[0])
            template[0] = 1
            mask[bad] = template
        self.hidden_weights, self.attention_mask = hidden, mask
        self._slice_transformer_layers()
        self._lora_encoder()                                          # once_operator.py:128-151: slice first, then the adapters
        self.transformer.to(dev)

    # The transformer blocks run on the path's own kernels over RAGGED rows (legommenders_amd/bert_native.py: MFMA products, the
    # head-dim-64 attention core, fused Dropout + residual + LayerNorm, GELU) unless the configuration is outside what they cover
    # (LoRA-wrapped projections, sequences wider than 64, a non-GELU activation): then -- and with LEGO_BERT_NATIVE=0 -- through the
    # `transformers` modules on PyTorch-ROCm, as SURVEY.md section 8f-2 first prescribed.
    native = os.environ.get("LEGO_BERT_NATIVE", "1") != "0"
    native_page_floor = int(os.environ.get("LEGO_BERT_PAGE_FLOOR", "4096"))

    def _native_ok(self, L):
        if not self.native:
            return False
        from legommenders_amd import bert_native
        return bert_native.supported(self.transformer, int(L)) is None

    @property
    def page_floor(self):
        """items per operator call the paging of Legommender.get_item_content is raised to: one call for the whole batch keeps the
        block products at tens of thousands of rows (a 256-item page is 4.7 k rows: each CU would stage a whole weight panel for 18
        of them); the saved activations of 1 600 items x 11 blocks are ~15 GB of 288"""
        return self.native_page_floor if self._native_ok(32) else 0

    def _loop_forward(self, hidden_states, attention_mask):
        """bert_operator.py:30-45: the kept blocks on cached states"""
        if self._native_ok(hidden_states.shape[1]):
            from legommenders_amd import bert_native
            return bert_native.encoder_forward(self.transformer, hidden_states, attention_mask, embed=False)
        ext = (1.0 - attention_mask[:, None, None, :].to(hidden_states.dtype)) * torch.finfo(hidden_states.dtype).min
        return self.transformer.encoder(hidden_states=hidden_states, attention_mask=ext, return_dict=True).last_hidden_state

    # The reference runs every item at the inputer's full sequence length (title cap + category: 31 positions for MIND) with the
    # pad positions masked.  The live positions of a ConcatInputer sequence are a PREFIX, so a call can drop the trailing
    # positions that are pads in ALL of its items: masked keys never enter a softmax and the pool after the blocks skips masked
    # positions, so the live outputs are the same numbers (up to the GEMM library's summation order for another row count).
    # `Legommender.get_item_content` sorts the items of a batch by live length before paging, so a page of 64 items is cut to its
    # own longest sequence: 18.5 instead of 31 positions on average.  `trim_pads = False`: the full length, as the reference.
    trim_pads = True

    @staticmethod
    def _trim(states, mask):
        # cut behind the LAST live position of any row (not the live count: a mask that is not a prefix -- another inputer, left
        # padding -- must not lose live tokens)
        pos = torch.arange(1, mask.shape[1] + 1, device=mask.device)
        live = int(((mask != 0) * pos).max().item()) if mask.numel() else 0
        live = max(1, min(int(mask.shape[1]), live))
        return states[:, :live], mask[:, :live]

    # ---- forward (once_operator.py:173-193, tune_from falsy)
    def forward(self, embeddings, mask=None, **kwargs):
        if self.config.tune_from:                                     # once_operator.py:182-188: `embeddings` are item ids
            if self.hidden_weights is None:
                raise RuntimeError(f"{self.classname}: tune_from = {self.config.tune_from} needs the layer cache -- "
                                   f"call Legommender.attach_item_table(...) first")
            indices = embeddings.to(Env.device).long().reshape(-1)
            mask = self.attention_mask[indices]
            states = self.hidden_weights[indices]
            if self.trim_pads:
                states, mask = self._trim(states, mask)
            outputs = self._loop_forward(states, mask.float())
        else:
            mask = mask.to(Env.device)
            if self.trim_pads and isinstance(embeddings, torch.Tensor):
                embeddings, mask = self._trim(embeddings, mask)
            if isinstance(embeddings, torch.Tensor) and self._native_ok(embeddings.shape[1]):
                from legommenders_amd import bert_native
                outputs = bert_native.encoder_forward(self.transformer, embeddings, mask, embed=True)
            else:
                outputs = self.transformer(inputs_embeds=embeddings.float(), attention_mask=mask.float(),
                                           return_dict=True).last_hidden_state
        outputs = F_hip.linear(outputs.float().contiguous(), self.linear.weight, self.linear.bias)
        return self.additive_attention(outputs, mask)


class BertBaseOperator(BertOperator):
    pass


class BertLargeOperator(BertOperator):
    pass
