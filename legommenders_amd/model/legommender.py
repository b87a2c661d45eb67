"""Legommender (mirror of the reference's model/legommender.py:60-336): item content -> user content ->
predictor -> loss / scores, with the reference's phase semantics (`Env`), `state_dict` layout
(`embedding_vocab_table.*`, `item_op.*`, `user_op.*`) and `get_parameters()`.

Two execution routes, same kernels underneath:
  * plug-in route  -- `forward(batch)` with the reference's nested batch (or id-only batches + device item
    table): inputer -> operator -> predictor through `legommenders_amd.functional` (autograd Functions over
    the C ABI).  Any operator/predictor written against the reference's interface runs here.
  * engine route   -- after `attach_engine(...)`, `forward(batch)` with `{item_id:[B,C], history:[B,S],
    __clicks_mask__ or hist_len}` id tensors runs the ragged `NamlEngine` / `NrmsEngine` as ONE autograd node
    (no per-op Python), writing parameter gradients straight into `.grad`.
"""
from __future__ import annotations

from typing import List, Tuple

import os

import torch
from torch import nn

from legommenders_amd import functional as F_hip
from legommenders_amd.loader.env import Env
from legommenders_amd.model.lego_config import LegoConfig


def _flatten(x):
    """Shaper.transform (utils/shaper.py:92-106): [B,C,...] leaves -> [B*C,...]; returns (flat, B, C)."""
    if isinstance(x, dict):
        out, b, c = {}, None, None
        for k, v in x.items():
            out[k], b, c = _flatten(v)
        return out, b, c
    B, C = x.shape[:2]
    return x.reshape(B * C, *x.shape[2:]), B, C


class _EngineStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, cand, hist, hist_len, training, want_loss):
        eng = model.engine
        scores, loss = eng.forward(cand, hist, hist_len, training=training, with_loss=want_loss)
        ctx.model = model
        return (loss.view(()) + 0.0 * anchor) if want_loss else scores.clone()

    @staticmethod
    def backward(ctx, g):
        model = ctx.model
        G = {}
        for name, p in model.named_parameters():
            if p.requires_grad and name in model.engine.P:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                G[name] = p.grad
        # the upstream gradient of the loss stays on the device (float(g) would be a host sync in every backward)
        model.engine.backward(G, gloss=1.0, gloss_dev=g.detach().reshape(-1)[:1].to(torch.float32).contiguous())
        return torch.zeros_like(model._anchor), None, None, None, None, None, None


class Legommender(nn.Module):
    def __init__(self, config: LegoConfig):
        super().__init__()
        self.config = config
        self.user_operator_class = config.user_operator_class
        self.predictor_class = config.predictor_class
        self.use_neg_sampling = config.use_neg_sampling
        self.neg_count = config.neg_count
        self.eh = config.eh
        self.embedding_vocab_table = self.eh.vocab_table
        self.embedding_feature_table = self.eh.feature_table
        self.user_hub = config.user_ut
        self.item_hub = config.item_ut
        self.cm = config.cm
        self.flatten_mode = self.user_operator_class.flatten_mode
        self.item_op = config.item_operator
        self.user_op = config.user_operator
        self.predictor = config.predictor
        Env.set_lm_cache(False)
        if config.use_item_content and hasattr(self.item_op, "use_lm_cache"):      # legommender.py:104-107
            Env.set_lm_cache(bool(self.item_op.use_lm_cache()))
        self.loss_func = nn.CrossEntropyLoss() if self.use_neg_sampling else nn.BCEWithLogitsLoss()
        self.engine = None
        self._anchor = torch.zeros((), requires_grad=True)
        self.item_repr = None      # eval-time representation caches (reference: loader/cacher/*)
        self.user_repr = None

    # ------------------------------------------------------------------ engine route
    def attach_engine(self, tables, B: int, kind: str = None, heads: int = 8, seed: int = 2023):
        """Bind the ragged HIP engine to this module's own parameter storage."""
        from legommenders_amd.engine import NamlEngine, NrmsEngine
        from legommenders_amd.model.operators.cnn_operator import CNNOperator
        P = {k: v.data for k, v in self.state_dict(keep_vars=True).items()}
        C, S = self.neg_count + 1, self.user_hub.meta.features[self.cm.history_col].max_len
        kind = kind or ("naml" if isinstance(self.item_op, CNNOperator) else "nrms")
        eh_drop = getattr(self.embedding_vocab_table[self._token_vocab()], "dropout", None)
        p_proj = eh_drop.p if eh_drop is not None else 0.0
        if kind == "naml":
            self.engine = NamlEngine(P, tables, B, C, S, seed=seed, p_proj=p_proj, p_conv=self.item_op.dropout.p)
        else:
            glove = any(k.endswith("glove.embedding.weight") for k in P)
            self.engine = NrmsEngine(P, tables, B, C, S, heads=self.item_op.config.num_attention_heads, glove=glove,
                                     seed=seed, p_proj=p_proj, p_att=self.item_op.config.attention_dropout)
        self._anchor = torch.zeros((), requires_grad=True, device=Env.device)
        return self.engine

    def _token_vocab(self):
        col = self.config.item_inputs[0]
        return self.item_hub.meta.features[col].tokenizer.vocab.name

    def _engine_forward(self, batch):
        cand = batch[self.cm.item_col].to(Env.device, torch.int32).contiguous()
        hist = batch[self.cm.history_col].to(Env.device, torch.int32).contiguous()
        if "hist_len" in batch:
            hist_len = batch["hist_len"].to(Env.device, torch.int32).contiguous()
        else:
            hist_len = batch[self.cm.mask_col].to(Env.device).sum(1).to(torch.int32).contiguous()
        want_loss = not (Env.is_testing or (Env.is_evaluating and not Env.simple_dev))
        return _EngineStep.apply(self._anchor, self, cand, hist, hist_len, self.training, want_loss)

    # ------------------------------------------------------------------ id-only batches
    def attach_item_table(self, tables):
        """Device-resident item table (engine.ItemTables): lets `forward` accept id-only batches, the layout the
        device Resampler emits (the reference ships stacked per-item tensors instead, resampler.py:191-193)."""
        self.item_table = tables
        if Env.lm_cache and hasattr(self.item_op, "build_layer_cache"):
            self.item_op.build_layer_cache(self)          # cached-layer LM operators (once_operator.py:99-134)

    def expand_item_ids(self, ids: torch.Tensor):
        """ids [B,C] -> the nested `{input_ids, attention_mask}` batch the item inputer expects (index
        bookkeeping only: SimpleInputer.sample_rebuilder / ConcatInputer.sample_rebuilder vectorised)."""
        from legommenders_amd.model.inputer.concat_inputer import ConcatInputer
        tb = self.item_table
        ids = ids.to(Env.device).long()
        tcol, ccol = self.config.item_inputs[0], self.config.item_inputs[1]
        tok = tb.title_tok.long()[ids]                               # [B,C,T], -1 pads
        cat = tb.cat.long()[ids]
        inputer = self.item_op.inputer
        if not isinstance(inputer, ConcatInputer):
            return {"input_ids": {tcol: tok, ccol: cat.unsqueeze(-1)},
                    "attention_mask": {tcol: (tok >= 0).long(), ccol: torch.ones_like(cat).unsqueeze(-1)}}
        L, T = inputer.max_sequence_len, tok.shape[-1]
        tl = tb.title_len.long()[ids]
        ar = torch.arange(L, device=Env.device).view(1, 1, L)
        t_ids = torch.full((*ids.shape, L), Env.UNSET, dtype=torch.long, device=Env.device)
        t_ids[..., :T] = tok
        off = int(inputer.use_cls_token)
        if off:
            t_ids = torch.roll(t_ids, off, -1)
            t_ids[..., 0] = Env.UNSET
        sep = int(inputer.use_sep_token)
        c_pos = (tl + off + sep).unsqueeze(-1)
        c_ids = torch.where(ar == c_pos, cat.unsqueeze(-1).expand(*ids.shape, L), torch.full_like(t_ids, Env.UNSET))
        live = tl + off + 1 + 2 * sep
        out = {tcol: t_ids, ccol: c_ids}
        if inputer.vocab_activated:
            s_ids = torch.full_like(t_ids, Env.UNSET)
            if off:
                s_ids[..., 0] = inputer.CLS
            if sep:
                s_ids = torch.where((ar == (tl + off).unsqueeze(-1)) | (ar == (tl + off + 2).unsqueeze(-1)),
                                    torch.full_like(s_ids, inputer.SEP), s_ids)
            s_ids = torch.where(ar >= live.unsqueeze(-1), torch.full_like(s_ids, inputer.PAD), s_ids)
            out[inputer.vocab.name] = s_ids
        return {"input_ids": out, "attention_mask": (ar < live.unsqueeze(-1)).long()}

    # ------------------------------------------------------------------ plug-in route (reference control flow)
    def get_item_content(self, batch: dict, col: str):
        if self.item_repr is not None:                               # cached path (legommender.py:153-157)
            indices = batch[col].to(Env.device)
            return self.item_repr[indices.reshape(-1)].reshape(*indices.shape, -1)
        content = batch[col]
        if Env.lm_cache:                                             # legommender.py:166-169,187-188: ids go to the operator
            ids = content.to(Env.device)
            flat = ids.reshape(-1)
            page = self._item_page(flat.numel())
            order = None
            if getattr(self.item_op, "trim_pads", False) and flat.numel() > page:      # pages of similar live length (see BertOperator)
                order = torch.argsort(self.item_op.attention_mask[flat.long()].sum(1), stable=True)
                flat = flat[order]
            outs = [self.item_op(flat[s:s + page], mask=None) for s in range(0, flat.numel(), page)]
            rep = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
            if order is not None:
                rep = torch.zeros_like(rep).index_copy(0, order, rep)
            return rep.view(*ids.shape, -1)
        if isinstance(content, torch.Tensor):                        # id-only batch: expand through the item table
            content = self.expand_item_ids(content)
        item_content, B, C = _flatten(content)
        mask = self.item_op.inputer.get_mask(item_content)
        emb = self.item_op.inputer.get_embeddings(item_content)
        n = B * C
        page = self._item_page(n)
        order = None
        if getattr(self.item_op, "trim_pads", False) and n > page and isinstance(emb, torch.Tensor) and isinstance(mask, torch.Tensor):
            order = torch.argsort(mask.sum(1), stable=True)          # pages of similar live length: each is cut to its own longest
            emb, mask = emb[order], mask[order]                      # sequence inside the operator (BertOperator._trim)
        outs = []
        for s in range(0, n, page):                                  # item_page_size chunking (legommender.py:174-184)
            sl = slice(s, min(s + page, n))
            sub_e = {k: v[sl] for k, v in emb.items()} if isinstance(emb, dict) else emb[sl]
            sub_m = {k: v[sl] for k, v in mask.items()} if isinstance(mask, dict) else mask[sl]
            outs.append(self.item_op(sub_e, mask=sub_m))
        item_repr = outs[0] if len(outs) == 1 else torch.cat(outs, 0)
        if order is not None:
            item_repr = torch.zeros_like(item_repr).index_copy(0, order, item_repr)
        return item_repr.view(B, C, -1)

    def get_user_content(self, batch: dict):
        if self.user_repr is not None:
            return self.user_repr[batch[self.cm.user_col].to(Env.device)]
        if self.config.use_item_content and not self.flatten_mode:
            hist = batch[self.cm.history_col]
            if self.skip_pad_items and self.item_repr is None and isinstance(hist, torch.Tensor) and hist.dim() == 2:
                clicks = self._encode_live_history(hist, batch[self.cm.mask_col])
            else:
                clicks = self.get_item_content(batch, self.cm.history_col)
        else:
            hist = batch[self.cm.history_col]
            if isinstance(hist, torch.Tensor):                       # id-only batch: the sample the resampler's user inputer would have
                m = batch[self.cm.mask_col].to(Env.device).long()    # built (resampler.py:222-226) -- ids, pads UNSET, the clicks mask
                ids = hist.to(Env.device).long()                     # (the pad value is built on the id tensor: a bool / uint8 mask cannot hold -1)
                hist = {"input_ids": {self.cm.history_col: ids.masked_fill(m <= 0, Env.UNSET)}, "attention_mask": m}
            clicks = self.user_op.inputer.get_embeddings(hist)
        return self.user_op(clicks, mask=batch[self.cm.mask_col].to(Env.device))

    # The reference pads every history to `max_click_num` slots with item 0 and ENCODES the pads (resampler.py:222-223; the
    # user operator then masks them).  On an id-only batch the pads are known before anything is encoded: only the live slots
    # go through the item operator (for the BERT news encoder that is ~1 500 of the 3 520 items of a B = 64 batch), the pad slots
    # get zero vectors.  Exact: a masked slot has attention / pooling weight exactly 0 in the user operators, so neither the
    # output nor any gradient depends on what the slot holds.  LEGO_SKIP_PAD_ITEMS=0 encodes them as the reference does.
    skip_pad_items = os.environ.get("LEGO_SKIP_PAD_ITEMS", "1") != "0"
    # `item_page_size` (64 in config/model/bert-naml.yaml, trainer.py:311) chunks the item operator's calls to bound the
    # reference's GPU memory; a 64-item call is 2 k token rows -- too few to fill an MI355X (BERT-base: 175 impressions/s at 64,
    # 318 at 256), and 288 GB of HBM do not need the bound.  The chunking changes no value, so a configured page is raised to this
    # floor (LEGO_ITEM_PAGE_FLOOR=0: exactly the yaml's page).
    item_page_floor = int(os.environ.get("LEGO_ITEM_PAGE_FLOOR", "256"))

    def _item_page(self, n: int) -> int:
        page = int(self.config.item_page_size or 0)
        floor = max(self.item_page_floor, int(getattr(self.item_op, "page_floor", 0) or 0)) if self.item_page_floor else 0
        return max(page, floor) if page else n

    def _encode_live_history(self, hist: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        ids = hist.to(Env.device)
        B, S = ids.shape
        live = mask.to(Env.device).reshape(-1) != 0
        idx = live.nonzero(as_tuple=False).squeeze(1)
        col = self.cm.history_col
        out = None
        if idx.numel() > 0:
            vec = self.get_item_content({col: ids.reshape(-1)[idx].unsqueeze(1)}, col)[:, 0]           # [n_live, D]
            out = torch.zeros(B * S, vec.shape[-1], dtype=vec.dtype, device=vec.device).index_copy(0, idx, vec)
        else:
            out = torch.zeros(B * S, int(self.config.hidden_size), dtype=torch.float32, device=Env.device)
        return out.view(B, S, -1)

    # On an id-only batch the candidates and the live history slots are the same kind of thing -- item ids -- so ONE item-operator call
    # encodes both (round 4): for the BERT news encoder that is one 30 k-row pass through the blocks instead of a 24 k-row and a 6 k-row
    # one (the products lose efficiency below ~10 k rows, and every launch is paid once).  Values are those of the two-call form
    # (class attribute `one_item_call = False`); only the dropout streams are numbered differently.
    one_item_call = True

    def _one_call_ok(self, batch):
        col, hcol = self.cm.item_col, self.cm.history_col
        return (self.one_item_call and self.skip_pad_items and self.item_repr is None and self.user_repr is None
                and self.config.use_item_content and not self.flatten_mode and getattr(self, "item_table", None) is not None
                and isinstance(batch.get(col), torch.Tensor) and batch[col].dim() == 2
                and isinstance(batch.get(hcol), torch.Tensor) and batch[hcol].dim() == 2 and self.cm.mask_col in batch)

    def _encode_items_once(self, batch):
        col = self.cm.item_col
        cand = batch[col].to(Env.device)
        hist = batch[self.cm.history_col].to(Env.device)
        mask = batch[self.cm.mask_col].to(Env.device)
        B, C = cand.shape
        S = hist.shape[1]
        idx = (mask.reshape(-1) != 0).nonzero(as_tuple=False).squeeze(1)
        ids = torch.cat((cand.reshape(-1), hist.reshape(-1)[idx]))
        vec = self.get_item_content({col: ids.unsqueeze(1)}, col)[:, 0]                    # [B*C + n_live, D]
        items = vec[:B * C].reshape(B, C, -1)
        clicks = torch.zeros(B * S, vec.shape[-1], dtype=vec.dtype, device=vec.device).index_copy(0, idx, vec[B * C:]).view(B, S, -1)
        return items, self.user_op(clicks, mask=mask)

    def forward(self, batch: dict):
        if self.engine is not None and isinstance(batch[self.cm.item_col], torch.Tensor) \
                and batch[self.cm.item_col].dim() == 2 and self.item_repr is None \
                and isinstance(batch.get(self.cm.history_col), torch.Tensor):
            return self._engine_forward(batch)
        if isinstance(batch[self.cm.item_col], torch.Tensor) and batch[self.cm.item_col].dim() == 1:
            batch[self.cm.item_col] = batch[self.cm.item_col].unsqueeze(1)
        if self._one_call_ok(batch):
            item_embeddings, user_embeddings = self._encode_items_once(batch)
        elif self.config.use_item_content:
            item_embeddings = self.get_item_content(batch, self.cm.item_col)
            user_embeddings = self.get_user_content(batch)
        else:
            # ID-based models (config/model/naml_id.yaml, legommender.py:237-248): no item operator -- a candidate is the embedding of
            # its item id, looked up in the table of the HISTORY column's vocabulary; the clicked items go through the user inputer
            vocab_name = self.config.user_ut.meta.features[self.cm.history_col].tokenizer.vocab.name
            item_embeddings = self.eh(vocab_name, col_name=self.cm.history_col)(batch[self.cm.item_col].to(Env.device))
            user_embeddings = self.get_user_content(batch)
        if self.use_neg_sampling:
            scores = self._predict_for_neg_sampling(item_embeddings, user_embeddings)
            labels = torch.zeros(scores.size(0), dtype=torch.long, device=Env.device)
        else:
            scores = self.predictor(user_embeddings, item_embeddings.squeeze(1))
            labels = batch[self.cm.label_col].float().to(Env.device)
        if Env.is_testing or (Env.is_evaluating and not Env.simple_dev):
            return scores
        return self.loss_func(scores, labels)

    def _predict_for_neg_sampling(self, item_embeddings, user_embeddings):
        batch_size, candidate_size, hidden_size = item_embeddings.shape
        if self.predictor.keep_input_dim:
            return self.predictor(user_embeddings, item_embeddings)
        user_embeddings = self.user_op.prepare_for_predictor(user_embeddings, candidate_size)
        item_embeddings = item_embeddings.reshape(-1, hidden_size)
        return self.predictor(user_embeddings, item_embeddings).view(batch_size, -1)

    def get_parameters(self) -> Tuple[List[nn.Parameter], List[nn.Parameter]]:
        pretrained, other = [], []
        signals = self.item_op.get_pretrained_parameter_names() if self.item_op is not None else []      # ID-based models have no item operator
        for name, param in self.named_parameters():
            if not param.requires_grad:
                continue
            (pretrained if any(name.startswith(f"item_op.{s}") for s in signals) else other).append(param)
        return pretrained, other

    def __str__(self):
        return self.__class__.__name__

    __repr__ = __str__
