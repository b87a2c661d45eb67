"""Host-side orchestration of the HIP hot path: one training / scoring step of the two-tower
recommender (`Legommender.forward` + backward, reference model/legommender.py:219-263) as a
sequence of liblego_hip.so calls over a pre-allocated HBM workspace.

Data layout (everything resident in HBM, nothing crosses PCIe inside a step):
  * item tables   title_tok i32[n_items,T] (-1 pad), title_len i32[n_items], cat i32[n_items]
  * per batch     cand i32[B,C], hist i32[B,S], hist_len i32[B]
  * ragged plan   built on device by lego_plan_batch: only LIVE item instances (B*C candidates +
                  sum(hist_len) clicked items) and only LIVE title tokens become rows; row counts
                  stay on the device (`counters`) and are read by the kernels, never by the host.
  * row spaces    token rows [0,R): X (gathered GloVe rows), H (projected), Y (conv output);
                  Y rows [R,R+NI): one category row per instance; T = tanh hidden of the additive
                  attention over all Y rows; item vectors [NI,D]; history instances of one user are
                  contiguous, so the user encoder pools straight over item-vector rows.
No torch op touches the data path; torch only owns the memory and the stream.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import LegoDropout, call

SITE_PROJ, SITE_CONV, SITE_ITEM_ATT, SITE_USER_ATT = 0, 1, 2, 3
RI_LIVE_BIT = 4        # csrc/common.hpp RI_LIVE


# Host side of a launch.  A training step is ~40 launches with ~190 pointer arguments; on a slow host core the enqueue time of the
# Python loop (0.45-0.76 ms per step across the pool's boxes, tools/cpu_bound.py) brushes the 0.63 ms the device needs, so the
# per-argument and per-launch Python work is kept minimal: pointers travel as plain ints (ctypes converts them against the
# binding's argtypes), the current stream is looked up through the raw (id, device, type) triple with ONE Stream object per
# triple (current_stream() builds a new object per call: 9 us), and its handle is read once per object.
def _ptr(t: Optional[torch.Tensor], off: int = 0):
    if t is None:
        return None
    return t.data_ptr() + off * t.element_size() if off else t.data_ptr()


_CUR_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def current_stream(index: Optional[int] = None) -> "torch.cuda.Stream":
    """current_stream(), one cached Stream object per raw stream"""
    raw = torch._C._cuda_getCurrentStream(torch.cuda.current_device() if index is None else index)
    s = _CUR_STREAMS.get(raw)
    if s is None:
        s = _CUR_STREAMS[raw] = torch.cuda.Stream(stream_id=raw[0], device_index=raw[1], device_type=raw[2])
    return s


def stream_handle(stream) -> int:
    """hipStream_t of a torch stream as an int (read once per Stream object)"""
    h = getattr(stream, "_lego_handle", None)
    if h is None:
        h = int(stream.cuda_stream)
        try:
            stream._lego_handle = h
        except AttributeError:
            pass
    return h


def _stream():
    return stream_handle(current_stream())


def _check(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.LegoHipError(f"{name} must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype or not t.is_contiguous():
        raise _lib.LegoHipError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


class ItemTables:
    """Device-resident item table (the reference's Resampler.item_cache, loader/resampler.py:113-126,
    kept as flat int32 arrays instead of a python list of per-item tensor dicts)."""

    def __init__(self, title_tok, title_len, cat, device):
        self.title_tok = torch.as_tensor(title_tok).to(device=device, dtype=torch.int32).contiguous()
        self.title_len = torch.as_tensor(title_len).to(device=device, dtype=torch.int32).contiguous()
        self.cat = torch.as_tensor(cat).to(device=device, dtype=torch.int32).contiguous()
        self.n_items, self.T = self.title_tok.shape


_STREAMS: Dict[tuple, "torch.cuda.Stream"] = {}


def shared_stream(dev, role: str):
    """The process-wide HIP stream of a role ("side0", "side1", "prefetch") on a device.  Every engine / TrainStep of the
    process uses the same few streams: the runtime maps streams onto a handful of hardware queues in creation order, and a
    second TrainStep with fresh streams (bench.py's secondary lines) was seen to get a side stream on the SAME queue as the
    main stream -- its NRMS step took 2.70 ms instead of 1.31.  Streams created once, first, keep the queues the stand-alone
    run has."""
    d = torch.device(dev)
    key = (d.index if d.index is not None else torch.cuda.current_device(), role)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(d)
    return _STREAMS[key]


class _Base:
    def __init__(self, params: Dict[str, torch.Tensor], tables: ItemTables, B: int, C: int, S: int,
                 seed: int = 2023):
        self.P = params
        # switches read once per engine (an os.environ look-up per launch site and step is host time the step does not have)
        self._serial = os.environ.get("LEGO_SERIAL") == "1"      # profiling aid (tools/prof_*.sh): every launch of a step on ONE stream
        self.tb = tables
        self.B, self.C, self.S, self.T = B, C, S, tables.T
        self.dev = tables.title_tok.device
        self.seed = seed
        self.step = 0
        self.NIc = B * (C + S)
        self.nb = B               # rows of the batch being processed (<= B: the short last batch of an epoch)
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.counters = torch.zeros(8, **i32)
        self.inst_item = torch.zeros(self.NIc, **i32)
        self.seg_off = torch.zeros(self.NIc + 1, **i32)
        self.hist_off = torch.zeros(B + 1, **i32)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.scores = torch.zeros(B, C, dtype=torch.float32, device=self.dev)

    def _f(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32, device=self.dev)

    @property
    def BC(self):
        """candidate instances come first in the plan's instance order; the clicked items start here"""
        return self.nb * self.C

    def set_batch(self, nb: int):
        """process batches of `nb` <= B rows from now on (workspaces stay sized for B; every launch takes its row count as
        an argument, so this is host state only)"""
        if not 0 < nb <= self.B:
            raise _lib.LegoHipError(f"batch of {nb} rows does not fit an engine built for B={self.B}")
        self.nb = int(nb)

    # ---- double-buffered plans: the (tiny, latency-bound) plan kernels of step N+1 run on a side stream while
    # step N computes, so they leave the critical path (TrainStep._prefetch)
    _PLAN_FIELDS = ("counters", "inst_item", "seg_off", "hist_off", "rowinfo", "row_tok")

    def enable_plan_slots(self):
        if getattr(self, "_slots", None) is None:
            cur = {k: getattr(self, k) for k in self._PLAN_FIELDS}
            self._slots = [cur, {k: torch.zeros_like(v) for k, v in cur.items()}]
        return self._slots

    def use_slot(self, s: int):
        for k, v in self._slots[s].items():
            setattr(self, k, v)

    def plan_on(self, stream, slot: int, cand, hist, hist_len, nb=None):
        """enqueue the ragged plan of the first `nb` rows of (cand, hist, hist_len) into plan slot `slot` on `stream`"""
        b = self._slots[slot]
        tok, tlen, width = self._plan_tables()
        call("lego_plan_batch", _ptr(cand), _ptr(hist), _ptr(hist_len), nb or self.B, self.C, self.S, _ptr(tok), _ptr(tlen), width,
             _ptr(b["counters"]), _ptr(b["inst_item"]), _ptr(b["seg_off"]), _ptr(b["hist_off"]), _ptr(b["rowinfo"]),
             _ptr(b["row_tok"]), stream_handle(stream))

    def _plan_tables(self):
        return self.tb.title_tok, self.tb.title_len, self.T

    timers = None   # set to a dict to time tagged kernels with HIP events on the launch stream (bench.py)
    timer_tags = None   # optional set: only these tags are bracketed (every event pair costs the step a few microseconds)

    def k(self, tag, name, *args):
        """call() with optional per-kernel HIP-event timing (events on torch's current stream, which is
        the stream every kernel of the step is launched on)."""
        t = self.timers
        if t is None or (self.timer_tags is not None and tag not in self.timer_tags):
            return call(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(name, *args)
        e1.record()
        t.setdefault(tag, []).append((e0, e1))

    @staticmethod
    def _sp(stream):
        return stream_handle(stream)

    def kk(self, stream, tag, name, *args):
        """launch on `stream`; with timers on, bracket the launch with HIP events on that same stream"""
        t = self.timers
        if t is None or tag is None or (self.timer_tags is not None and tag not in self.timer_tags):
            return call(name, *args, self._sp(stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        call(name, *args, self._sp(stream))
        e1.record(stream)
        t.setdefault(tag, []).append((e0, e1))

    def cnt(self, k):
        """device pointer to counters[k] (0 = token rows R, 1 = instances NI, 2 = R+NI, 3 = history instances)"""
        return _ptr(self.counters, k)

    def drop(self, p: float, site: int, training: bool):
        if not training or p <= 0.0:
            return None
        return ctypes.byref(LegoDropout(p, self.seed, site + 16 * self.step))

    # ------------------------------------------------------------------ evaluation caches
    # ItemCacher / UserCacher of the reference (loader/cacher/item_cacher.py:51-97, user_cacher.py:63-97) encode
    # every item once and every user once per evaluation; here each page is ONE ragged launch sequence.
    def item_vectors(self, ids: torch.Tensor) -> torch.Tensor:
        """engine built with B=1, S=0, C>=len(ids): vectors of the given item ids (eval mode)."""
        n = ids.numel()
        assert self.S == 0 and self.B == 1 and n <= self.C, "item_vectors needs an engine built as (B=1, C=page, S=0)"
        cand = torch.zeros(self.C, dtype=torch.int32, device=self.dev)
        cand[:n] = ids.to(self.dev, torch.int32)
        dummy = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._training = False
        self._plan(cand, dummy, dummy)
        self._eval_only = True                       # no backward follows: the de-duplicated path skips its row sort
        try:
            self._forward_items(False)
        finally:
            self._eval_only = False
        return self.items[:n].clone()

    def user_vectors(self, item_repr: torch.Tensor, hist: torch.Tensor, hist_len: torch.Tensor) -> torch.Tensor:
        """engine built with C=0: user vectors from cached item vectors (legommender.py:153-157,202-214)."""
        assert self.C == 0, "user_vectors needs an engine built with C=0"
        n = hist.shape[0]
        h = torch.zeros(self.B, self.S, dtype=torch.int32, device=self.dev)
        hl = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        h[:n] = hist.to(self.dev, torch.int32)
        hl[:n] = hist_len.to(self.dev, torch.int32)
        dummy = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._training = False
        self._plan(dummy, h, hl)
        D = self.D
        call("lego_gather_rows", _ptr(item_repr), D, D, _ptr(self.inst_item), self.NIc, self.cnt(1), _ptr(self.items), D, 0,
             _stream())
        self._forward_users(False)
        return self.user[:n].clone()

    # ---- the de-duplication scratch (stamp / rank tables over the vocabulary, sort keys and temporaries) exists ONCE per engine,
    # while plans are made on the prefetch stream (plan_on) and un-planned calls (item_vectors, forward(planned=False), bench's
    # stand-alone gather) use it on the current stream: a use on another stream than the previous one waits for that one's
    # event.  The steady training loop only ever uses the prefetch stream, so it never waits.
    _uq_last = None

    def _uq_begin(self, stream):
        last = self._uq_last
        if last is not None and last[0] != stream.cuda_stream:
            stream.wait_event(last[1])

    def _uq_end(self, stream):
        last = self._uq_last
        ev = last[1] if last is not None else torch.cuda.Event()
        ev.record(stream)
        self._uq_last = (stream.cuda_stream, ev)

    fused_grads = None      # set by bind_grads(): enables the fused user tower (forward writes its gradient partials)
    touched_rows = None     # uint8 [V] flags of the trainable token table's rows that have received a gradient (TrainStep)
    grad_hooks = None       # (dense_ready(), bucket_ready(lo_row, hi_row), rows_per_bucket): data-parallel TrainStep, trainable table

    def bind_grads(self, G):
        """Let training forwards write gradient partials directly (TrainStep binds its flat grad views once)."""
        self.fused_grads = G

    def grads_like(self):
        return {k: torch.zeros_like(v) for k, v in self.P.items() if v.dtype == torch.float32 and k not in self.frozen}


class NamlEngine(_Base):
    """NAML: frozen-GloVe gather -> Linear(E0->D) -> Conv1d(k3)+ReLU -> additive pool (items),
    additive pool over the click history (users), dot product + CE.  Parameter names are the
    reference's state_dict keys."""

    frozen = ("embedding_vocab_table.glove.embedding.weight",)

    def __init__(self, params, tables, B, C=5, S=50, seed=2023, p_proj=0.1, p_conv=0.1, token_rows=True):
        super().__init__(params, tables, B, C, S, seed)
        P = params
        self.E0 = P["embedding_vocab_table.glove.embedding.weight"].shape[1]
        self.D = P["embedding_vocab_table.glove.linear.weight"].shape[0]
        self.A = P["item_op.additive_attention.encoder.0.weight"].shape[0]
        self.Au = P["user_op.additive_attention.encoder.0.weight"].shape[0]
        self.p_proj, self.p_conv = p_proj, p_conv
        D, A, E0 = self.D, self.A, self.E0
        for k, v in P.items():
            _check(v, torch.float32, k)
        if D % 32:
            raise _lib.LegoHipError(f"hidden size {D} must be a multiple of 32 for the conv implicit GEMM")
        self.Rc = self.NIc * self.T if token_rows else 0     # token_rows=False: user-cache engine (item vectors are given)
        self.Ryc = self.Rc + self.NIc
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.rowinfo = torch.zeros(max(self.Rc, self.NIc * self.T), **i32)     # the plan kernel always writes rowinfo/row_tok
        self.row_tok = torch.zeros(max(self.Rc, self.NIc * self.T), **i32)
        self.inst_cat = torch.zeros(self.NIc, **i32)
        # Token de-duplication of the projection (csrc/misc_ops.hip, "unique tokens of a batch"): Linear(glove[tok]) depends on
        # the token id alone, so it is computed once per DISTINCT token of the batch (Xu -> Hu), expanded to the token rows
        # with the site's per-row dropout (H), and its weight gradient is formed from per-token sums of dH (dHu).  Exact up
        # to fp32 summation order; LEGO_DEDUP=0 keeps the row-by-row projection (X -> H).
        self.dedup = os.environ.get("LEGO_DEDUP", "1") != "0" and self.Rc > 0
        # ... and the weight gradient from per-token sums of dH: rows grouped by token with a radix sort on the prefetch stream
        # (lego_sort_rows), lego_segment_sum_rows (32 sorted rows per wave; groups inside a wave's rows are stored, groups that
        # span waves -- a Zipf head holds a fifth of the rows -- are added with 256-B-contiguous float atomics onto rows cleared
        # ahead of time on the side stream), then the product over the distinct tokens: 18 + 21 us against 55 for the product over
        # the token rows.  (First version: 16 rows per wave, 16-B-strided atomics, a counting sort with an atomic cursor: 46 + 25 us
        # and 2 x 64 us of serialised int atomics on the prefetch stream -- slower than not de-duplicating.)
        self.dedup_bwd = self.dedup
        V = P["embedding_vocab_table.glove.embedding.weight"].shape[0]
        self.V = V
        self.Uc = min(self.Rc, V) if self.dedup else 0
        if self.dedup:
            self.uq_stamp = torch.zeros(V, dtype=torch.int32, device=self.dev)       # uint32 epochs; never cleared
            self.uq_rank = torch.zeros(V, **i32)
            self.uq_bsum = torch.zeros((V + 1023) // 1024 + 1, **i32)
            self.uq_cnt = torch.zeros(self.Uc + 1, **i32)
            self.uq_start = torch.zeros(self.Uc + 1, **i32)
            self.uniq = torch.zeros(self.Uc, **i32)
            self.inv = torch.zeros(self.Rc, **i32)
            self.perm = torch.zeros(self.Rc, **i32)
            self.uq_keys = torch.zeros(self.Rc, **i32)
            self.keys_sorted = torch.zeros(self.Rc, **i32)                        # per plan slot, like perm
            self.uq_temp = torch.zeros(max(int(_lib.lib().lego_sort_rows_temp_bytes(self.Rc)), 256), dtype=torch.uint8, device=self.dev)
            self.Xu = self._f(self.Uc, E0)
            self.Hu = self._f(self.Uc, D)
            self.dHu = self._f(self.Uc, D)
            self._uq_epoch = 0
        self.X = self._f(1, E0) if self.dedup_bwd else self._f(self.Rc, E0)
        self.H = self._f(self.Rc, D)
        self.Y = self._f(self.Ryc, D)
        self.Tt = self._f(self.Ryc, A)
        self.wrow = self._f(self.Ryc)
        self.cat_emb = self._f(self.NIc, D)
        self.items = self._f(self.NIc, D)
        self.Tu = self._f(B * S, self.Au)
        self.wu = self._f(B * S)
        self.user = self._f(B, D)
        self.wt = self._f(3, D, D)
        # Winograd F(2,3) conv over row pairs (csrc/gemm_wino.hpp): two thirds of the direct conv's MFMA work.
        # LEGO_WINO=0 keeps the direct three-tap implicit GEMM (also used when D > 256).
        # The opt-in split-bf16 product mode (_lib.set_product_mode) covers the direct conv entry points only: there the forward and the
        # data gradient take the direct split form (54 / 74 us against Winograd's 92 / 92) while the WEIGHT gradient stays on the exact
        # Winograd kernel (`wino_dw`: 82 us alone against 95 for three split TN taps; it needs the pair plan, nothing of the forward's form)
        self.wino_dw = os.environ.get("LEGO_WINO", "1") != "0" and D <= 256 and self.Rc > 0
        self._built_mode = _lib.product_mode()       # (forward() refuses a step in the other mode: the conv entry points are chosen here)
        self.wino = self.wino_dw and self._built_mode == _lib.EXACT_F32
        self.Pc = self.NIc * ((self.T + 1) // 2)
        self.pair_info = torch.zeros(max(self.Pc, 1), **i32)
        self.mask_proj = torch.zeros(((self.Rc + 3) // 4) * D + 1, dtype=torch.uint8, device=self.dev)
        self.mask_conv = torch.zeros(((self.Rc + 3) // 4) * D + 1, dtype=torch.uint8, device=self.dev)
        self._slot_mask_step, self._mask_step = {}, -1
        self._slot_clean = {}
        self._loss2 = [self.loss, torch.zeros_like(self.loss)]          # training steps alternate (pre_forward clears the next one)
        self.wino_u = self._f(4, D, D)
        self.wino_ut = self._f(4, D, D)                   # the same sets transposed (data gradient)
        self.wino_slabs = _lib.lib().lego_conv3_wino_du_slabs(D, D, max(self.Pc, 1)) if self.wino_dw else 1
        self.wino_du = self._f(self.wino_slabs, 4, D, D)     # one partial result per k split of the weight gradient
        # backward workspace
        self.d_user = self._f(B, D)
        self.d_items = self._f(self.NIc, D)
        self.dY = self._f(self.Ryc, D)
        self.dH = self._f(self.Rc, D)
        self.d_cat_emb = self._f(self.NIc, D)
        self.dwt = self._f(3, D, D)

    # ------------------------------------------------------------------ prefetched token-row gather
    # The GloVe table is frozen, so the gathered rows X of a batch depend on its plan only: with plan slots enabled
    # (TrainStep) the gather of step N+1 runs right after its plan on the prefetch stream, off the critical path.
    _PLAN_FIELDS = _Base._PLAN_FIELDS + ("X", "pair_info", "mask_proj", "mask_conv", "inst_cat")
    _DEDUP_FIELDS = ("Xu", "uniq", "inv", "perm", "keys_sorted", "dHu")

    def enable_plan_slots(self):
        if getattr(self, "_slots", None) is None and self.dedup:
            self._PLAN_FIELDS = type(self)._PLAN_FIELDS + self._DEDUP_FIELDS
        return super().enable_plan_slots()

    # ---- dropout keep bits made ahead of time (lego_dropout_mask, same bits the epilogues would draw): the GEMM
    # epilogues of the step then read one byte per 4 rows x column instead of running Philox on the critical path
    def prefetch_masks(self, stream, slot):
        b = self._slots[slot]
        st = stream_handle(stream)
        for p, site, key in ((self.p_proj, SITE_PROJ, "mask_proj"), (self.p_conv, SITE_CONV, "mask_conv")):
            if p > 0.0 and self.Rc > 0:
                call("lego_dropout_mask", ctypes.byref(LegoDropout(p, self.seed, site + 16 * self.step, None)), self.Rc,
                     _ptr(b["counters"], 0), self.D, _ptr(b[key]), st)
        self._slot_mask_step[slot] = self.step

    def use_slot(self, s):
        super().use_slot(s)
        self._mask_step = self._slot_mask_step.get(s, -1)
        # the slot's per-token sums dHu are zero rows between its plan (plan_on clears them on the prefetch stream) and the first backward
        # pass that adds into them; a slot used again WITHOUT a new plan clears them in the forward pass (side stream), as an un-planned pass does
        self._cur_slot = s
        self._dhu_zeroed = bool(self.dedup_bwd and self._slot_clean.get(s, False))

    _cur_slot = None
    _pre_step = None        # (training step index, plan slot) whose parameter-dependent prologue ran at the end of the previous step (pre_forward)

    def pre_forward(self, slot):
        """TrainStep, right after Adam on the main stream: the part of the NEXT training step's forward pass that depends on the parameters
        and the slot's plan only -- the Winograd weight sets, the category rows of Y (embedding + Linear, cnn_operator.py:58-60) and the clear
        of the next loss accumulator.  Rounds 2-5 ran these on a side stream at the head of the forward pass; the main stream then waited for
        them twice (in front of the conv, in front of the additive product) and every cross-stream wait costs it 6-17 us even when the event
        has long been signalled (profiles/r05_timeline.txt: 12.6 + 17.4 us holes around the conv).  Here they cost the main stream their own
        ~20 us and no wait at all; the forward pass of a step prepared this way touches no side stream."""
        P, D = self.P, self.D
        b = self._slots[slot]
        m = current_stream()
        if self.wino:
            self.kk(m, None, "lego_conv3_wino_pack", _ptr(P["item_op.cnn.weight"]), _ptr(self.wino_u), _ptr(self.wino_ut), D, D)
        else:
            self.kk(m, None, "lego_conv3_pack", _ptr(P["item_op.cnn.weight"]), _ptr(self.wt), D, D)
        nxt = self._loss2[(self.step + 0) % 2]       # forward() of training step `self.step` accumulates into buffer step % 2
        nxt.zero_()
        self.kk(m, None, "lego_gather_rows", _ptr(P["embedding_vocab_table.category.weight"]), D, D, _ptr(b["inst_cat"]),
                self.NIc, _ptr(b["counters"], 1), _ptr(self.cat_emb), D, 0)
        self.kk(m, None, "lego_linear_fwd", _ptr(self.cat_emb), D, _ptr(P["item_op.linear.weight"]), D,
                _ptr(P["item_op.linear.bias"]), _ptr(self.Y), D, self.NIc, _ptr(b["counters"], 1), D, D, 0,
                None, None, None, _ptr(b["counters"], 0))
        self._pre_step = (self.step, slot)

    def drop(self, p, site, training):
        if not training or p <= 0.0:
            return None
        d = LegoDropout(p, self.seed, site + 16 * self.step, None)
        if site in (SITE_PROJ, SITE_CONV):
            if self._mask_step != self.step and self.Rc > 0:
                # a training pass whose keep bits were not drawn with the plan (an un-planned forward): drawn here, on the current
                # stream -- the Winograd kernels and the expansion read keep bits, they do not run Philox (round 6)
                st = _stream()
                for p_, site_, buf in ((self.p_proj, SITE_PROJ, self.mask_proj), (self.p_conv, SITE_CONV, self.mask_conv)):
                    if p_ > 0.0:
                        call("lego_dropout_mask", ctypes.byref(LegoDropout(p_, self.seed, site_ + 16 * self.step, None)), self.Rc, self.cnt(0),
                             self.D, _ptr(buf), st)
                self._mask_step = self.step
            d.mask = (self.mask_proj if site == SITE_PROJ else self.mask_conv).data_ptr()
        return ctypes.byref(d)

    def gather_tokens(self, stream=None, into=None, need_perm=True):
        """k1: X[r, :] = glove[row_tok[r], :] for the planned token rows (embedding_hub.py:95, frozen table)"""
        b = self.__dict__ if into is None else into
        s = current_stream() if stream is None else stream
        tag = "gather_rows_in_step" if stream is not None else "gather_rows"
        if self.dedup:
            # distinct tokens of the planned rows (uniq / inv / perm, U -> counters[6]), then ONE table row per distinct token
            self._uq_epoch = self._uq_epoch % 0x7FFFFFF0 + 1
            self._uq_begin(s)
            self.kk(s, None, "lego_unique_tokens", _ptr(b["row_tok"]), self.Rc, _ptr(b["counters"], 0), self.V, _ptr(self.uq_stamp),
                    self._uq_epoch, _ptr(self.uq_rank), _ptr(self.uq_bsum), _ptr(b["uniq"]), _ptr(b["inv"]), _ptr(self.uq_cnt),
                    _ptr(self.uq_start), None, _ptr(self.uq_keys) if self.dedup_bwd else None, _ptr(b["counters"], 6))
            self.kk(s, tag, "lego_gather_rows", _ptr(self.P["embedding_vocab_table.glove.embedding.weight"]), self.E0, self.E0,
                    _ptr(b["uniq"]), self.Uc, _ptr(b["counters"], 6), _ptr(b["Xu"]), self.E0, 0)
            if self.dedup_bwd and need_perm:         # rows grouped by distinct token: perm (the weight gradient sums dH per token).
                # Behind the table gather: the sort's ~10 small launches would push the gather into the backward's HBM-heavy phase
                self.kk(s, None, "lego_sort_rows", _ptr(self.uq_keys), self.Rc, _ptr(b["keys_sorted"]), _ptr(b["perm"]),
                        _ptr(self.uq_temp), self.uq_temp.numel())
            self._uq_end(s)
            if not self.dedup_bwd:                   # the weight gradient still runs over the token rows: X[r] = Xu[inv[r]]
                self.kk(s, "expand_rows_in_step" if stream is not None else None, "lego_expand_rows", _ptr(b["Xu"]), self.E0, _ptr(b["inv"]),
                        self.Rc, _ptr(b["counters"], 0), self.E0, None, None, None, 0, None, None, 0, None, _ptr(b["X"]), self.E0)
            return
        # with timers on (bench.py) the launch is bracketed by HIP events on the stream it runs on: on the prefetch stream
        # that is the gather's duration INSIDE the step, overlapped with the previous step's user-side chain
        self.kk(s, tag, "lego_gather_rows",
                _ptr(self.P["embedding_vocab_table.glove.embedding.weight"]), self.E0, self.E0,
                _ptr(b["row_tok"]), self.Rc, _ptr(b["counters"], 0), _ptr(b["X"]), self.E0, 0)

    def plan_pairs(self, stream=None, into=None):
        """row pairs of the Winograd conv from the plan's seg_off; the pair count lands in counters[5]"""
        b = self.__dict__ if into is None else into
        st = _stream() if stream is None else stream_handle(stream)
        call("lego_plan_pairs", _ptr(b["seg_off"]), self.NIc, _ptr(b["counters"], 1), _ptr(b["pair_info"]),
             _ptr(b["counters"], 5), st)

    def plan_on(self, stream, slot, cand, hist, hist_len, nb=None):
        super().plan_on(stream, slot, cand, hist, hist_len, nb)
        b = self._slots[slot]                        # category id of every planned instance: a function of the plan alone
        self.kk(stream, None, "lego_gather_i32", _ptr(self.tb.cat), _ptr(b["inst_item"]), self.NIc, _ptr(b["counters"], 1), _ptr(b["inst_cat"]))
        if self.Rc > 0:
            if self.wino_dw:
                self.plan_pairs(stream, self._slots[slot])
            self.gather_tokens(stream, self._slots[slot])
            if self.dedup_bwd:                       # the slot's per-token gradient sums start from zero rows: cleared here, off the main stream
                self.kk(stream, None, "lego_zero_rows", _ptr(b["dHu"]), self.D, self.D, self.Uc, _ptr(b["counters"], 6))
                self._slot_clean[slot] = True

    # ------------------------------------------------------------------ streams
    # Independent branches of the step run on side HIP streams so the small category / user-side kernels
    # and the weight-gradient GEMMs fill the CUs the big token-row GEMMs leave idle in their last wave.
    def _lanes(self):
        if getattr(self, "_side", None) is None:
            self._side = [shared_stream(self.dev, "side0"), shared_stream(self.dev, "side1")]
            self._evs = [torch.cuda.Event() for _ in range(10)]
        m = current_stream()
        if self._serial:                             # profiling aid: one stream, so per-kernel times do not overlap
            return m, m, m
        return m, self._side[0], self._side[1]

    def _fork(self, ev, src, *dst):
        dst = [d for d in dst if d is not src]
        if not dst:
            return
        ev.record(src)
        for d in dst:
            d.wait_event(ev)

    # ------------------------------------------------------------------ forward
    def forward(self, cand, hist, hist_len, training=False, with_loss=True, planned=False, gloss=1.0, fork_ev=None,
                neck_ev=None):
        """`fork_ev`: an event the caller has already recorded on the current stream after everything this forward
        depends on; None = record one here.  `neck_ev`: recorded behind the conv of the item tower: TrainStep starts the next batch's
        sample / plan / gather chain there.  Its first 56 us are small latency-bound kernels (sample, history, plan) that run beside
        the additive product and the pool; the HBM-heavy row gather then lands in the latency-bound user-side chain, where most CUs
        idle (20 us = 0.39 of the HBM roofline in the step), instead of beside the HBM-bound item pool backward (29 us = 0.27 when the
        chain started behind the pool); step time equal within noise (0.6765 vs 0.6739 ms over three alternating runs)."""
        P, B, C, S, D, A, E0 = self.P, self.nb, self.C, self.S, self.D, self.A, self.E0
        m, sb, sc = self._lanes()
        ev = self._evs
        _check(cand, torch.int32, "cand"); _check(hist, torch.int32, "hist"); _check(hist_len, torch.int32, "hist_len")
        if _lib.product_mode() != self._built_mode:
            raise _lib.LegoHipError("the product mode (lego_set_product_mode) changed after this engine was built: build engines in the mode they run in")
        self._training = training
        if not planned:
            self._plan(cand, hist, hist_len)
            fork_ev = None                           # the plan was enqueued after the caller's event
            self._dhu_zeroed, self._cur_slot = False, None
        if training:
            self.loss = self._loss2[self.step % 2]
        self._forward_items(training, fork_ev, zero_loss=True, gathered=planned and getattr(self, "_slots", None) is not None,
                            neck_ev=neck_ev)
        self._fused = bool(training and with_loss and self.fused_grads is not None)
        if self._fused:
            # tanh GEMM over the clicked-item rows, then ONE kernel for pool + dot + CE + their backward
            G = self.fused_grads
            self.kk(m, "additive_fwd_user", "lego_linear_fwd", _ptr(self.items, self.BC * D), D,
                    _ptr(P["user_op.additive_attention.encoder.0.weight"]), D,
                    _ptr(P["user_op.additive_attention.encoder.0.bias"]), _ptr(self.Tu), self.Au, B * S, self.cnt(3),
                    self.Au, D, 2, None, None, None, None)
            self.kk(m, None, "lego_user_tower_train", _ptr(self.Tu), self.Au, _ptr(self.items), D,
                    _ptr(P["user_op.additive_attention.encoder.2.weight"]), _ptr(self.hist_off), B, C, S, D, self.Au,
                    float(gloss) / B, _ptr(self.user), _ptr(self.scores), _ptr(self.loss), _ptr(self.d_items), D,
                    _ptr(G["user_op.additive_attention.encoder.2.weight"]),
                    _ptr(G["user_op.additive_attention.encoder.0.bias"]))
            self._fused_gloss = float(gloss)
        else:
            self._forward_users(training)
            # k11/k12: dot predictor + CE(label 0)
            self.kk(m, None, "lego_dot_ce_fwd", _ptr(self.user), D, _ptr(self.items), D, B, C, D, _ptr(self.scores),
                    _ptr(self.loss) if with_loss else None)
        self.step += 1 if training else 0
        return self.scores, self.loss

    def _plan(self, cand, hist, hist_len):
        m, _, _ = self._lanes()
        self.kk(m, None, "lego_plan_batch", _ptr(cand), _ptr(hist), _ptr(hist_len), self.nb, self.C, self.S,
                _ptr(self.tb.title_tok), _ptr(self.tb.title_len), self.T,
                _ptr(self.counters), _ptr(self.inst_item), _ptr(self.seg_off), _ptr(self.hist_off),
                _ptr(self.rowinfo), _ptr(self.row_tok))
        self.kk(m, None, "lego_gather_i32", _ptr(self.tb.cat), _ptr(self.inst_item), self.NIc, self.cnt(1), _ptr(self.inst_cat))
        if self.wino_dw:
            self.plan_pairs(m)

    def _forward_items(self, training, fork_ev=None, zero_loss=False, gathered=False, neck_ev=None):
        """item vectors of every planned instance -> self.items[0:NI]"""
        P, D, A, E0 = self.P, self.D, self.A, self.E0
        m, sb, sc = self._lanes()
        ev = self._evs
        if not gathered:                             # un-planned call: the token rows (and, de-duplicated, the distinct-token count in
            self.gather_tokens(need_perm=not getattr(self, "_eval_only", False))   # counters[6], read by the side chain's lego_zero_rows
        prepared = training and gathered and self._pre_step == (self.step, self._cur_slot)
        self._pre_step = None
        if prepared:
            # pre_forward() ran this step's parameter-dependent prologue at the end of the previous step, on this stream: no side chain, no waits
            if self.dedup:
                self.kk(m, "proj_fwd", "lego_linear_fwd", _ptr(self.Xu), E0, _ptr(P["embedding_vocab_table.glove.linear.weight"]), E0,
                        _ptr(P["embedding_vocab_table.glove.linear.bias"]), _ptr(self.Hu), D, self.Uc, self.cnt(6), D, E0, 0,
                        None, None, None, None)
                self.kk(m, "proj_expand", "lego_expand_rows", _ptr(self.Hu), D, _ptr(self.inv), self.Rc, self.cnt(0), D,
                        self.drop(self.p_proj, SITE_PROJ, training), None, None, 0, None, None, 0, None, _ptr(self.H), D)
            else:
                self.kk(m, "proj_fwd", "lego_linear_fwd", _ptr(self.X), E0, _ptr(P["embedding_vocab_table.glove.linear.weight"]), E0,
                        _ptr(P["embedding_vocab_table.glove.linear.bias"]), _ptr(self.H), D, self.Rc, self.cnt(0), D, E0, 0,
                        None, self.drop(self.p_proj, SITE_PROJ, training), None, None)
            if not self._dhu_zeroed and self.dedup_bwd:
                self.kk(m, None, "lego_zero_rows", _ptr(self.dHu), D, D, self.Uc, self.cnt(6))
                self._dhu_zeroed = True
            self._conv_fwd(m, training)
            if neck_ev is not None:
                neck_ev.record(m)
            self._additive_fwd(m, "item_op.", _ptr(self.Y), self.Ryc, self.cnt(2), self.Tt, A, self.seg_off, self.cnt(0),
                               self.NIc, self.cnt(1), self.items, self.wrow)
            return
        if fork_ev is not None and sb is not m:
            sb.wait_event(fork_ev)
        else:
            self._fork(ev[0], m, sb)
        # side stream, in the order the main stream needs things: the tap-major conv weights first (their own event:
        # the conv must not wait for the category GEMM), then the loss accumulator, then
        # k2/k4 category embedding + Linear on the length-1 column (cnn_operator.py:58-60) -> Y rows R..R+NI
        if self.wino:
            self.kk(sb, None, "lego_conv3_wino_pack", _ptr(P["item_op.cnn.weight"]), _ptr(self.wino_u), _ptr(self.wino_ut), D, D)
        else:
            self.kk(sb, None, "lego_conv3_pack", _ptr(P["item_op.cnn.weight"]), _ptr(self.wt), D, D)
        if sb is not m:
            ev[8].record(sb)
        if zero_loss:
            with torch.cuda.stream(sb):
                self.loss.zero_()
        if training and self.dedup_bwd and not self._dhu_zeroed:      # the per-token gradient sums start from zero rows: cleared here, on the
            self.kk(sb, None, "lego_zero_rows", _ptr(self.dHu), D, D, self.Uc, self.cnt(6))      # side chain the conv waits for anyway
            self._dhu_zeroed = True                                       # (a planned slot: cleared with its plan)
        self.kk(sb, None, "lego_gather_rows", _ptr(P["embedding_vocab_table.category.weight"]), D, D, _ptr(self.inst_cat),
                self.NIc, self.cnt(1), _ptr(self.cat_emb), D, 0)
        self.kk(sb, None, "lego_linear_fwd", _ptr(self.cat_emb), D, _ptr(P["item_op.linear.weight"]), D,
                _ptr(P["item_op.linear.bias"]), _ptr(self.Y), D, self.NIc, self.cnt(1), D, D, 0,
                None, None, None, self.cnt(0))
        if sb is not m:
            ev[1].record(sb)
        # main stream: k1 frozen GloVe row gather, then Transformation = Dropout(Linear(.)) (embedding_hub.py:95-96)
        if self.dedup:
            self.kk(m, "proj_fwd", "lego_linear_fwd", _ptr(self.Xu), E0, _ptr(P["embedding_vocab_table.glove.linear.weight"]), E0,
                    _ptr(P["embedding_vocab_table.glove.linear.bias"]), _ptr(self.Hu), D, self.Uc, self.cnt(6), D, E0, 0,
                    None, None, None, None)
            self.kk(m, "proj_expand", "lego_expand_rows", _ptr(self.Hu), D, _ptr(self.inv), self.Rc, self.cnt(0), D,
                    self.drop(self.p_proj, SITE_PROJ, training), None, None, 0, None, None, 0, None, _ptr(self.H), D)
        else:
            self.kk(m, "proj_fwd", "lego_linear_fwd", _ptr(self.X), E0, _ptr(P["embedding_vocab_table.glove.linear.weight"]), E0,
                    _ptr(P["embedding_vocab_table.glove.linear.bias"]), _ptr(self.H), D, self.Rc, self.cnt(0), D, E0, 0,
                    None, self.drop(self.p_proj, SITE_PROJ, training), None, None)      # every planned row is live
        if sb is not m:
            # the side chain is three short kernels (packed conv weights, category rows, category Linear: the category ids come
            # with the plan) and ends while the projection still runs: ONE wait here covers the conv's weights and the category
            # rows of Y -- every cross-stream wait costs the main stream 6-14 us of idle even when already signalled
            m.wait_event(ev[8])
        self._conv_fwd(m, training)
        if neck_ev is not None:
            neck_ev.record(m)                        # the next batch's prefetch chain starts here (see forward)
        if sb is not m:
            m.wait_event(ev[1])                      # category rows of Y
        # k5: additive attention pool over [title tokens..., category] (attention.py:31-38)
        self._additive_fwd(m, "item_op.", _ptr(self.Y), self.Ryc, self.cnt(2), self.Tt, A, self.seg_off, self.cnt(0),
                           self.NIc, self.cnt(1), self.items, self.wrow)

    def _conv_fwd(self, m, training):
        """k3: conv + relu + mask + dropout (cnn_operator.py:54-57)"""
        P, D = self.P, self.D
        if self.wino:
            self.kk(m, "conv3_fwd", "lego_conv3_wino_fwd", _ptr(self.H), D, _ptr(self.wino_u), _ptr(P["item_op.cnn.bias"]),
                    _ptr(self.pair_info), self.Pc, self.cnt(5), _ptr(self.Y), D, D, D,
                    self.drop(self.p_conv, SITE_CONV, training))
        else:
            self.kk(m, "conv3_fwd", "lego_conv3_fwd", _ptr(self.H), D, _ptr(self.wt), _ptr(P["item_op.cnn.bias"]), _ptr(self.rowinfo),
                    _ptr(self.Y), D, self.Rc, self.cnt(0), D, D, self.drop(self.p_conv, SITE_CONV, training), 0)

    def _forward_users(self, training):
        """k7: AdaOperator = additive pool over the clicked items of each user (ada_operator.py:31-34)"""
        m, _, _ = self._lanes()
        self._additive_fwd(m, "user_op.", _ptr(self.items, self.BC * self.D), self.nb * self.S, self.cnt(3), self.Tu, self.Au,
                           self.hist_off, None, self.nb, None, self.user, self.wu)

    def _additive_fwd(self, st, prefix, xp, rows_cap, rows_dyn, t, A, seg_off, extra, n_cap, n_dyn, out, wrow):
        P, D = self.P, self.D
        self.kk(st, "additive_fwd_" + prefix[:4], "lego_linear_fwd", xp, D, _ptr(P[prefix + "additive_attention.encoder.0.weight"]), D,
                _ptr(P[prefix + "additive_attention.encoder.0.bias"]), _ptr(t), A, rows_cap, rows_dyn, A, D, 2,
                None, None, None, None)
        self.kk(st, None, "lego_additive_pool_fwd", _ptr(t), A, xp, D, _ptr(P[prefix + "additive_attention.encoder.2.weight"]),
                _ptr(seg_off), None, extra, n_cap, n_dyn, D, A, _ptr(out), D, _ptr(wrow))

    # ------------------------------------------------------------------ backward
    def backward(self, G: Dict[str, torch.Tensor], gloss: float = 1.0, gloss_dev: Optional[torch.Tensor] = None, defer_join: bool = False):
        """Accumulates d(loss)/d(param) into G (reference key names); call after forward(training...).
        `defer_join`: leave the final join of the side stream to `finish_backward()` (TrainStep enqueues the next batch's prefetch chain in
        between: every launch of this pass's critical chain is then in its queue before the host turns to the ~23 small prefetch launches).
        `gloss_dev`: a one-element device tensor multiplied into the loss gradient ON THE DEVICE (autograd's upstream gradient:
        `Legommender`'s engine route passes it instead of reading it on the host, which would be a device sync per backward)."""
        P, B, C, S, D, A, E0 = self.P, self.nb, self.C, self.S, self.D, self.A, self.E0
        m, sb, sc = self._lanes()
        ev = self._evs
        training = self._training
        step_save = self.step
        if training:
            self.step -= 1          # regenerate the masks of the forward pass of this step
        hist_items = _ptr(self.items, self.BC * D)
        d_hist_items = _ptr(self.d_items, self.BC * D)
        if getattr(self, "_fused", False):
            if G is not self.fused_grads or abs(float(gloss) - self._fused_gloss) > 0 or gloss_dev is not None:
                raise _lib.LegoHipError("fused user tower: backward() must use the bound gradient buffers and the forward's gloss")
        else:
            self.kk(m, None, "lego_dot_ce_bwd", _ptr(self.user), D, _ptr(self.items), D, _ptr(self.scores), B, C, D,
                    float(gloss) / B, _ptr(gloss_dev), _ptr(self.d_user), D, _ptr(self.d_items), D)
            self._pool_bwd(m, "user_op.", G, hist_items, d_hist_items, self.Tu, self.Au, self.hist_off, None, B, None,
                           self.d_user, self.wu)
            self._pool_fold(m, "user_op.", G, self.Au)
        # main: user-side dx += dpre . W1   (the user dW1 product runs on the side stream after the ONE fork below:
        # every event recorded on the main stream costs it a few microseconds, tools/event_cost.py)
        self.kk(m, None, "lego_linear_bwd_data", _ptr(self.Tu), self.Au, _ptr(P["user_op.additive_attention.encoder.0.weight"]), D,
                d_hist_items, D, B * S, self.cnt(3), self.Au, D, 1, None, 0, 1.0, None, None, None, None, None)
        # item-side pool backward: dY direct part, dpre in place of T
        self._pool_bwd(m, "item_op.", G, _ptr(self.Y), _ptr(self.dY), self.Tt, A, self.seg_off, self.cnt(0), self.NIc,
                       self.cnt(1), self.d_items, self.wrow)
        keep = 1.0 / (1.0 - self.p_conv) if (training and self.p_conv > 0) else 1.0
        w1 = _ptr(P["item_op.additive_attention.encoder.0.weight"])
        self._fork(ev[4], m, sb)
        # ---- main: token rows: dY = relu'(.)*keep * (dY + dpre.W1); column sums -> conv bias grad (enqueued before the side
        # stream's launches so that the main chain's next kernel is in its queue first; measured equal to the other order)
        self.kk(m, "additive_bwd_data", "lego_linear_bwd_data", _ptr(self.Tt), A, w1, D, _ptr(self.dY), D, self.Rc, self.cnt(0), A, D, 1,
                _ptr(self.Y), D, keep, None, None, _ptr(G["item_op.cnn.bias"]), None, None)
        # ---- side stream B: fold of the pool backward's parameter-gradient copies, additive weight gradients, category branch
        self._pool_fold(sb, "item_op.", G, A)
        self.kk(sb, "additive_bwd_weight_user", "lego_linear_bwd_weight", _ptr(self.Tu), self.Au, hist_items, D,
                _ptr(G["user_op.additive_attention.encoder.0.weight"]), D, B * S, self.cnt(3), self.Au, D, None, None)
        self.kk(sb, None, "lego_linear_bwd_data", _ptr(self.Tt), A, w1, D, _ptr(self.dY), D, self.NIc, self.cnt(1), A, D, 1,
                None, 0, 1.0, None, None, _ptr(G["item_op.linear.bias"]), self.cnt(0), self.cnt(0))
        self.kk(sb, None, "lego_linear_bwd_weight", _ptr(self.dY), D, _ptr(self.cat_emb), D, _ptr(G["item_op.linear.weight"]), D,
                self.NIc, self.cnt(1), D, D, self.cnt(0), None)
        self.kk(sb, None, "lego_linear_bwd_data", _ptr(self.dY), D, _ptr(P["item_op.linear.weight"]), D, _ptr(self.d_cat_emb), D,
                self.NIc, self.cnt(1), D, D, 0, None, 0, 1.0, None, None, None, self.cnt(0), None)
        self.kk(sb, None, "lego_scatter_add_rows", _ptr(G["embedding_vocab_table.category.weight"]), D, D,
                G["embedding_vocab_table.category.weight"].shape[0], _ptr(self.inst_cat),
                self.NIc, self.cnt(1), _ptr(self.d_cat_emb), D)
        # the one LARGE side-stream product goes last: queued first, its ~1000 workgroups (4 x 32 KB of LDS per CU) took every CU
        # before the main stream's accumulate strip (123 KB of LDS) could start and left a 44 us hole in the main chain; behind
        # the small category-branch kernels it arrives when the strip is already running (0.705 -> 0.6985 ms/step, same-box A/B;
        # on the main stream instead -- after the strip or after the conv data gradient -- 0.72 ms)
        # (round 6, same box: this product on the MAIN stream behind the additive data gradient -- the conv kernels then run uncontended --
        # 0.5515 -> 0.564 ms; on the LDS-free tnd_kernel instead of the 64 x 64 tile kernel: 0.559 ms; profiles/r06_naml_tn_ab.txt)
        self.kk(sb, "additive_bwd_weight_item", "lego_linear_bwd_weight", _ptr(self.Tt), A, _ptr(self.Y), D,
                _ptr(G["item_op.additive_attention.encoder.0.weight"]), D, self.Ryc, self.cnt(2), A, D, None, None)
        # ---- main: conv data gradient -> projection weight gradient
        if self.wino:
            self.kk(m, "conv3_bwd_data", "lego_conv3_wino_bwd_data", _ptr(self.dY), D, _ptr(self.wino_u), _ptr(self.wino_ut), _ptr(self.pair_info),
                    self.Pc, self.cnt(5), _ptr(self.dH), D, D, D, self.drop(self.p_proj, SITE_PROJ, training),
                    _ptr(G["embedding_vocab_table.glove.linear.bias"]))
        else:
            self.kk(m, "conv3_bwd_data", "lego_conv3_bwd_data", _ptr(self.dY), D, _ptr(self.wt), _ptr(self.rowinfo), _ptr(self.dH), D,
                    self.Rc, self.cnt(0), D, D, self.drop(self.p_proj, SITE_PROJ, training),
                    _ptr(G["embedding_vocab_table.glove.linear.bias"]), 0)
        pst = m      # (the projection's weight-gradient tail beside the conv weight gradient on the side stream: 0.578 -> 0.601 ms)
        # ---- conv weight gradient after the data gradient on the main stream (a third stream measured 1-1.5 % slower)
        if self.wino_dw:
            self.kk(m, "conv3_bwd_weight", "lego_conv3_wino_bwd_weight", _ptr(self.dY), D, _ptr(self.H), D,
                    _ptr(self.pair_info), self.Pc, self.cnt(5), _ptr(self.wino_du), self.wino_slabs, D, D)
            self.kk(m, None, "lego_conv3_wino_unpack_add", _ptr(self.wino_du), self.wino_slabs, _ptr(G["item_op.cnn.weight"]), D, D)
        else:
            self.kk(m, "conv3_bwd_weight", "lego_conv3_bwd_weight", _ptr(self.dY), D, _ptr(self.H), D, _ptr(self.rowinfo),
                    _ptr(self.dwt), self.Rc, self.cnt(0), D, D)
            self.kk(m, None, "lego_conv3_unpack_add", _ptr(self.dwt), _ptr(G["item_op.cnn.weight"]), D, D)
        if self.dedup_bwd:                           # per-token sums of dH, then the product over the distinct tokens only
            self.kk(pst, "proj_bwd_segsum", "lego_segment_sum_rows", _ptr(self.dH), D, D, _ptr(self.perm), _ptr(self.inv), self.Rc,
                    _ptr(self.keys_sorted), self.cnt(0), _ptr(self.dHu), D, self.Uc, self.cnt(6),
                    0 if getattr(self, "_dhu_zeroed", False) else 1, None, None)
            self._dhu_zeroed = False
            if self._cur_slot is not None:
                self._slot_clean[self._cur_slot] = False
            self.kk(pst, "proj_bwd_weight", "lego_linear_bwd_weight", _ptr(self.dHu), D, _ptr(self.Xu), E0,
                    _ptr(G["embedding_vocab_table.glove.linear.weight"]), E0, self.Uc, self.cnt(6), D, E0, None, None)
        else:
            self.kk(pst, "proj_bwd_weight", "lego_linear_bwd_weight", _ptr(self.dH), D, _ptr(self.X), E0,
                    _ptr(G["embedding_vocab_table.glove.linear.weight"]), E0, self.Rc, self.cnt(0), D, E0, None, None)
        self.step = step_save
        if not defer_join:
            self.finish_backward()

    def finish_backward(self, join_ev=None):
        """the main stream's ONE wait at the end of a backward pass: for the side stream and -- `join_ev`, an event of ANOTHER stream (TrainStep:
        the next batch's plan on the prefetch stream) that the side stream is made to wait for first -- for whatever else the caller needs
        before the optimiser step and the next forward pass"""
        m, sb, _ = self._lanes()
        if join_ev is not None:
            (sb if sb is not m else m).wait_event(join_ev)
        self._fork(self._evs[6], sb, m)

    def _pool_bwd(self, st, prefix, G, x_ptr, dx_ptr, t, A, seg_off, extra, n_cap, n_dyn, gout, wrow):
        P, D = self.P, self.D
        self.kk(st, None, "lego_additive_pool_bwd", _ptr(t), A, x_ptr, D, _ptr(P[prefix + "additive_attention.encoder.2.weight"]),
                _ptr(seg_off), extra, n_cap, n_dyn, D, A, _ptr(gout), D, _ptr(wrow), dx_ptr, D,
                _ptr(G[prefix + "additive_attention.encoder.2.weight"]), _ptr(G[prefix + "additive_attention.encoder.0.bias"]),
                _ptr(self._pool_scratch(A, prefix)))

    def _pool_fold(self, st, prefix, G, A):
        """add the 32 scratch copies of the pool backward's parameter gradients into G (any stream ordered after it)"""
        self.kk(st, None, "lego_additive_pool_bwd_fold", _ptr(self._pool_scratch(A, prefix)), A,
                _ptr(G[prefix + "additive_attention.encoder.2.weight"]), _ptr(G[prefix + "additive_attention.encoder.0.bias"]))

    def _pool_scratch(self, A, key):
        """LEGO_POOL_SCRATCH(A) zeroed floats (32 copies of the two A-vectors the pool backward reduces into), one
        buffer per pooling site: a site's copies are folded later, possibly on another stream"""
        d = self.__dict__.setdefault("_pscr", {})
        if key not in d or d[key].numel() < 64 * A:
            d[key] = torch.zeros(64 * A, dtype=torch.float32, device=self.dev)
        return d[key]


class NrmsEngine(_Base):
    """NRMS: ConcatInputer sequence [title..., SEP, category, SEP] -> AttentionOperator (MHSA -> Linear ->
    additive pool) for items, AttentionOperator over the clicked item vectors for users, dot + CE
    (reference model/operators/attention_operator.py:46-59, model/inputer/concat_inputer.py:58-114).
    `glove=True`: frozen GloVe + Linear(E0->D) token embedding; `glove=False`: trainable [V,D] table
    (config/embed/null.yaml, dense-gradient semantics)."""

    def __init__(self, params, tables, B, C=5, S=50, heads=8, glove=False, seed=2023, p_proj=0.1, p_att=0.1,
                 token_rows=True, fold_linear=True):
        super().__init__(params, tables, B, C, S, seed)
        P = params
        self.glove, self.heads = glove, heads
        # AttentionOperator applies Linear straight after the attention's out-projection (attention_operator.py:49-56, nothing
        # between them): lin = (o Wo^T + bo) Wl^T + bl = o (Wl Wo)^T + (Wl bo + bl).  With fold_linear the two products over all
        # sequence rows become ONE in the forward pass, ONE data-gradient product and ONE weight-gradient product (T = d_lin^T o);
        # the four parameter gradients follow from T by D x D products (see _att_bwd).  Same function, one rounding fewer.
        # fold_linear = 2 (True) goes one step further: the additive attention reads `lin` only through a product (its hidden layer)
        # and a weighted sum (the pooled vector), so `lin` is never formed:
        #     t = tanh(o (W1 Wc)^T + (W1 bc + b1)),   out_i = (sum_r w_r o_r) Wc^T + bc
        # -- ONE product over the sequence rows in the forward pass (K = D, N = A) instead of three, one data-gradient product
        # instead of three, one weight-gradient product (Tp = dpre^T o) instead of three; every parameter gradient follows from
        # Tp, d_out^T pooled and two column sums by D x D / A x D products on the side stream.  The bias term uses sum_r w_r = 1;
        # AdditiveAttention's sum is S / (S + 2^-23) (attention.py:36), a relative deviation of 2^-23 / S in that term only.
        self.fold = 2 if fold_linear is True else int(fold_linear)
        self.frozen = ("embedding_vocab_table.glove.embedding.weight",) if glove else ()
        self.D = P["item_op.linear.weight"].shape[1]
        self.A = P["item_op.additive_attention.encoder.0.weight"].shape[0]
        self.E0 = P["embedding_vocab_table.glove.embedding.weight"].shape[1] if glove else self.D
        self.p_proj, self.p_att = (p_proj if glove else 0.0), p_att
        for k, v in P.items():
            _check(v, torch.float32, k)
        D, A = self.D, self.A
        # encoded per-item sequence table: token id >= 0, SEP = -2, category = -(3 + cat), pad = -1
        T = tables.T
        self.L = T + 3
        n = tables.n_items
        seq = torch.full((n, self.L), -1, dtype=torch.int32, device=self.dev)
        seq[:, :T] = tables.title_tok
        ar = torch.arange(self.L, device=self.dev)[None, :]
        tl = tables.title_len[:, None].to(torch.int64)
        seq = torch.where(ar < tl, seq, torch.full_like(seq, -1))
        rows = torch.arange(n, device=self.dev)
        tli = tables.title_len.to(torch.int64)
        seq[rows, tli] = -2
        seq[rows, tli + 1] = (-(3 + tables.cat)).to(torch.int32)
        seq[rows, tli + 2] = -2
        self.seq_tok = seq.contiguous()
        self.seq_len = (tables.title_len + 3).contiguous()
        self.Rc = self.NIc * self.L if token_rows else 0
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.rowinfo = torch.zeros(max(self.Rc, self.NIc * self.L), **i32)
        self.row_tok = torch.zeros(max(self.Rc, self.NIc * self.L), **i32)
        self.idx_tok = torch.zeros(self.Rc, **i32)
        self.idx_spec = torch.zeros(self.Rc, **i32)
        self.idx_cat = torch.zeros(self.Rc, **i32)
        self.tokinfo = torch.zeros(self.Rc, **i32)
        # GloVe variant: the projection of a token row depends on the token id alone (Dropout(Linear(frozen table)), embedding_hub.py:95-96),
        # so it runs once per DISTINCT token of the batch and is expanded to the sequence rows with the per-row dropout and the token
        # mask; its weight gradient comes from per-token sums of dE (NamlEngine has the same scheme).  The [SEP] / category positions
        # (row_tok < 0) fall into group 0 of the inverse map: the expansion writes them as zeros (tokinfo's live bit) and their dE
        # rows are zero when the sums are formed (lego_mask_dropout_rows runs first).  LEGO_NRMS_DEDUP=0: row-by-row projection.
        # Trainable table (embed/null): the same per-token sums feed the table's gradient -- the ~4.5 k DISTINCT rows of a batch are added
        # once each instead of 30.7 k sequence rows through float atomics (a Zipf head makes them queue on a few rows: 125 us per step).
        self.dedup = self.Rc > 0 and os.environ.get("LEGO_NRMS_DEDUP", "1") != "0"
        if self.dedup:
            V = P["embedding_vocab_table.glove.embedding.weight" if glove else "embedding_vocab_table.glove.weight"].shape[0]
            self.V, self.Uc = V, min(self.Rc, V)
            self.uq_stamp = torch.zeros(V, **i32)
            self.uq_rank = torch.zeros(V, **i32)
            self.uq_bsum = torch.zeros((V + 1023) // 1024 + 1, **i32)
            self.uq_cnt = torch.zeros(self.Uc + 1, **i32)
            self.uq_start = torch.zeros(self.Uc + 1, **i32)
            self.uniq = torch.zeros(self.Uc, **i32)
            self.inv = torch.zeros(self.Rc, **i32)
            self.perm = torch.zeros(self.Rc, **i32)
            self.uq_keys = torch.zeros(self.Rc, **i32)
            self.keys_sorted = torch.zeros(self.Rc, **i32)
            self.uq_temp = torch.zeros(max(int(_lib.lib().lego_sort_rows_temp_bytes(self.Rc)), 256), dtype=torch.uint8, device=self.dev)
            self.Xu = self._f(self.Uc if glove else 1, self.E0)
            self.Hu = self._f(self.Uc if glove else 1, D)
            self.dHu = self._f(self.Uc, D)
            self._uq_epoch = 0
            # keep bits of the projection's Dropout, drawn with the plan (lego_dropout_mask: the bits the in-kernel draw would take): the
            # expansion reads a byte per 4 rows x column, and the per-token sums apply the Dropout backward + the token mask while they
            # read dE -- the separate mask pass over dE (20 us) is gone
            self.mask_proj = torch.zeros((((self.Rc + 3) // 4) * D if glove else 0) + 4, dtype=torch.uint8, device=self.dev)
        self._slot_mask_step, self._mask_step = {}, -1
        self._slot_clean = {}
        # Trainable table (embed/null), round 4: EVERY position of a sequence row is a pure function of an id -- the token id, the [SEP]
        # id or the category id (ConcatInputer sums three look-ups of which exactly one is live per position, concat_inputer.py:96-114;
        # no Dropout sits between the table and the attention's in-projection when the table is not pre-trained) -- so the in-projection
        # q|k|v = E W_in^T + b is computed once per DISTINCT key of the batch (~4.5 k of 30.7 k rows) and expanded; its weight gradient and
        # its data gradient (= the three tables' gradients) are formed from per-key sums of d(qkv).  Key space: tokens [0, V), [SEP] -> V,
        # category c -> V + 1 + c.  Exact up to fp32 summation order.  LEGO_NRMS_QKV_DEDUP=0: row by row.
        self.qkv_dedup = self.dedup and not glove and os.environ.get("LEGO_NRMS_QKV_DEDUP", "1") != "0"
        if self.qkv_dedup:
            n_cat = P["embedding_vocab_table.category.weight"].shape[0]
            self.Vk = self.V + 1 + n_cat
            self.Uc = min(self.Rc, self.Vk)
            self.uq_stamp = torch.zeros(self.Vk, **i32)
            self.uq_rank = torch.zeros(self.Vk, **i32)
            self.uq_bsum = torch.zeros((self.Vk + 1023) // 1024 + 1, **i32)
            self.uq_cnt = torch.zeros(self.Uc + 1, **i32)
            self.uq_start = torch.zeros(self.Uc + 1, **i32)
            self.uniq = torch.zeros(self.Uc, **i32)
            self.dHu = self._f(self.Uc, D)                              # d(E) per distinct key
            self.row_key = torch.zeros(self.Rc, **i32)
            self.idx_tok_u, self.idx_spec_u = torch.zeros(self.Uc, **i32), torch.zeros(self.Uc, **i32)
            self.idx_cat_u, self.tokinfo_u = torch.zeros(self.Uc, **i32), torch.zeros(self.Uc, **i32)
            self.Eu = self._f(self.Uc, D)
            self.QKVu = self._f(self.Uc, 3 * D)
            self.dQKVu = self._f(self.Uc, 3 * D)
        # GloVe projection, round 5: the same per-key in-projection WITH the Dropout that sits between the projection and the attention
        # (embedding_hub.py:95-96): a token row is s (m_r . h_k), so W E_r = s (W h_k - sum over the dropped coordinates c of h_k[c] W[:, c])
        # -- the per-key product plus ~26 multiply-adds of a 3D-vector per row (csrc/dropcorr_ops.hip); the data gradient splits the same
        # way.  The weight gradient stays the dense product over the rows (its correction is a 10 %-dense sparse product), so the row
        # embeddings E are still expanded -- on the side stream, for the backward pass only.  Needs the site's keep bits ahead of time
        # (TrainStep draws them with the plan); a step without them, or LEGO_NRMS_DROPCORR=0, runs the in-projection row by row.
        self.dropcorr = self.dedup and glove and D <= 256 and os.environ.get("LEGO_NRMS_DROPCORR", "1") != "0"
        # (only the FORWARD in-projection runs per key; the backward pass is the row form over the same key space -- its per-key form
        # measured 620 us against 130, DESIGN.md section 11.4, and was removed in round 6)
        if self.dropcorr:
            n_cat = P["embedding_vocab_table.category.weight"].shape[0]
            self.Vk = self.V + 1 + n_cat
            self.Uc = min(self.Rc, self.Vk)
            self.uq_stamp = torch.zeros(self.Vk, **i32)
            self.uq_rank = torch.zeros(self.Vk, **i32)
            self.uq_bsum = torch.zeros((self.Vk + 1023) // 1024 + 1, **i32)
            self.uq_cnt = torch.zeros(self.Uc + 1, **i32)
            self.uq_start = torch.zeros(self.Uc + 1, **i32)
            self.uniq = torch.zeros(self.Uc, **i32)
            self.row_key = torch.zeros(self.Rc, **i32)
            self.idx_tok_u, self.idx_spec_u = torch.zeros(self.Uc, **i32), torch.zeros(self.Uc, **i32)
            self.idx_cat_u, self.tokinfo_u = torch.zeros(self.Uc, **i32), torch.zeros(self.Uc, **i32)
            self.Xu = self._f(self.Uc, self.E0)
            self.Hu = self._f(self.Uc, D)
            self.dHu = self._f(self.Uc, D)                              # d(Eu): per-key gradient rows
            self.Eu = self._f(self.Uc, D)
            self.QKVu = self._f(self.Uc, 3 * D)
            self.dQKVu = self._f(self.Uc, 3 * D)
            self.WinT = self._f(D, 3 * D)
            self.iota_u = torch.arange(self.Uc, **i32)
        self.keyspace = self.qkv_dedup or self.dropcorr
        self.X = (self._f(1, self.E0) if self.dedup else self._f(self.Rc, self.E0)) if glove else None
        self.E = self._f(self.Rc, D)
        self.dE = self._f(self.Rc, D)
        self.items = self._f(self.NIc, D)
        self.d_items = self._f(self.NIc, D)
        self.user = self._f(B, D)
        self.d_user = self._f(B, D)
        self.item_ws = self._att_ws(self.Rc, self.L, max(self.NIc, 1))
        self.user_ws = self._att_ws(B * S, S, B)
        # segments of more than 32 rows (an item whose title fills all T positions, a user with more than 32 clicks), listed with the
        # plan: the long-segment attention launch gives each (segment, head) pair of the list its own workgroup (csrc/mhsa.hip)
        self.long_items = torch.zeros(max(self.NIc, 1), **i32)
        self.long_users = torch.zeros(max(B, 1), **i32)
        self.long_cnt = torch.zeros(2, **i32)

    # with the plan (TrainStep's prefetch stream): the decoded index rows and, GloVe variant, the gathered token rows X -- the table is
    # frozen, so the 20 us gather and the 6 us decode of batch N+1 run beside batch N instead of at the head of its own step
    _PLAN_FIELDS = _Base._PLAN_FIELDS + ("idx_tok", "idx_spec", "idx_cat", "tokinfo", "long_items", "long_users", "long_cnt")

    def enable_plan_slots(self):
        if getattr(self, "_slots", None) is None and (self.glove or self.dedup):
            self._PLAN_FIELDS = NrmsEngine._PLAN_FIELDS + (
                (("Xu", "mask_proj") if self.glove else ()) + ("uniq", "inv", "perm", "keys_sorted", "dHu") if self.dedup else ("X",)) + (
                ("idx_tok_u", "idx_spec_u", "idx_cat_u", "tokinfo_u") if self.keyspace else ())
        return super().enable_plan_slots()

    def prefetch_masks(self, stream, slot):
        """TrainStep, after plan_on: the projection site's keep bits of the coming training step (engine.step) into the slot"""
        if self.dedup and self.glove and self.p_proj > 0.0:
            b = self._slots[slot]
            call("lego_dropout_mask", ctypes.byref(LegoDropout(self.p_proj, self.seed, SITE_PROJ + 16 * self.step, None)), self.Rc,
                 _ptr(b["counters"], 0), self.D, _ptr(b["mask_proj"]), stream_handle(stream))
            self._slot_mask_step[slot] = self.step

    def use_slot(self, s):
        super().use_slot(s)
        self._mask_step = self._slot_mask_step.get(s, -1)
        # the slot's per-token sums dHu are zero rows only between its plan (plan_on -> _decode_gather clears them) and the first
        # backward that adds into them: a slot used again WITHOUT a new plan (a replayed batch) must clear them in the sums' launch
        self._cur_slot = s
        self._dhu_zeroed = bool(self.dedup and self._slot_clean.get(s, False))

    _dhu_zeroed = False
    _cur_slot = None
    _dc_active = False
    _dc_drop = None
    _dc_e_pending = False

    def _dhu_consumed(self):
        """backward has added into dHu: the current slot's rows are not clean any more"""
        self._dhu_zeroed = False
        if self._cur_slot is not None:
            self._slot_clean[self._cur_slot] = False

    def drop(self, p, site, training):
        if not training or p <= 0.0:
            return None
        d = LegoDropout(p, self.seed, site + 16 * self.step, None)
        if site == SITE_PROJ and self.dedup and self._mask_step == self.step:
            d.mask = self.mask_proj.data_ptr()
        return ctypes.byref(d)

    def plan_on(self, stream, slot, cand, hist, hist_len, nb=None):
        super().plan_on(stream, slot, cand, hist, hist_len, nb)
        self._long_lists(self._slots[slot], stream_handle(stream), nb)
        if self.Rc > 0:
            self._decode_gather(self._slots[slot], stream)
            self._slot_clean[slot] = bool(self.dedup)

    def _long_lists(self, b, st, nb=None):
        if self.L > 32 and self.Rc > 0:
            call("lego_mhsa_long_segments", _ptr(b["seg_off"]), self.NIc, _ptr(b["counters"], 1), _ptr(b["long_items"]), _ptr(b["long_cnt"], 0), st)
        if self.S > 32:
            call("lego_mhsa_long_segments", _ptr(b["hist_off"]), (nb or self.B) if b is not self.__dict__ else self.nb, None, _ptr(b["long_users"]), _ptr(b["long_cnt"], 1), st)

    def _decode_gather(self, b, stream):
        st = stream_handle(stream)
        call("lego_nrms_decode_rows", _ptr(b["row_tok"]), self.Rc, _ptr(b["counters"], 0), _ptr(b["idx_tok"]), _ptr(b["idx_spec"]),
             _ptr(b["idx_cat"]), _ptr(b["tokinfo"]), st)
        if self.dedup:
            E0 = self.E0
            self._uq_epoch = self._uq_epoch % 0x7FFFFFF0 + 1
            self._uq_begin(stream)
            keys, nkeys = b["row_tok"], self.V
            if self.keyspace:                        # every position gets a key >= 0: tokens as they are, [SEP] (-2) -> V, category -(3+c) -> V+1+c
                call("lego_nrms_key_rows", _ptr(b["row_tok"]), self.Rc, _ptr(b["counters"], 0), self.V, _ptr(self.row_key), st)
                keys, nkeys = self.row_key, self.Vk
            call("lego_unique_tokens", _ptr(keys), self.Rc, _ptr(b["counters"], 0), nkeys, _ptr(self.uq_stamp), self._uq_epoch,
                 _ptr(self.uq_rank), _ptr(self.uq_bsum), _ptr(b["uniq"]), _ptr(b["inv"]), _ptr(self.uq_cnt), _ptr(self.uq_start), None,
                 _ptr(self.uq_keys), _ptr(b["counters"], 6), st)
            if self.keyspace:                        # the distinct keys back into per-table row indices (-1 = not this table's)
                call("lego_nrms_decode_keys", _ptr(b["uniq"]), self.Uc, _ptr(b["counters"], 6), self.V, _ptr(b["idx_tok_u"]), _ptr(b["idx_spec_u"]),
                     _ptr(b["idx_cat_u"]), _ptr(b["tokinfo_u"]), st)
            if self.glove:                           # table rows of the distinct tokens (key space: zero rows for the [SEP] / category keys)
                call("lego_gather_rows", _ptr(self.P["embedding_vocab_table.glove.embedding.weight"]), E0, E0,
                     _ptr(b["idx_tok_u"] if self.keyspace else b["uniq"]), self.Uc, _ptr(b["counters"], 6), _ptr(b["Xu"]), E0, 0, st)
            call("lego_sort_rows", _ptr(self.uq_keys), self.Rc, _ptr(b["keys_sorted"]), _ptr(b["perm"]), _ptr(self.uq_temp),
                 self.uq_temp.numel(), st)
            self._uq_end(stream)
            if self.qkv_dedup:                       # (its backward pass clears dEu in the sums' launch)
                pass
            elif b is not self.__dict__:             # a plan slot: its per-token sums start from rows cleared here, off the main stream
                call("lego_zero_rows", _ptr(b["dHu"]), self.D, self.D, self.Uc, _ptr(b["counters"], 6), st)
        elif self.glove:
            E0 = self.E0
            call("lego_gather_rows", _ptr(self.P["embedding_vocab_table.glove.embedding.weight"]), E0, E0, _ptr(b["idx_tok"]),
                 self.Rc, _ptr(b["counters"], 0), _ptr(b["X"]), E0, 0, st)

    # how the attention core's backward gets the softmax: recomputed from Q, K and one log-sum-exp per (row, head) (no [rows, heads, L]
    # tensor: 64 MB less HBM traffic per step at the bench shape) or read back from probabilities the forward pass saved.  Same-box
    # A/B in DESIGN.md section 4: the saved probabilities are faster (the kernel is issue-bound, not HBM-bound); set this attribute before
    # building an engine to trade 122 MB of workspace for the recomputing form (the plug-in route always recomputes)
    mhsa_recompute = False

    def _att_ws(self, rows, Lmax, n_seg):
        D, A, H = self.D, self.A, self.heads
        # the four parameter-space sums of a folded backward pass share ONE buffer: cleared by one memset (`.zero_()` = a 1.6 us fill
        # kernel; the multi-tensor zero launch it replaces sat 13-47 us on the side stream next to the attention core, and the weight
        # gradient behind it then ran into the long-segment attention launch)
        zs = self._f(A * D + D * D + D + A)
        Tp, T = zs[:A * D].view(A, D), zs[A * D:A * D + D * D].view(D, D)
        s_, sp = zs[A * D + D * D:A * D + D * D + D], zs[A * D + D * D + D:]
        return dict(zsum=zs, Tp=Tp, T=T, s=s_, sp=sp, **self._att_ws_rest(rows, Lmax, n_seg))

    def _att_ws_rest(self, rows, Lmax, n_seg):
        D, A, H = self.D, self.A, self.heads
        return dict(rows=rows, Lmax=Lmax, qkv=self._f(rows, 3 * D), o=self._f(rows, D), att=self._f(rows, D),
                    lin=self._f(rows, D), t=self._f(rows, A), wrow=self._f(rows),
                    lse=self._f(rows, H) if self.mhsa_recompute else None, probs=None if self.mhsa_recompute else self._f(rows, H, Lmax),
                    d_lin=self._f(rows, D), d_att=self._f(rows, D), d_o=self._f(rows, D), d_qkv=self._f(rows, 3 * D),
                    seg_live=torch.zeros(max(n_seg, 1), dtype=torch.int32, device=self.dev),
                    Wc=self._f(D, D), bc=self._f(D), U=self._f(D, D),
                    W2=self._f(A, D), b2=self._f(A), U2=self._f(A, D),
                    pooled=self._f(n_seg, D), d_pooled=self._f(n_seg, D))

    # AttentionOperator.forward over ragged segments
    def _att_fwd(self, pre, ws, x_ptr, rows_dyn, seg_off, n_cap, n_dyn, out, site, training, st, head=False, per_key=False):
        P, D, A = self.P, self.D, self.A
        rows = ws["rows"]
        m = current_stream()          # == st; tagged launches are HIP-event timed on it when bench.py asks
        tg = pre[:4]
        W1, b1 = P[pre + "additive_attention.encoder.0.weight"], P[pre + "additive_attention.encoder.0.bias"]
        Wo, bo = P[pre + "multi_head_attention.out_proj.weight"], P[pre + "multi_head_attention.out_proj.bias"]
        Wl, bl = P[pre + "linear.weight"], P[pre + "linear.bias"]
        w2 = P[pre + "additive_attention.encoder.2.weight"]
        if per_key == "dropcorr":                    # x_ptr = Eu [U, D]: q|k|v of the distinct keys (no bias), then the rows with the
            self.kk(m, "qkv_fwd_" + tg, "lego_linear_fwd", x_ptr, D, _ptr(P[pre + "multi_head_attention.in_proj_weight"]), D,     # sparse Dropout correction
                    None, _ptr(self.QKVu), 3 * D, self.Uc, self.cnt(6), 3 * D, D, 0, None, None, None, None)
            self.kk(m, "qkv_expand_" + tg, "lego_qkv_expand_dropcorr", _ptr(self.QKVu), 3 * D, x_ptr, D, _ptr(self.WinT), 3 * D,
                    _ptr(P[pre + "multi_head_attention.in_proj_bias"]), _ptr(self.inv), _ptr(self.tokinfo), self._dc_drop, rows, rows_dyn,
                    D, 3 * D, _ptr(ws["qkv"]), 3 * D)
        elif per_key:                                # x_ptr = Eu [U, D]: project the distinct keys, expand q|k|v to the sequence rows
            self.kk(m, "qkv_fwd_" + tg, "lego_linear_fwd", x_ptr, D, _ptr(P[pre + "multi_head_attention.in_proj_weight"]), D,
                    _ptr(P[pre + "multi_head_attention.in_proj_bias"]), _ptr(self.QKVu), 3 * D, self.Uc, self.cnt(6), 3 * D, D, 0,
                    None, None, None, None)
            self.kk(m, "qkv_expand_" + tg, "lego_expand_rows", _ptr(self.QKVu), 3 * D, _ptr(self.inv), rows, rows_dyn, 3 * D, None, None,
                    None, 0, None, None, 0, None, _ptr(ws["qkv"]), 3 * D)
        else:
            self.kk(m, "qkv_fwd_" + tg, "lego_linear_fwd", x_ptr, D, _ptr(P[pre + "multi_head_attention.in_proj_weight"]), D,
                    _ptr(P[pre + "multi_head_attention.in_proj_bias"]), _ptr(ws["qkv"]), 3 * D, rows, rows_dyn, 3 * D, D, 0,
                    None, None, None, None)
        core = ("lego_mhsa_core_fwd", _ptr(ws["qkv"]), 3 * D, _ptr(seg_off), n_cap, n_dyn, D, self.heads,
                _ptr(ws["o"]), D, _ptr(ws["lse"]), _ptr(ws["probs"]), ws["Lmax"], self.drop(self.p_att, site, training), rows)
        self.kk(m, "mhsa_core_fwd_" + tg, *core, *self._part(pre, ws))
        if self.fold and self._fold_ev is not None:          # the folded weights come from the side stream (_prepare_folds)
            m.wait_event(self._fold_ev)
            self._fold_ev = None
        if self.fold == 2:
            self.kk(m, "additive_fwd_" + tg, "lego_linear_fwd", _ptr(ws["o"]), D, _ptr(ws["W2"]), D, _ptr(ws["b2"]), _ptr(ws["t"]), A,
                    rows, rows_dyn, A, D, 2, None, None, None, None)
            call("lego_additive_pool_fwd", _ptr(ws["t"]), A, _ptr(ws["o"]), D, _ptr(w2), _ptr(seg_off), None, None, n_cap, n_dyn, D, A,
                 _ptr(ws["pooled"]), D, _ptr(ws["wrow"]), st)
            if head:
                # the user vector, the dot predictor, the loss and their backward down to d(pooled) in ONE launch (training steps)
                call("lego_nrms_user_head_train", _ptr(ws["pooled"]), D, _ptr(ws["Wc"]), _ptr(ws["bc"]), _ptr(self.items), D,
                     self.nb, self.C, D, 1.0 / self.nb, _ptr(out), D, _ptr(self.scores), _ptr(self.loss), _ptr(self.d_user), D,
                     _ptr(self.d_items), D, _ptr(ws["d_pooled"]), D, _ptr(seg_off), st)
                return
            # an EMPTY segment (a user without clicked items) gets the zero vector the un-folded operator pools for it, not the
            # folded bias bc: the product's live-mask epilogue reads one live bit per segment (item segments always have rows)
            live = None
            if pre == "user_op.":
                call("lego_segment_live", _ptr(seg_off), n_cap, n_dyn, _ptr(ws["seg_live"]), st)
                live = _ptr(ws["seg_live"])
            call("lego_linear_fwd", _ptr(ws["pooled"]), D, _ptr(ws["Wc"]), D, _ptr(ws["bc"]), _ptr(out), D, n_cap, n_dyn, D, D, 0,
                 live, None, None, None, st)
            return
        if self.fold:
            self.kk(m, "outlin_fwd_" + tg, "lego_linear_fwd", _ptr(ws["o"]), D, _ptr(ws["Wc"]), D, _ptr(ws["bc"]), _ptr(ws["lin"]), D,
                    rows, rows_dyn, D, D, 0, None, None, None, None)
        else:
            self.kk(m, "out_proj_fwd_" + tg, "lego_linear_fwd", _ptr(ws["o"]), D, _ptr(Wo), D, _ptr(bo), _ptr(ws["att"]), D,
                    rows, rows_dyn, D, D, 0, None, None, None, None)
            self.kk(m, "linear_fwd_" + tg, "lego_linear_fwd", _ptr(ws["att"]), D, _ptr(Wl), D, _ptr(bl), _ptr(ws["lin"]), D,
                    rows, rows_dyn, D, D, 0, None, None, None, None)
        self.kk(m, "additive_fwd_" + tg, "lego_linear_fwd", _ptr(ws["lin"]), D, _ptr(W1), D, _ptr(b1), _ptr(ws["t"]), A, rows, rows_dyn, A, D, 2,
                None, None, None, None)
        call("lego_additive_pool_fwd", _ptr(ws["t"]), A, _ptr(ws["lin"]), D, _ptr(w2), _ptr(seg_off), None, None, n_cap, n_dyn, D, A,
             _ptr(out), D, _ptr(ws["wrow"]), st)

    _fold_ev = None

    # user side: a few hundred (user, head) pairs, a fifth of them longer than 32 clicks -- ONE launch of the two-wave (<= 64 rows)
    # instantiation for all of them instead of a short-segment and a long-segment launch (LEGO_MHSA_ALL_LONG)
    user_one = True

    def _part(self, pre, ws):
        """(part, long_list, long_count) arguments of lego_mhsa_core_*"""
        if pre == "user_op." and self.user_one and ws["Lmax"] > 32:
            return 3, None, None
        return (0,) + self._long(pre, ws)

    def _long(self, pre, ws):
        """(list, count) of the operator's segments of more than 32 rows, or (None, None) when it cannot have any"""
        if ws["Lmax"] <= 32:
            return None, None
        user = pre == "user_op."
        return _ptr(self.long_users if user else self.long_items), _ptr(self.long_cnt, 1 if user else 0)

    def _prepare_folds(self):
        """Wc = Wl Wo, bc = Wl bo + bl (and, level 2, W2 = W1 Wc, b2 = W1 bc + b1) of both operators: D x D work that only needs the
        parameters, so it runs on the side stream beside the in-projection and the attention core; the main stream waits for it
        once, in front of the first product that uses a folded weight."""
        if not self.fold:
            return
        P, D, A = self.P, self.D, self.A
        m, sw = self._side()
        if sw is not m:
            self._sev[5].record(m)                       # the parameters are final on the main stream (Adam of the last step)
            sw.wait_event(self._sev[5])
        sp = stream_handle(sw)
        for pre, ws in (("item_op.", self.item_ws), ("user_op.", self.user_ws)):
            Wo, bo = P[pre + "multi_head_attention.out_proj.weight"], P[pre + "multi_head_attention.out_proj.bias"]
            Wl, bl = P[pre + "linear.weight"], P[pre + "linear.bias"]
            W1, b1 = P[pre + "additive_attention.encoder.0.weight"], P[pre + "additive_attention.encoder.0.bias"]
            two = self.fold == 2
            call("lego_attn_fold_prepare", _ptr(Wo), _ptr(bo), _ptr(Wl), _ptr(bl), _ptr(W1) if two else None, _ptr(b1) if two else None,
                 _ptr(ws["Wc"]), _ptr(ws["bc"]), _ptr(ws["W2"]) if two else None, _ptr(ws["b2"]) if two else None, D, A if two else 0, sp)
        if sw is not m:
            with torch.cuda.stream(sw):                  # the loss accumulator too (a fill on the main stream costs it the launch
                self.loss.zero_()                        # and a gap in front of the user head); ordered by the same event
            self._loss_zeroed = True
            self._fold_ev = self._sev[6]
            self._fold_ev.record(sw)

    def _pool_scratch(self, A, key):
        d = self.__dict__.setdefault("_pscr", {})
        if key not in d or d[key].numel() < 64 * A:
            d[key] = torch.zeros(64 * A, dtype=torch.float32, device=self.dev)
        return d[key]

    def _side(self):
        """side HIP stream for the weight-gradient products: they overlap the latency-bound attention-core kernels of the
        data-gradient chain (LEGO_SERIAL=1: everything on the current stream)"""
        if getattr(self, "_sw", None) is None:
            self._sw = shared_stream(self.dev, "side0")
            self._sev = [torch.cuda.Event() for _ in range(12)]
        m = current_stream()
        return m, (m if self._serial else self._sw)

    def _att_bwd(self, pre, ws, G, x_ptr, dx_ptr, rows_dyn, seg_off, n_cap, n_dyn, gout, site, training, st, ev, per_key=False):
        """data-gradient chain on the current stream `st`; weight gradients in two groups on the side stream, each behind
        ONE event (`ev[0]`, `ev[1]`) recorded where its inputs are final (the workspace is not overwritten before the
        next forward, which the caller orders after the side stream)"""
        P, D, A = self.P, self.D, self.A
        rows = ws["rows"]
        m, sw = self._side()
        sp = stream_handle(sw)
        if self.fold == 2:
            self._att_bwd_folded(pre, ws, G, rows_dyn, seg_off, n_cap, n_dyn, gout, st, ev, m, sw, sp)
        else:
            self._att_bwd_head(pre, ws, G, rows_dyn, seg_off, n_cap, n_dyn, gout, st, ev, m, sw, sp)
        core = ("lego_mhsa_core_bwd", _ptr(ws["qkv"]), 3 * D, _ptr(seg_off),
                n_cap, n_dyn, D, self.heads, _ptr(ws["d_o"]), D, _ptr(ws["lse"]), _ptr(ws["probs"]), ws["Lmax"],
                self.drop(self.p_att, site, training), rows, _ptr(ws["d_qkv"]), 3 * D,
                _ptr(G[pre + "multi_head_attention.in_proj_bias"]))      # bias gradient = column sums of d_qkv, fused
        self.kk(m, "mhsa_core_bwd_" + pre[:4], *core, *self._part(pre, ws))
        if per_key:
            # d(qkv) summed per distinct key, then both products of the in-projection's backward over the ~4.5 k keys instead of the
            # ~31 k sequence rows: dW_in = dQKVu^T Eu (side stream), dEu = dQKVu W_in (x_ptr = Eu, dx_ptr = dEu)
            W_in = P[pre + "multi_head_attention.in_proj_weight"]
            self.kk(m, "qkv_bwd_segsum", "lego_segment_sum_rows", _ptr(ws["d_qkv"]), 3 * D, 3 * D, _ptr(self.perm), _ptr(self.inv), rows,
                    _ptr(self.keys_sorted), rows_dyn, _ptr(self.dQKVu), 3 * D, self.Uc, self.cnt(6), 1, None, None)
            if sw is not m:
                ev[1].record(m)

            def side2k():
                if sw is not m:
                    sw.wait_event(ev[1])
                call("lego_linear_bwd_weight", _ptr(self.dQKVu), 3 * D, x_ptr, D, _ptr(G[pre + "multi_head_attention.in_proj_weight"]), D,
                     self.Uc, self.cnt(6), 3 * D, D, None, None, sp)
            if self.fold == 2:
                self._deferred.append(side2k)
            else:
                side2k()
            call("lego_linear_bwd_data", _ptr(self.dQKVu), 3 * D, _ptr(W_in), D, dx_ptr, D, self.Uc, self.cnt(6), 3 * D, D, 0,
                 None, 0, 1.0, None, None, None, None, None, st)
            return
        if sw is not m:
            ev[1].record(m)

        def side2():
            # ---- side, group 2: in-projection weight gradient (its bias gradient came out of the attention core)
            if sw is not m:
                sw.wait_event(ev[1])
            call("lego_linear_bwd_weight", _ptr(ws["d_qkv"]), 3 * D, x_ptr, D,
                 _ptr(G[pre + "multi_head_attention.in_proj_weight"]), D, rows, rows_dyn, 3 * D, D, None, None, sp)
        if self.fold == 2:
            self._deferred.append(side2)
        else:
            side2()
        # dx = d_qkv W_in
        call("lego_linear_bwd_data", _ptr(ws["d_qkv"]), 3 * D, _ptr(P[pre + "multi_head_attention.in_proj_weight"]), D,
             dx_ptr, D, rows, rows_dyn, 3 * D, D, 0, None, 0, 1.0, None, None, None, None, None, st)

    _deferred = ()

    def _side_fold_grads(self, pre, ws, G, sp, two):
        """side stream: lego_attn_fold_grads -- from T = dL/dWc, s = dL/dbc (and, fold level 2, Tp = dpre^T o, sp = colsum(dpre)) to the
        gradients of out_proj, Linear (and the additive hidden layer): two launches"""
        P, D, A = self.P, self.D, self.A
        g = lambda k: _ptr(G[pre + k])
        call("lego_attn_fold_grads", _ptr(P[pre + "multi_head_attention.out_proj.weight"]), _ptr(P[pre + "multi_head_attention.out_proj.bias"]),
             _ptr(P[pre + "linear.weight"]), _ptr(P[pre + "additive_attention.encoder.0.weight"]) if two else None,
             _ptr(ws["Wc"]), _ptr(ws["bc"]), _ptr(ws["Tp"]) if two else None, _ptr(ws["sp"]) if two else None, _ptr(ws["T"]), _ptr(ws["s"]),
             g("multi_head_attention.out_proj.weight"), g("multi_head_attention.out_proj.bias"), g("linear.weight"), g("linear.bias"),
             g("additive_attention.encoder.0.weight") if two else None, g("additive_attention.encoder.0.bias") if two else None,
             D, A if two else 0, sp)

    def _att_bwd_folded(self, pre, ws, G, rows_dyn, seg_off, n_cap, n_dyn, gout, st, ev, m, sw, sp):
        """fold level 2 (see __init__): with W2 = W1 Wc, pre = o W2^T + b2, out_i = pooled_i Wc^T + bc, pooled_i = sum_r w_r o_r:
            d pooled = d_out Wc;   pool backward on (t, o) gives dpre (in t) and d_o = w (x) d pooled;   d_o += dpre W2
            Tp = dpre^T o,  sp = colsum(dpre):   d W1 = Tp Wc^T + sp (x) bc,   d b1 = sp
            T  = d_out^T pooled + W1^T Tp,       s = colsum(d_out) + sp W1     (gradients of Wc and bc; then lego_attn_fold_grads)"""
        P, D, A = self.P, self.D, self.A
        rows = ws["rows"]
        W1 = P[pre + "additive_attention.encoder.0.weight"]
        gw2, gW1, gb1 = (G[pre + "additive_attention.encoder." + k] for k in ("2.weight", "0.weight", "0.bias"))
        if not (pre == "user_op." and self._have_d_pooled):
            if pre == "user_op.":                    # the zero vector of an empty segment is a constant: no gradient through it
                call("lego_segment_live", _ptr(seg_off), n_cap, n_dyn, _ptr(ws["seg_live"]), st)
                call("lego_mask_dropout_rows", _ptr(gout), D, n_cap, n_dyn, D, _ptr(ws["seg_live"]), None, None, st)
            call("lego_linear_bwd_data", _ptr(gout), D, _ptr(ws["Wc"]), D, _ptr(ws["d_pooled"]), D, n_cap, n_dyn, D, D, 0,
                 None, 0, 1.0, None, None, None, None, None, st)
        call("lego_additive_pool_bwd", _ptr(ws["t"]), A, _ptr(ws["o"]), D, _ptr(P[pre + "additive_attention.encoder.2.weight"]),
             _ptr(seg_off), None, n_cap, n_dyn, D, A, _ptr(ws["d_pooled"]), D, _ptr(ws["wrow"]), _ptr(ws["d_o"]), D,
             _ptr(gw2), _ptr(ws["sp"]), _ptr(self._pool_scratch(A, pre)), st)
        call("lego_linear_bwd_data", _ptr(ws["t"]), A, _ptr(ws["W2"]), D, _ptr(ws["d_o"]), D, rows, rows_dyn, A, D, 1,
             None, 0, 1.0, None, None, None, None, None, st)
        if sw is not m:
            ev[0].record(m)

        # ---- side stream: everything that ends in a parameter gradient.  Enqueued by backward() AFTER the main chain of both operators
        # (the host would otherwise be busy with these launches while the main stream waits for its attention-core launch).  Order on
        # the side stream: this operator's row product, its parameter-space launches, then (side2) the in-projection weight gradient --
        # the longest launch goes last so that the short ones run beside the main stream's attention core, not after everything
        def side_rows():
            if sw is not m:
                sw.wait_event(ev[0])
            with torch.cuda.stream(sw):                  # the four sums of this pass start from zero (one memset: _att_ws)
                ws["zsum"].zero_()
            call("lego_linear_bwd_weight", _ptr(ws["t"]), A, _ptr(ws["o"]), D, _ptr(ws["Tp"]), D, rows, rows_dyn, A, D, None, None, sp)

        def side_params():
            call("lego_additive_pool_bwd_fold", _ptr(self._pool_scratch(A, pre)), A, _ptr(gw2), _ptr(ws["sp"]), sp)
            call("lego_linear_bwd_weight", _ptr(gout), D, _ptr(ws["pooled"]), D, _ptr(ws["T"]), D, n_cap, n_dyn, D, D, None, None, sp)
            call("lego_colsum", _ptr(gout), D, n_cap, n_dyn, None, D, _ptr(ws["s"]), sp)
            self._side_fold_grads(pre, ws, G, sp, True)
        self._deferred.append(side_rows)
        self._deferred.append(side_params)

    def _att_bwd_head(self, pre, ws, G, rows_dyn, seg_off, n_cap, n_dyn, gout, st, ev, m, sw, sp):
        """the additive attention and the two affine layers behind the attention core, fold levels 0 and 1"""
        P, D, A = self.P, self.D, self.A
        rows = ws["rows"]
        call("lego_additive_pool_bwd", _ptr(ws["t"]), A, _ptr(ws["lin"]), D,
             _ptr(P[pre + "additive_attention.encoder.2.weight"]), _ptr(seg_off), None, n_cap, n_dyn, D, A,
             _ptr(gout), D, _ptr(ws["wrow"]), _ptr(ws["d_lin"]), D,
             _ptr(G[pre + "additive_attention.encoder.2.weight"]), _ptr(G[pre + "additive_attention.encoder.0.bias"]),
             _ptr(self._pool_scratch(A, pre)), st)
        Wo, bo = P[pre + "multi_head_attention.out_proj.weight"], P[pre + "multi_head_attention.out_proj.bias"]
        Wl = P[pre + "linear.weight"]
        # d_lin += dpre . W1 ; its column sums s are the gradient of linear.bias (ws["s"] is zero here: cleared on the side stream
        # after its last use, which the end-of-backward join orders before the next step)
        call("lego_linear_bwd_data", _ptr(ws["t"]), A, _ptr(P[pre + "additive_attention.encoder.0.weight"]), D,
             _ptr(ws["d_lin"]), D, rows, rows_dyn, A, D, 1, None, 0, 1.0, None, None,
             _ptr(ws["s"] if self.fold else G[pre + "linear.bias"]), None, None, st)
        if not self.fold:
            call("lego_linear_bwd_data", _ptr(ws["d_lin"]), D, _ptr(Wl), D, _ptr(ws["d_att"]), D,
                 rows, rows_dyn, D, D, 0, None, 0, 1.0, None, None, _ptr(G[pre + "multi_head_attention.out_proj.bias"]), None, None, st)
        if sw is not m:
            ev[0].record(m)
            sw.wait_event(ev[0])
        # ---- side, group 1: fold of the pool backward's copies; W1, Linear and out-projection weight gradients
        call("lego_additive_pool_bwd_fold", _ptr(self._pool_scratch(A, pre)), A,
             _ptr(G[pre + "additive_attention.encoder.2.weight"]), _ptr(G[pre + "additive_attention.encoder.0.bias"]), sp)
        call("lego_linear_bwd_weight", _ptr(ws["t"]), A, _ptr(ws["lin"]), D,
             _ptr(G[pre + "additive_attention.encoder.0.weight"]), D, rows, rows_dyn, A, D, None, None, sp)
        if self.fold:
            # T = d_lin^T o is the only product over the sequence rows; with att = o Wo^T + bo and d_att = d_lin Wl:
            #   d Wl = d_lin^T att = T Wo^T + s (x) bo      d bl = s
            #   d Wo = d_att^T o   = Wl^T T                 d bo = colsum(d_att) = s Wl
            with torch.cuda.stream(sw):
                ws["T"].zero_()
            call("lego_linear_bwd_weight", _ptr(ws["d_lin"]), D, _ptr(ws["o"]), D, _ptr(ws["T"]), D, rows, rows_dyn, D, D, None, None, sp)
            self._side_fold_grads(pre, ws, G, sp, False)
            with torch.cuda.stream(sw):
                ws["s"].zero_()                          # the next pass's column sums (main stream epilogue) start from zero
            # ---- main: data gradient through both layers at once
            call("lego_linear_bwd_data", _ptr(ws["d_lin"]), D, _ptr(ws["Wc"]), D, _ptr(ws["d_o"]), D,
                 rows, rows_dyn, D, D, 0, None, 0, 1.0, None, None, None, None, None, st)
        else:
            call("lego_linear_bwd_weight", _ptr(ws["d_lin"]), D, _ptr(ws["att"]), D, _ptr(G[pre + "linear.weight"]), D,
                 rows, rows_dyn, D, D, None, None, sp)
            call("lego_linear_bwd_weight", _ptr(ws["d_att"]), D, _ptr(ws["o"]), D,
                 _ptr(G[pre + "multi_head_attention.out_proj.weight"]), D, rows, rows_dyn, D, D, None, None, sp)
            # ---- main: out-projection data gradient, attention core
            call("lego_linear_bwd_data", _ptr(ws["d_att"]), D, _ptr(Wo), D,
                 _ptr(ws["d_o"]), D, rows, rows_dyn, D, D, 0, None, 0, 1.0, None, None, None, None, None, st)

    def _key_table_grads(self, G, g_spec, g_cat, n_cat, st):
        """per-key in-projection (qkv_dedup): dEu [U, D] is the gradient of the ONE live look-up of every distinct key -- added to the row
        of the table that key belongs to (no two keys share a destination row)"""
        D, V = self.D, self.V
        # (the `_range` entry point = the plain one-atomic-per-element kernel: no two keys share a destination row, so the LDS
        # pre-reduction of the small-table path -- made for thousands of rows hitting 3 / 18 rows -- would only add its latency: 2 x 18 us)
        call("lego_scatter_add_rows_range", _ptr(g_spec), D, D, _ptr(self.idx_spec_u), self.Uc, self.cnt(6), _ptr(self.dHu), D, 0, 3, st)
        call("lego_scatter_add_rows_range", _ptr(g_cat), D, D, _ptr(self.idx_cat_u), self.Uc, self.cnt(6), _ptr(self.dHu), D, 0, n_cat, st)
        if self.touched_rows is not None:
            call("lego_mark_rows", _ptr(self.idx_tok_u), self.Uc, self.cnt(6), V, _ptr(self.touched_rows), st)
        if self.grad_hooks is None:
            call("lego_scatter_add_rows", _ptr(G["embedding_vocab_table.glove.weight"]), D, D, V, _ptr(self.idx_tok_u), self.Uc, self.cnt(6),
                 _ptr(self.dHu), D, st)

    def _key_table_buckets(self, G, st):
        """data parallel: the table gradient one destination-row bucket at a time, each handed to the exchange behind its scatter"""
        if self.grad_hooks is None:
            return
        D, V = self.D, self.V
        dense_ready, bucket_ready, per = self.grad_hooks
        dense_ready()
        for lo in range(0, V, per):
            hi = min(V, lo + per)
            call("lego_scatter_add_rows_range", _ptr(G["embedding_vocab_table.glove.weight"]), D, D, _ptr(self.idx_tok_u), self.Uc,
                 self.cnt(6), _ptr(self.dHu), D, lo, hi, st)
            bucket_ready(lo, hi)

    def _plan_tables(self):
        return self.seq_tok, self.seq_len, self.L

    def forward(self, cand, hist, hist_len, training=False, with_loss=True, planned=False, fork_ev=None, neck_ev=None):
        P, B, C, S, D = self.P, self.nb, self.C, self.S, self.D
        st = _stream()
        _check(cand, torch.int32, "cand"); _check(hist, torch.int32, "hist"); _check(hist_len, torch.int32, "hist_len")
        self._training = training
        if not planned:
            self._plan(cand, hist, hist_len)
        self._forward_items(training, planned)
        if neck_ev is not None:
            neck_ev.record(current_stream())
        if not self._loss_zeroed:
            self.loss.zero_()
        self._loss_zeroed = False
        self._head_done = self.fold == 2 and training and with_loss
        self._forward_users(training, head=self._head_done)
        if not self._head_done:
            call("lego_dot_ce_fwd", _ptr(self.user), D, _ptr(self.items), D, B, C, D, _ptr(self.scores),
                 _ptr(self.loss) if with_loss else None, st)
        self.step += 1 if training else 0
        return self.scores, self.loss

    def _plan(self, cand, hist, hist_len):
        call("lego_plan_batch", _ptr(cand), _ptr(hist), _ptr(hist_len), self.nb, self.C, self.S,
             _ptr(self.seq_tok), _ptr(self.seq_len), self.L,
             _ptr(self.counters), _ptr(self.inst_item), _ptr(self.seg_off), _ptr(self.hist_off),
             _ptr(self.rowinfo), _ptr(self.row_tok), _stream())
        self._mask_step = -1                         # no keep bits were drawn for an un-planned batch
        self._dhu_zeroed = False
        self._cur_slot = None
        self._long_lists(self.__dict__, _stream())

    _folds_fresh = False
    def _forward_items(self, training, planned=False):
        P, D = self.P, self.D
        st = _stream()
        self._prepare_folds()
        self._folds_fresh = True
        if self._dc_e_pending:                       # a per-key forward pass with no backward behind it left its row expansion (E, on the side
            m_, sw_ = self._side()                   # stream, reading Eu) un-joined: EVERY branch below rewrites E or Eu
            if sw_ is not m_:
                m_.wait_event(self._sev[9])
            self._dc_e_pending = False
        if not (planned and getattr(self, "_slots", None) is not None):      # else: done with the plan (plan_on)
            self._decode_gather(self.__dict__, current_stream())
        self._dc_active = False
        dp = self.drop(self.p_proj, SITE_PROJ, training) if self.dropcorr else None
        if self.dropcorr and (dp is None or self._mask_step == self.step):        # (keep bits drawn ahead of time, or nothing to drop)
            E0 = self.E0
            spec, catw = P["embedding_vocab_table.__cat_inputer_special_ids.weight"], P["embedding_vocab_table.category.weight"]
            m, sw = self._side()
            if dp is not None:                       # W_in^T for the two correction kernels (768 KB, once per step)
                with torch.cuda.stream(m):
                    self.WinT.copy_(P["item_op.multi_head_attention.in_proj_weight"].t())
            # Hu = projection of the distinct TOKEN keys ([SEP] / category keys: zero rows of Xu, masked next) ...
            call("lego_linear_fwd", _ptr(self.Xu), E0, _ptr(P["embedding_vocab_table.glove.linear.weight"]), E0,
                 _ptr(P["embedding_vocab_table.glove.linear.bias"]), _ptr(self.Hu), D, self.Uc, self.cnt(6), D, E0, 0,
                 None, None, None, None, st)
            # ... Eu[k] = the ONE live look-up of key k: the token's projection, or the [SEP] / category row (identity expansion)
            call("lego_expand_rows", _ptr(self.Hu), D, _ptr(self.iota_u), self.Uc, self.cnt(6), D, None, _ptr(self.tokinfo_u),
                 _ptr(spec), D, _ptr(self.idx_spec_u), _ptr(catw), D, _ptr(self.idx_cat_u), _ptr(self.Eu), D, st)
            if not getattr(self, "_eval_only", False):
                # the row embeddings E (Dropout applied per row) are needed by the in-projection's WEIGHT gradient only: side stream
                # (every forward pass a backward pass may follow -- the parity tests differentiate the eval-mode forward too)
                if sw is not m:
                    self._sev[8].record(m)
                    sw.wait_event(self._sev[8])
                call("lego_expand_rows", _ptr(self.Eu), D, _ptr(self.inv), self.Rc, self.cnt(0), D, dp, _ptr(self.tokinfo),
                     _ptr(spec), D, _ptr(self.idx_spec), _ptr(catw), D, _ptr(self.idx_cat), _ptr(self.E), D, stream_handle(sw))
                if sw is not m:
                    self._sev[9].record(sw)
                    self._dc_e_pending = True
            self._dc_active = True
            self._dc_drop = dp
            self._att_fwd("item_op.", self.item_ws, _ptr(self.Eu), self.cnt(0), self.seg_off, self.NIc, self.cnt(1),
                          self.items, SITE_ITEM_ATT, training, st, per_key="dropcorr")
            return
        # (a training step whose keep bits were NOT drawn with the plan falls through to the row-by-row in-projection below: the key space
        # serves it too -- the [SEP] / category keys have zero rows in Xu and the live bit of tokinfo masks their rows)
        if self.glove and self.dedup:
            E0 = self.E0
            call("lego_linear_fwd", _ptr(self.Xu), E0, _ptr(P["embedding_vocab_table.glove.linear.weight"]), E0,
                 _ptr(P["embedding_vocab_table.glove.linear.bias"]), _ptr(self.Hu), D, self.Uc, self.cnt(6), D, E0, 0,
                 None, None, None, None, st)
            # ... and the two other look-ups ConcatInputer sums in (the [SEP] / category rows) added by the same kernel
            call("lego_expand_rows", _ptr(self.Hu), D, _ptr(self.inv), self.Rc, self.cnt(0), D, self.drop(self.p_proj, SITE_PROJ, training),
                 _ptr(self.tokinfo), _ptr(P["embedding_vocab_table.__cat_inputer_special_ids.weight"]), D, _ptr(self.idx_spec),
                 _ptr(P["embedding_vocab_table.category.weight"]), D, _ptr(self.idx_cat), _ptr(self.E), D, st)
            self._att_fwd("item_op.", self.item_ws, _ptr(self.E), self.cnt(0), self.seg_off, self.NIc, self.cnt(1),
                          self.items, SITE_ITEM_ATT, training, st)
            return
        elif self.glove:
            E0 = self.E0
            call("lego_linear_fwd", _ptr(self.X), E0, _ptr(P["embedding_vocab_table.glove.linear.weight"]), E0,
                 _ptr(P["embedding_vocab_table.glove.linear.bias"]), _ptr(self.E), D, self.Rc, self.cnt(0), D, E0, 0,
                 _ptr(self.tokinfo), self.drop(self.p_proj, SITE_PROJ, training), None, None, st)
        else:
            # trainable table (embed/null): the three look-ups ConcatInputer sums (concat_inputer.py:96-114) in ONE pass over the sequence
            # rows -- token row where the position holds a token (tokinfo's live bit; idx_tok is -1 elsewhere and not read), plus the
            # special-id and category rows where those indices are >= 0
            if self.qkv_dedup:                       # Eu[u] = the one live look-up of distinct key u (token / [SEP] / category row)
                self.kk(current_stream(), "embed_gather_item", "lego_expand_rows", _ptr(P["embedding_vocab_table.glove.weight"]), D,
                        _ptr(self.idx_tok_u), self.Uc, self.cnt(6), D, None, _ptr(self.tokinfo_u),
                        _ptr(P["embedding_vocab_table.__cat_inputer_special_ids.weight"]), D, _ptr(self.idx_spec_u),
                        _ptr(P["embedding_vocab_table.category.weight"]), D, _ptr(self.idx_cat_u), _ptr(self.Eu), D)
                self._att_fwd("item_op.", self.item_ws, _ptr(self.Eu), self.cnt(0), self.seg_off, self.NIc, self.cnt(1),
                              self.items, SITE_ITEM_ATT, training, st, per_key=True)
                return
            # (tagged: bench.py prices this launch -- the trainable table's row gather, on the step's critical path -- against the HBM roof)
            self.kk(current_stream(), "embed_gather_item", "lego_expand_rows", _ptr(P["embedding_vocab_table.glove.weight"]), D,
                    _ptr(self.idx_tok), self.Rc, self.cnt(0), D, None,
                    _ptr(self.tokinfo), _ptr(P["embedding_vocab_table.__cat_inputer_special_ids.weight"]), D, _ptr(self.idx_spec),
                    _ptr(P["embedding_vocab_table.category.weight"]), D, _ptr(self.idx_cat), _ptr(self.E), D)
            self._att_fwd("item_op.", self.item_ws, _ptr(self.E), self.cnt(0), self.seg_off, self.NIc, self.cnt(1),
                          self.items, SITE_ITEM_ATT, training, st)
            return
        call("lego_gather_rows", _ptr(P["embedding_vocab_table.__cat_inputer_special_ids.weight"]), D, D,
             _ptr(self.idx_spec), self.Rc, self.cnt(0), _ptr(self.E), D, 1, st)
        call("lego_gather_rows", _ptr(P["embedding_vocab_table.category.weight"]), D, D, _ptr(self.idx_cat),
             self.Rc, self.cnt(0), _ptr(self.E), D, 1, st)
        self._att_fwd("item_op.", self.item_ws, _ptr(self.E), self.cnt(0), self.seg_off, self.NIc, self.cnt(1),
                      self.items, SITE_ITEM_ATT, training, st)

    _head_done = False
    _have_d_pooled = False
    _loss_zeroed = False

    def _forward_users(self, training, head=False):
        if not self._folds_fresh:                # evaluation caches call this without a preceding _forward_items
            self._prepare_folds()
        self._folds_fresh = False
        self._att_fwd("user_op.", self.user_ws, _ptr(self.items, self.BC * self.D), self.cnt(3), self.hist_off, self.nb, None,
                      self.user, SITE_USER_ATT, training, _stream(), head=head)

    def backward(self, G, gloss: float = 1.0, gloss_dev: Optional[torch.Tensor] = None):
        P, B, C, S, D = self.P, self.nb, self.C, self.S, self.D
        st = _stream()
        training = self._training
        step_save = self.step
        if training:
            self.step -= 1
        # forward's fused head already holds d_user, d_items (candidates), d_pooled -- for a unit loss gradient known on the host
        head = self._head_done and float(gloss) == 1.0 and gloss_dev is None
        self._head_done = False
        if not head:
            call("lego_dot_ce_bwd", _ptr(self.user), D, _ptr(self.items), D, _ptr(self.scores), B, C, D,
                 float(gloss) / B, _ptr(gloss_dev), _ptr(self.d_user), D, _ptr(self.d_items), D, st)
        self._have_d_pooled = head
        m, sw = self._side()
        sev = self._sev if sw is not m else [None] * 6
        self._dc_e_pending = False                   # (this pass ends with the side stream joined: the forward's row expansion with it)
        self._deferred = []
        self._att_bwd("user_op.", self.user_ws, G, _ptr(self.items, self.BC * D), _ptr(self.d_items, self.BC * D),
                      self.cnt(3), self.hist_off, B, None, self.d_user, SITE_USER_ATT, training, st, sev[0:2])
        for side in self._deferred:                  # the user operator's side-stream launches: its main chain is enqueued
            side()
        self._deferred = []
        g_spec, g_cat = G["embedding_vocab_table.__cat_inputer_special_ids.weight"], G["embedding_vocab_table.category.weight"]
        n_cat = g_cat.shape[0]
        if self.qkv_dedup:
            self._att_bwd("item_op.", self.item_ws, G, _ptr(self.Eu), _ptr(self.dHu), self.cnt(0), self.seg_off, self.NIc,
                          self.cnt(1), self.d_items, SITE_ITEM_ATT, training, st, sev[2:4], per_key=True)
            self._key_table_grads(G, g_spec, g_cat, n_cat, st)
            for side in self._deferred:
                side()
            self._deferred = ()
            if sw is not m:
                sev[4].record(sw)
                m.wait_event(sev[4])
            self._key_table_buckets(G, st)
            self.step = step_save
            return
        self._att_bwd("item_op.", self.item_ws, G, _ptr(self.E), _ptr(self.dE), self.cnt(0), self.seg_off, self.NIc,
                      self.cnt(1), self.d_items, SITE_ITEM_ATT, training, st, sev[2:4])
        # embedding tables: the three summed look-ups of ConcatInputer.get_embeddings.  [SEP] (id 2 of the special table) and the
        # category row come from fixed places of every item's sequence (lego_nrms_special_grads)
        if n_cat > 32:                               # category tables beyond the segment kernel's 32-row LDS image: two generic
            # scatters over the sequence rows (the segment kernel replaced them and hard-failed above 32 rows)
            call("lego_scatter_add_rows", _ptr(g_cat), D, D, n_cat, _ptr(self.idx_cat), self.Rc, self.cnt(0), _ptr(self.dE), D, st)
            call("lego_scatter_add_rows", _ptr(g_spec), D, D, 3, _ptr(self.idx_spec), self.Rc, self.cnt(0), _ptr(self.dE), D, st)
        else:
            call("lego_nrms_special_grads", _ptr(self.seg_off), self.NIc, self.cnt(1), _ptr(self.idx_cat), _ptr(self.dE), D, D,
                 _ptr(g_spec, 2 * D), _ptr(g_cat), D, n_cat, st)
        if self.glove:
            E0 = self.E0
            gb = _ptr(G["embedding_vocab_table.glove.linear.bias"])
            dp = self.drop(self.p_proj, SITE_PROJ, training)
            # with the keep bits at hand (or nothing to drop) the per-token sums mask and rescale dE as they read it
            in_sums = self.dedup and (dp is None or self._mask_step == self.step)
            if not in_sums:                          # the three-pass form (keep bits not drawn with the plan)
                call("lego_mask_dropout_rows", _ptr(self.dE), D, self.Rc, self.cnt(0), D, _ptr(self.tokinfo),
                     dp, None if self.dedup else gb, st)
            if self.dedup:                           # per-token sums of the masked dE, then the product over the distinct tokens
                call("lego_segment_sum_rows", _ptr(self.dE), D, D, _ptr(self.perm), _ptr(self.inv), self.Rc, _ptr(self.keys_sorted),
                     self.cnt(0), _ptr(self.dHu), D, self.Uc, self.cnt(6), 0 if self._dhu_zeroed else 1, dp if in_sums else None,
                     _ptr(self.tokinfo) if in_sums else None, st)
                self._dhu_consumed()
                call("lego_colsum", _ptr(self.dHu), D, self.Uc, self.cnt(6), None, D, gb, st)      # bias gradient = column sums of the per-token sums
                call("lego_linear_bwd_weight", _ptr(self.dHu), D, _ptr(self.Xu), E0,
                     _ptr(G["embedding_vocab_table.glove.linear.weight"]), E0, self.Uc, self.cnt(6), D, E0, None, None, st)
            else:
                call("lego_linear_bwd_weight", _ptr(self.dE), D, _ptr(self.X), E0,
                     _ptr(G["embedding_vocab_table.glove.linear.weight"]), E0, self.Rc, self.cnt(0), D, E0, None, None, st)
        else:
            V = G["embedding_vocab_table.glove.weight"].shape[0]
            if self.touched_rows is not None:       # TrainStep: rows that have ever had a gradient (row-skipping dense Adam)
                call("lego_mark_rows", _ptr(self.idx_tok), self.Rc, self.cnt(0), V, _ptr(self.touched_rows), st)
            if self.dedup:                           # per-token sums of dE (the [SEP] / category positions masked out) ...
                call("lego_segment_sum_rows", _ptr(self.dE), D, D, _ptr(self.perm), _ptr(self.inv), self.Rc, _ptr(self.keys_sorted),
                     self.cnt(0), _ptr(self.dHu), D, self.Uc, self.cnt(6), 0 if self._dhu_zeroed else 1, None, _ptr(self.tokinfo), st)
                self._dhu_consumed()
            if self.grad_hooks is None:
                if self.dedup:                       # ... added to the DISTINCT table rows: no two rows of this launch share a destination
                    call("lego_scatter_add_rows", _ptr(G["embedding_vocab_table.glove.weight"]), D, D, V, _ptr(self.uniq),
                         self.Uc, self.cnt(6), _ptr(self.dHu), D, st)
                else:
                    call("lego_scatter_add_rows", _ptr(G["embedding_vocab_table.glove.weight"]), D, D, V, _ptr(self.idx_tok),
                         self.Rc, self.cnt(0), _ptr(self.dE), D, st)
        for side in self._deferred:                  # the item operator's side-stream launches, behind its whole main chain
            side()
        self._deferred = ()
        if sw is not m:
            sev[4].record(sw)
            m.wait_event(sev[4])                     # every gradient is ordered on the caller's stream again
        if self.grad_hooks is not None and not self.glove:
            # data parallel with the trainable table: every dense gradient is final here, so its all-reduce starts now; the
            # table gradient follows bucket by bucket (destination-row ranges), each bucket handed to the exchange as soon as
            # its scatter is enqueued -- the ring works on bucket k while bucket k+1 is being scattered
            dense_ready, bucket_ready, per = self.grad_hooks
            dense_ready()
            for lo in range(0, V, per):
                hi = min(V, lo + per)
                if self.dedup:
                    call("lego_scatter_add_rows_range", _ptr(G["embedding_vocab_table.glove.weight"]), D, D, _ptr(self.uniq),
                         self.Uc, self.cnt(6), _ptr(self.dHu), D, lo, hi, st)
                else:
                    call("lego_scatter_add_rows_range", _ptr(G["embedding_vocab_table.glove.weight"]), D, D, _ptr(self.idx_tok),
                         self.Rc, self.cnt(0), _ptr(self.dE), D, lo, hi, st)
                bucket_ready(lo, hi)
        self.step = step_save
