"""Workspace arena of the plug-in route's large operators (config 5: the BERT news encoder, bert_native.py).

A training step of the BERT-base encoder saves ~46 KB per live row and block -- ~15 GB at B = 64 -- in ~20 tensors per block whose
sizes follow the batch's live-row count, i.e. differ from step to step.  Through torch's caching allocator that is ~250 differently
sized requests per step: every batch larger than the ones before sends it for fresh segments (hipMalloc inside a steady-state step:
120 -> 160-360 ms).  Here the operator takes all of it -- saved activations, LayerNorm statistics, backward
temporaries -- as views of ONE buffer with a bump pointer:

  * frames: a forward pass opens a frame, its backward pass (or the death of its autograd node) closes it; frames close in LIFO order
    (out of order: the space is reclaimed when everything above it has closed too);
  * the buffer only grows: a request that does not fit gets a new chunk (sized with `headroom` over everything in use), and when the
    arena is empty again the chunks are replaced by one of the peak size times `headroom`.  After the first steps of a run no step
    allocates: `Arena.allocations` counts the chunk allocations, tests/test_bert_operator.py holds it constant over steady-state steps.

Memory is sized for 288 GB of HBM: the arena never returns memory to the driver while the process lives (`reset()` does)."""
from __future__ import annotations

from typing import Dict, List

import torch

ALIGN = 256


class _Mark:
    """the arena's own record of an open frame (the `Frame` handle the caller holds points at it; the arena never references the handle, so a
    handle dropped with its autograd node is finalised)"""
    __slots__ = ("chunk", "off", "used", "alive")

    def __init__(self, chunk, off, used):
        self.chunk, self.off, self.used, self.alive = chunk, off, used, True


class Frame:
    __slots__ = ("arena", "mark")

    def __init__(self, arena, mark):
        self.arena, self.mark = arena, mark

    @property
    def alive(self):
        return self.mark.alive

    def release(self):
        if self.mark.alive:
            self.mark.alive = False
            self.arena._reclaim()

    def __del__(self):                               # the autograd node that owned the frame died without a backward pass (eval, a dropped graph)
        try:
            self.release()
        except Exception:                            # interpreter shutdown
            pass


class Arena:
    def __init__(self, device, headroom: float = 1.3):
        self.device, self.headroom = torch.device(device), float(headroom)
        self.chunks: List[torch.Tensor] = []
        self.cur, self.off = 0, 0                    # bump pointer: chunk index, byte offset inside it
        self.used = 0                                # bytes handed out in the open frames (aligned), over all chunks
        self.peak = 0
        self.frames: List[_Mark] = []
        self.allocations = 0                         # chunk allocations so far (growth events)

    # ------------------------------------------------------------------ frames
    def push(self) -> Frame:
        m = _Mark(self.cur, self.off, self.used)
        self.frames.append(m)
        return Frame(self, m)

    def _reclaim(self):
        low = None
        while self.frames and not self.frames[-1].alive:
            low = self.frames.pop()
        if low is not None:
            self.cur, self.off, self.used = low.chunk, low.off, low.used
        if not self.frames and len(self.chunks) > 1:
            # empty again after a growth event: one chunk of the peak size (with headroom) replaces the pieces
            want = int(self.peak * self.headroom)
            self.chunks = []
            torch.cuda.empty_cache()
            self._new_chunk(want)
            self.cur, self.off, self.used = 0, 0, 0

    def reset(self):
        """drop every chunk (tests; a trainer that is done with the operator)"""
        for f in self.frames:
            f.alive = False
        self.frames, self.chunks, self.cur, self.off, self.used, self.peak = [], [], 0, 0, 0, 0

    # ------------------------------------------------------------------ memory
    def _new_chunk(self, nbytes: int):
        nbytes = (int(nbytes) + ALIGN - 1) // ALIGN * ALIGN
        self.chunks.append(torch.empty(nbytes, dtype=torch.uint8, device=self.device))
        self.allocations += 1

    def take(self, *shape, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = (n * torch.empty(0, dtype=dtype).element_size() + ALIGN - 1) // ALIGN * ALIGN
        if nbytes == 0:
            return torch.empty(*shape, dtype=dtype, device=self.device)
        while True:
            if self.cur < len(self.chunks) and self.off + nbytes <= self.chunks[self.cur].numel():
                break
            if self.cur + 1 < len(self.chunks):       # (a later chunk from an earlier growth event)
                self.cur, self.off = self.cur + 1, 0
                continue
            self._new_chunk(max(nbytes, int((self.used + nbytes) * self.headroom)) if self.chunks else int(nbytes * self.headroom))
            self.cur, self.off = len(self.chunks) - 1, 0
        t = self.chunks[self.cur][self.off:self.off + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(*shape)
        self.off += nbytes
        self.used += nbytes
        self.peak = max(self.peak, self.used)
        if zero:
            t.zero_()
        return t

    def reserve(self, nbytes: int):
        """size the arena ahead of the first step (a caller that knows its capacity: no growth event at all)"""
        if not self.frames and sum(c.numel() for c in self.chunks) < nbytes:
            self.chunks = []
            self._new_chunk(nbytes)
            self.cur, self.off, self.used = 0, 0, 0
            self.peak = max(self.peak, int(nbytes / self.headroom))


_ARENAS: Dict[str, Arena] = {}


def arena_of(device) -> Arena:
    """the process-wide arena of a device"""
    key = str(torch.device(device))
    if key not in _ARENAS:
        _ARENAS[key] = Arena(device)
    return _ARENAS[key]
