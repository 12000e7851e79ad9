"""Overlapped input feed: uint8 clips travel host -> HBM on a COPY stream while the previous step computes.

The reference moves every batch with a blocking `b.to(device)` at the top of the step (train_net_dynamic.py:174; volleyball.py:223-275
builds fp32 images on the host, 33 MB per clip).  Here clips stay uint8 (8.3 MB per clip) and the transfer of batch k+1 is enqueued on
its own HIP stream as soon as batch k has been handed out, into the other half of a double buffer; the compute stream only waits on
the event of the batch it is about to read.  Host tensors that are not pinned are staged through persistent pinned buffers (one
hipHostMalloc per slot and tensor position, not per step)."""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence

import torch


class DeviceFeed:
    """Wrap an iterable of host batches (tensor, or tuple / list of tensors); iterate device batches.

    for batch in DeviceFeed(loader, device):   # batch k+1 is already in flight while the body runs on batch k
        ...
    """

    def __init__(self, batches: Iterable, device: torch.device, slots: int = 2):
        assert slots >= 2, "a double buffer needs two slots"
        self.batches, self.device, self.slots = batches, torch.device(device), slots
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned: List[dict] = [dict() for _ in range(slots)]        # slot -> {position: pinned staging tensor}
        self._copied: List[Optional[torch.cuda.Event]] = [None] * slots   # slot -> completion of the last transfer out of its pinned buffers

    def _stage(self, slot: int, pos: int, t: torch.Tensor) -> torch.Tensor:
        if t.is_pinned():
            return t
        buf = self._pinned[slot].get(pos)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = self._pinned[slot][pos] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        buf.copy_(t)
        return buf

    def _enqueue(self, slot: int, host_batch):
        single = isinstance(host_batch, torch.Tensor)
        items: Sequence[torch.Tensor] = (host_batch,) if single else tuple(host_batch)
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()                              # host: the slot's pinned buffers were last read two batches ago
        with torch.cuda.stream(self.copy_stream):
            dev = [self._stage(slot, i, t).to(self.device, non_blocking=True) for i, t in enumerate(items)]
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        self._copied[slot] = ready
        return (dev[0] if single else dev), ready

    def __iter__(self) -> Iterator:
        it = iter(self.batches)
        try:
            pending = self._enqueue(0, next(it))
        except StopIteration:
            return
        k = 0
        while pending is not None:
            dev, ready = pending
            try:
                nxt = next(it)
            except StopIteration:
                nxt = None
            pending = self._enqueue((k + 1) % self.slots, nxt) if nxt is not None else None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            for t in ((dev,) if isinstance(dev, torch.Tensor) else dev):
                t.record_stream(cur)                                      # allocator: the compute stream uses memory the copy stream allocated
            yield dev
            k += 1

    def __len__(self):
        return len(self.batches)
