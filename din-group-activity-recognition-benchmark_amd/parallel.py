"""Data parallelism over clips: one process per GPU, gradients all-reduced by RCCL over xGMI (backend "nccl" on ROCm;
"gloo" for the CPU tests).  Replaces the reference's nn.DataParallel (train_net_dynamic.py:95-96) -- no per-step
parameter broadcast, no scatter/gather through GPU 0.

Clips are independent units, so the ONLY exchange step is the gradient all-reduce.  Gradients are packed
into a few large flat buckets in reverse registration order (head/DIN first, conv1 last -- the order backward produces
them) and reduced with asynchronous collectives on the communication stream; xGMI is point-to-point (7 links x ~153
GB/s), so few, large messages (default 64 MiB buckets; the whole VGG16 model is 117 MB = 2 buckets) beat many small ones.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # DIN_DIST_BACKEND=gloo + DIN_SINGLE_DEVICE=1: debugging aid to exercise the N>1 control flow on a 1-GPU box
            backend = os.environ.get("DIN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("DIN_SINGLE_DEVICE") == "1":
            local = 0
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif os.environ.get("DIN_SINGLE_DEVICE") == "1":
        local = 0
    return rank, local, world


def shard_range(total: int, rank: int, world: int) -> range:
    """Clips [r*B/G, (r+1)*B/G) of a global batch; remainder clips go to the lowest ranks."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


class GradBuckets:
    """Flat fp32 gradient buckets over a parameter list; `allreduce()` averages them across ranks in place."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        order = list(reversed(self.params))                    # backward produces the last layers first
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        for p in order:
            nb = p.numel() * 4
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)

    def allreduce(self, world: Optional[int] = None, async_op: bool = True) -> None:
        """Average the gradients across ranks.  Per bucket: ONE concatenation into the flat buffer (not one copy per tensor: a
        backbone has ~220 parameter tensors, most of them BatchNorm vectors), one asynchronous all-reduce, one scale; afterwards
        every `p.grad` IS a view of the flat buffer (no copy back) -- the optimizer reads the views."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        world = world or dist.get_world_size()
        handles = []
        for bi, bucket in enumerate(self.buckets):
            total = sum(p.numel() for p in bucket)
            dev = bucket[0].device
            flat = self._flat[bi]
            if flat is None or flat.numel() != total or flat.device != dev:
                flat = self._flat[bi] = torch.empty(total, dtype=torch.float32, device=dev)
            pieces, off, already = [], 0, True
            for p in bucket:
                n = p.numel()
                g = p.grad
                if g is None:
                    g = torch.zeros(n, dtype=torch.float32, device=dev)
                # a gradient that already lives at its slot (previous step's view, accumulated into in place) needs no packing
                if not (g.dtype == torch.float32 and g.is_contiguous() and g.data_ptr() == flat.data_ptr() + 4 * off):
                    already = False
                pieces.append(g.reshape(-1).float())
                off += n
            if not already:
                lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * total
                if any(lo <= g.data_ptr() < hi for g in pieces):
                    flat.copy_(torch.cat(pieces))              # some pieces are views of `flat` itself (kept from the last step)
                else:
                    torch.cat(pieces, out=flat)
            handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op), bi))
        for h, bi in handles:
            if h is not None and async_op:
                h.wait()
            flat, off = self._flat[bi], 0
            flat.div_(world)
            for p in self.buckets[bi]:
                n = p.numel()
                p.grad = flat[off:off + n].view(p.shape)
                off += n


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """One-time sync at start-up (replicas then stay identical by construction: same grads, same optimiser)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
