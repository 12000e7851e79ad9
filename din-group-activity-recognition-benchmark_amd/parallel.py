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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL (effective when set before the HIP runtime starts)
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
    """Flat fp32 gradient buckets over a parameter list; `allreduce()` averages them across ranks in place.

    Overlap with the backward pass: the conv stack is one autograd node, so autograd hooks would only fire at its end.  Instead
    `nhwc.GRAD_HOOK` reports every conv weight gradient the moment its kernel is enqueued (last layer first).  The first step
    learns that order; from then on the hooked weights are laid out in buckets in exactly that order ("early" buckets), each
    packed with one concatenation and sent off with an asynchronous all-reduce as soon as its last gradient was reported --
    RCCL then runs over xGMI while the remaining layers are still computing.  Everything else (BatchNorm vectors, which one
    kernel produces at the very end, and the head) goes into the final bucket, reduced in `allreduce()`."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, overlap: bool = True, force: bool = False):
        """force: run the whole bucket path in a 1-rank group too (bench.py --force-buckets: measures its host / copy overhead)"""
        self.force = bool(force)
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self._by_ptr = {p.data_ptr(): p for p in self.params}
        self._hook_order: List[torch.nn.Parameter] = []         # order in which hooked gradients arrived this step
        self._learned: Optional[List[int]] = None               # data_ptrs of the hooked params, in arrival order
        self._early = 0                                          # number of leading buckets driven by the hook
        self._got: dict = {}                                     # data_ptr -> gradient tensor reported this step
        self._inflight: dict = {}                                # bucket index -> async work handle
        self._build(list(reversed(self.params)), 0)             # backward produces the last layers first
        self._views: dict = {}                                   # data_ptr -> cached view of the parameter's slot in its flat bucket
        self.grad_scale = 1.0                                    # what the optimizer must multiply gradients by (scale_in_optimizer mode)
        if overlap and dist.is_initialized() and (dist.get_world_size() > 1 or self.force):
            try:
                from . import nhwc
                nhwc.GRAD_HOOK = self._on_grad
                nhwc.GRAD_BUFFER = self._grad_buffer
                nhwc.GRAD_ASSIGN = self._grad_assign
            except Exception:                                    # host-only use (CPU tests): no conv executor
                pass

    def _build(self, order: List[torch.nn.Parameter], early_params: int) -> None:
        """buckets over `order`; the first `early_params` parameters (hook-driven) never share a bucket with the rest"""
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes = [], 0
        for i, p in enumerate(order):
            nb = p.numel() * 4
            if cur and (cur_bytes + nb > self.bucket_bytes or i == early_params):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._early = 0
        seen = 0
        for bkt in self.buckets:
            if seen + len(bkt) <= early_params:
                self._early += 1
                seen += len(bkt)
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._views = {}
        self._slot = {}
        self._bkeys: List[List[int]] = []                        # per bucket: its parameters' data_ptrs (taken once: parameters do not move)
        for bi, bkt in enumerate(self.buckets):
            off, keys = 0, []
            for p in bkt:
                k = p.data_ptr()
                self._slot[k] = (bi, off)
                keys.append(k)
                off += p.numel()
            self._bkeys.append(keys)
        self._arrived = [0] * len(self.buckets)                  # hooked gradients reported per bucket this step

    def _view_of(self, p: torch.nn.Parameter) -> torch.Tensor:
        key = p.data_ptr()
        v = self._views.get(key)
        if v is None:
            bi, off = self._slot[key]
            v = self._views[key] = self._flat_of(bi)[off:off + p.numel()].view(p.shape)
        return v

    def _grad_buffer(self, param: torch.Tensor) -> Optional[torch.Tensor]:
        """the slot of `param` in its flat bucket, once the bucket layout is final (from the second step on): the conv executor writes
        the weight gradient straight into it and the bucket needs no packing copy"""
        p = self._by_ptr.get(param.data_ptr()) if self._learned is not None else None
        if p is None or p.data_ptr() not in self._slot or p.grad is not None:
            return None                                          # a live .grad would be ACCUMULATED into by autograd: it must not alias the kernel's output
        return self._view_of(p)

    def _grad_assign(self, param: torch.Tensor, grad: torch.Tensor) -> bool:
        """a gradient the conv executor wrote into the parameter's bucket slot becomes `.grad` here (autograd is handed None for it): the
        cached view is referenced from this object too, so AccumulateGrad would otherwise clone it -- one copy launch per conv weight per step,
        reading the flat buffer while its asynchronous all-reduce may already be running"""
        p = self._by_ptr.get(param.data_ptr())
        if p is None or p.grad is not None:
            return False
        v = self._views.get(p.data_ptr())
        if v is None or v.data_ptr() != grad.data_ptr():
            return False
        p.grad = v
        return True

    def _flat_of(self, bi: int) -> torch.Tensor:
        bucket = self.buckets[bi]
        total = sum(p.numel() for p in bucket)
        dev = bucket[0].device
        flat = self._flat[bi]
        if flat is None or flat.numel() != total or flat.device != dev:
            flat = self._flat[bi] = torch.empty(total, dtype=torch.float32, device=dev)
        return flat

    def _pack(self, bi: int, grads: List[Optional[torch.Tensor]]) -> torch.Tensor:
        flat = self._flat_of(bi)
        base = flat.data_ptr()
        moved, off = [], 0
        for p, g in zip(self.buckets[bi], grads):
            n = p.numel()
            if g is None:
                g = torch.zeros(n, dtype=torch.float32, device=flat.device)
            # a gradient that already lives at its slot (written there by the conv executor through nhwc.GRAD_BUFFER, or the previous
            # step's view accumulated into in place) needs no packing
            if not (g.data_ptr() == base + 4 * off and g.dtype == torch.float32 and g.is_contiguous()):
                moved.append((off, n, g))
            off += n
        # pack the stragglers (fused sibling groups, BN vectors, FCs): one launch per RUN of adjacent slots, not one per tensor
        i = 0
        while i < len(moved):
            j = i
            while j + 1 < len(moved) and moved[j + 1][0] == moved[j][0] + moved[j][1]:
                j += 1
            o0, o1 = moved[i][0], moved[j][0] + moved[j][1]
            if j == i:
                flat[o0:o1].copy_(moved[i][2].reshape(-1))
            else:
                torch.cat([g.reshape(-1).float() for _, _, g in moved[i:j + 1]], out=flat[o0:o1])
            i = j + 1
        return flat

    def _on_grad(self, param: torch.Tensor, grad: torch.Tensor) -> None:
        key = param.data_ptr()
        if key not in self._by_ptr:
            return
        self._hook_order.append(key)
        if self._learned is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not self.force):
            return                                               # first step: only learn the order (single process: nothing to send)
        if key in self._got and self._got[key].data_ptr() != grad.data_ptr():
            #  a weight applied twice in one forward reports two partial gradients: the bucket would be sent after the first
            raise RuntimeError("parallel.GradBuckets: a parameter passed directly to more than one conv / linear op per backward pass is not "
                               "supported with the overlapped all-reduce (its bucket would leave after the first partial gradient)")
        if self._by_ptr[key].grad is not None and self._by_ptr[key].grad.data_ptr() != grad.data_ptr():
            # a live .grad means the caller accumulates over micro-batches: the early all-reduce would send only THIS micro-batch's
            # gradient and allreduce() would then replace the accumulated .grad with it
            raise RuntimeError("parallel.GradBuckets: gradient accumulation (p.grad kept between backward passes) is not supported with the "
                               "overlapped all-reduce; call optimizer.zero_grad() before every backward pass")
        bi, _ = self._slot[key]
        if key not in self._got:
            self._arrived[bi] += 1                               # (a counter, not a scan of the bucket per report: the scan was O(n^2) data_ptr() calls
        self._got[key] = grad                                    #  per step -- ~2 ms of host time on the 4-clip step, which the bucket path made host-bound)
        if bi < self._early and bi not in self._inflight and self._arrived[bi] == len(self._bkeys[bi]):
            flat = self._pack(bi, [self._got[k] for k in self._bkeys[bi]])
            self._inflight[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def allreduce(self, world: Optional[int] = None, async_op: bool = True, scale_in_optimizer: bool = False) -> None:
        """Average the gradients across ranks.  Per bucket: ONE concatenation into the flat buffer (not one copy per tensor: a
        backbone has ~220 parameter tensors, most of them BatchNorm vectors), one asynchronous all-reduce, one scale; afterwards
        every `p.grad` IS a view of the flat buffer (no copy back) -- the optimizer reads the views.  scale_in_optimizer: leave the SUM in
        the buffers and set `self.grad_scale = 1 / world` for the optimizer's fused multiply (FusedAdam.step(grad_scale=...)) instead of
        one more pass over every bucket."""
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not self.force):
            self._hook_order.clear()
            return
        world = world or dist.get_world_size()
        handles = []
        for bi, bucket in enumerate(self.buckets):
            if bi in self._inflight:                             # started from the hook while backward was still running
                handles.append((self._inflight[bi], bi))
                continue
            flat = self._pack(bi, [p.grad for p in bucket])
            handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op), bi))
        for h, bi in handles:
            if h is not None and (async_op or bi in self._inflight):
                h.wait()
            if not scale_in_optimizer:
                self._flat[bi].div_(world)
            for p, k in zip(self.buckets[bi], self._bkeys[bi]):
                v = self._views.get(k)
                if v is None:
                    v = self._view_of(p)
                if p.grad is not v:
                    p.grad = v                                   # cached view of the reduced flat buffer (no per-step view construction)
        self.grad_scale = 1.0 / world if scale_in_optimizer else 1.0
        # ---- bookkeeping for the next step
        self._inflight.clear()
        self._got.clear()
        self._arrived = [0] * len(self.buckets)
        if self._learned is None and self._hook_order:
            # every rank saw the same order (same graph): hooked weights first, in arrival order, then everything else
            self._learned = list(dict.fromkeys(self._hook_order))
            hooked = [self._by_ptr[k] for k in self._learned]
            hooked_set = set(self._learned)
            rest = [p for p in reversed(self.params) if p.data_ptr() not in hooked_set]
            self._build(hooked + rest, len(hooked))
        self._hook_order.clear()


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """One-time sync at start-up (replicas then stay identical by construction: same grads, same optimiser)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
