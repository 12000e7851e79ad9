"""A training step's forward + loss + backward captured ONCE in a HIP graph and replayed (hipGraphLaunch instead of ~450 launches).

Why: at small per-GPU batches (4 clips per rank when 32 clips are sharded over 8 GPUs) the step is host-bound -- ~10 ms of Python /
ctypes / allocator work to enqueue ~10 ms of kernels.  A replayed graph costs the host ~0.1 ms, and all of a step's allocations come
out of the graph's private pool, at fixed addresses.

What is captured: `loss = loss_fn()` and `loss.backward()` -- every kernel of this package is a plain launch on torch's current stream,
which stream capture records; the library neither allocates nor synchronises.  What is NOT captured: the gradient all-reduce (RCCL calls
stay ordinary, eager calls on the buckets the captured kernels wrote into: `parallel.GradBuckets.allreduce()`) and the optimizer (one
launch).  Per-step host state that a capture would freeze:
  * dropout seeds: computed on the host and baked into the launches; their per-step part therefore lives in device memory
    (`ops.SEED_OFFSET`, added to the seed by the kernels) and the captured step advances it once per replay (din_counter_add);
  * the input batch: replays read the tensors `loss_fn` closed over -- copy a new batch INTO them (`static.copy_(batch)`).
The sibling-pacing tags of the pipelined wgrad kernel are baked too; pacing is a fetch-traffic hint only (never correctness).

Replays reproduce the eager step bit for bit (same kernels, same order, same seeds while SEED_OFFSET is zero):
tests/test_gpu_din_model.py::test_captured_step_matches_eager.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib as L
from . import nhwc, ops

# What a replay adds to the device-side seed offset (odd: full period).  It must NOT be congruent to ops.mask_seed's per-rank increment
# (0x9E3779B97F4A7C15) mod 2^63: round 3 used that very constant, so rank r at replay k drew the masks rank 0 drew at replay k + r
# (ADVICE r3).  tests/test_host_cpu.py::test_dropout_seeds_distinct_over_ranks_and_replays pins the property.
SEED_STRIDE = 0xD1342543DE82EF95 & 0x7FFFFFFFFFFFFFFF


class CapturedStep:
    """cap = CapturedStep(loss_fn, params); then per step: `loss = cap.replay()`; gradients are in `p.grad` (static tensors of the
    graph's pool, or the all-reduce bucket slots when parallel.GradBuckets hands them out).

    loss_fn() -> scalar loss, reading only tensors that stay alive and in place between replays.  The caller must have run at least one
    eager step of the same shape before (workspaces, packed-filter caches, LDS limits and the bucket layout are set up by it)."""

    def __init__(self, loss_fn: Callable[[], torch.Tensor], params: Sequence[torch.nn.Parameter], counters: Sequence[object] = ()):
        """counters: objects with a `_step` attribute (the modules' host-side dropout counters); they are advanced per replay by what the
        captured step advanced them, so a checkpoint written after replays resumes the eager sequence position."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.loss_fn = loss_fn
        dev = self.params[0].device
        self.seed_offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self.graph = torch.cuda.CUDAGraph()
        self.counters = list(counters)
        self.replays = 0
        hook, nhwc.GRAD_HOOK = nhwc.GRAD_HOOK, None             # no collective may start inside the capture
        prev_offset, ops.SEED_OFFSET = ops.SEED_OFFSET, self.seed_offset
        before = [c._step for c in self.counters]
        for p in self.params:
            p.grad = None                                       # autograd then installs fresh tensors from the graph's pool
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(self.graph):
                loss = loss_fn()
                loss.backward()
                L.check(L.load().din_counter_add(C.c_void_p(self.seed_offset.data_ptr()), SEED_STRIDE,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "counter_add")
        finally:
            nhwc.GRAD_HOOK = hook
            ops.SEED_OFFSET = prev_offset
        self.loss = loss.detach()
        self.grads: List[Optional[torch.Tensor]] = [p.grad for p in self.params]
        self.counter_delta = [c._step - b for c, b in zip(self.counters, before)]
        # the capture itself ran nothing: gradients and loss hold garbage until the first replay
        self.seed_offset.zero_()

    def replay(self) -> torch.Tensor:
        for p, g in zip(self.params, self.grads):
            p.grad = g                                          # (an eager step or the bucket path may have re-pointed .grad)
        self.graph.replay()
        if self.replays:                                        # the first replay IS the step the counters already counted at capture
            for c, d in zip(self.counters, self.counter_delta):
                c._step += d
        self.replays += 1
        return self.loss


def dropout_counters(model: torch.nn.Module) -> List[object]:
    """every module of `model` that owns a host-side dropout counter (`_step`)"""
    return [m for m in model.modules() if isinstance(getattr(m, "_step", None), int)]
