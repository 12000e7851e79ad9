"""FusedAdam: torch.optim.Adam semantics (train_net_dynamic.py:104: lr, weight_decay, default betas/eps) with the update
of the WHOLE parameter list done by one din_adam_step_multi launch (fp32 moments; pointer / chunk tables live on the
device, only the gradient addresses are refreshed per step).  Exposes `param_groups[i]['lr']` so the reference's
adjust_lr (train_net_dynamic.py:22-25) works unchanged."""
from __future__ import annotations

import torch

import ctypes as C

from . import _lib as L
from . import ops

CHUNK = 8192          # elements per workgroup of the multi-tensor launch


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.state = {}
        self.step_count = 0
        self._tables = None     # (key, ptrs_cpu, ptrs_dev, sizes_dev, chunk_tensor_dev, chunk_index_dev, nchunks)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        g = self.param_groups[0]
        self.step_count += 1
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        for p in live:
            if p not in self.state:
                self.state[p] = (torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32))
        fused = all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in live)
        if not fused:
            for p in live:
                st = self.state[p]
                ops.adam_step(p.data, p.grad.contiguous(), st[0], st[1], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                              g["weight_decay"], self.step_count, grad_scale)
            return
        dev = live[0].device
        key = tuple(id(p) for p in live)
        if self._tables is None or self._tables[0] != key:
            sizes = torch.tensor([p.numel() for p in live], dtype=torch.int64)
            ct, ci = [], []
            for t, p in enumerate(live):
                nchunk = (p.numel() + CHUNK - 1) // CHUNK
                ct += [t] * nchunk
                ci += list(range(nchunk))
            ptrs_cpu = torch.zeros((len(live), 4), dtype=torch.int64).pin_memory()
            for t, p in enumerate(live):
                ptrs_cpu[t, 0], ptrs_cpu[t, 2], ptrs_cpu[t, 3] = p.data_ptr(), self.state[p][0].data_ptr(), self.state[p][1].data_ptr()
            self._tables = (key, ptrs_cpu, torch.empty((len(live), 4), dtype=torch.int64, device=dev), sizes.to(dev),
                            torch.tensor(ct, dtype=torch.int32, device=dev), torch.tensor(ci, dtype=torch.int32, device=dev), len(ct))
        _, ptrs_cpu, ptrs_dev, sizes_dev, ct_dev, ci_dev, nchunks = self._tables
        grads = [p.grad if p.grad.is_contiguous() and p.grad.dtype == torch.float32 else p.grad.float().contiguous() for p in live]
        if getattr(self, "_copied", None) is not None:
            self._copied.synchronize()          # the previous step's table upload must have left the pinned buffer
        ptrs_cpu[:, 1] = torch.tensor([gr.data_ptr() for gr in grads], dtype=torch.int64)
        ptrs_dev.copy_(ptrs_cpu, non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record()
        L.check(L.load().din_adam_step_multi(C.c_void_p(ptrs_dev.data_ptr()), C.c_void_p(sizes_dev.data_ptr()),
                                             C.c_void_p(ct_dev.data_ptr()), C.c_void_p(ci_dev.data_ptr()), nchunks, CHUNK,
                                             g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count,
                                             grad_scale, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "adam_step_multi")
        self._keep = grads      # the launch reads these asynchronously

    def state_dict(self):
        return dict(step=self.step_count, lr=self.param_groups[0]["lr"],
                    moments=[tuple(t.cpu() for t in self.state[p]) if p in self.state else None for p in self.params])
