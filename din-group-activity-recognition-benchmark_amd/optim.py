"""FusedAdam: torch.optim.Adam semantics (train_net_dynamic.py:104: lr, weight_decay, default betas/eps) with the update
of each parameter tensor done by one din_adam_step launch (fp32 moments).  Exposes `param_groups[i]['lr']` so the
reference's adjust_lr (train_net_dynamic.py:22-25) works unchanged."""
from __future__ import annotations

import torch

from . import ops


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.state = {}
        self.step_count = 0

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
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = (torch.zeros_like(p, dtype=torch.float32), torch.zeros_like(p, dtype=torch.float32))
            ops.adam_step(p.data, p.grad.contiguous(), st[0], st[1], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                          g["weight_decay"], self.step_count, grad_scale)

    def state_dict(self):
        return dict(step=self.step_count, lr=self.param_groups[0]["lr"],
                    moments=[tuple(t.cpu() for t in self.state[p]) if p in self.state else None for p in self.params])
