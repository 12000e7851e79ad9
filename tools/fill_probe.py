#!/usr/bin/env python3
"""Who launches the fill kernels of a training step?  torch.profiler over 3 small-batch steps: aten::fill_ / aten::zero_ / aten::zeros* calls
grouped by their Python call site (diagnostic for the FillFunctor launches in profiles/*kernel_stats.csv)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

def main():
    sys.argv = ["bench.py", "--global-batch", "4", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        bench.main()
    log = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::fill_", "aten::zero_"):
            st = [s for s in ev.stack if "din" in s or "bench" in s or "torch/autograd" in s]
            site = st[0] if st else (ev.stack[0] if ev.stack else "?")
            par = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
            log[(ev.name, par, site[-90:], tuple(ev.input_shapes[0]) if ev.input_shapes else None)] += 1
    for k, v in sorted(log.items(), key=lambda kv: -kv[1])[:60]:
        print(v, k)
main()
