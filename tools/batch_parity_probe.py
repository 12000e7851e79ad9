#!/usr/bin/env python3
"""Eval-mode logits (and backbone maps) of clip 0 inside a B-clip batch against the same model on clip 0 alone, for growing B:
separates "a kernel chosen at large frame counts computes something else" from "the bf16 mode's sensitivity".
usage: python tools/batch_parity_probe.py [--hierarchical] [--frames T] [--dtype bf16|fp32] [--batches 1,2,4,8,16,32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--hierarchical", action="store_true")
ap.add_argument("--frames", type=int, default=None)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batches", default="2,4,8,16,32")
a = ap.parse_args()
T = a.frames or (10 if a.hierarchical else 3)
wl = "inv3_" + a.dtype
from din_amd.infer_model import Dynamic_volleyball
cfg = B.make_cfg(wl, T, 12, 720, 1280, hierarchical=a.hierarchical)
torch.manual_seed(0)
model = Dynamic_volleyball(cfg)
B.synth_weights(model)
model = model.cuda().eval()
_, _, (OH, OW), _ = B.WORKLOADS[wl]
nmax = max(int(x) for x in a.batches.split(","))
g = torch.Generator().manual_seed(1000)
images = torch.randint(0, 256, (nmax, T, 3, 720, 1280), dtype=torch.uint8, generator=g).cuda()
boxes, _ = B.synth_boxes_labels(nmax, T, 12, OH, OW, 8, seed=0)
boxes = boxes.cuda()
cap = {}
real = model.backbone.forward_nhwc
def tap(images_flat, prenormalised=False):
    bufs, graph = real(images_flat, prenormalised)
    cap["fm"] = [b[:T].float().clone() for b in bufs]
    return bufs, graph
model.backbone.forward_nhwc = tap
with torch.no_grad():
    one = model((images[:1], boxes[:1]))["activities"].float()
    fm1 = cap["fm"]
    for b in [int(x) for x in a.batches.split(",")]:
        full = model((images[:b], boxes[:b]))["activities"].float()
        d = float((full[:1] - one).abs().max() / one.abs().max())
        fd = [float((x - y).abs().max() / y.abs().max()) for x, y in zip(cap["fm"], fm1)]
        nz = [float((x != y).float().mean()) for x, y in zip(cap["fm"], fm1)]
        print(f"{a.dtype} T={T} hier={a.hierarchical} B={b:3d} ({b * T:4d} frames): clip-0 logits vs 1-clip run {d:.3e}; backbone maps of clip 0: max rel diff {['%.2e' % v for v in fd]}, "
              f"fraction of elements that differ {['%.4f' % v for v in nz]}", flush=True)
