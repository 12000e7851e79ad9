#!/usr/bin/env python3
"""where does the bf16 mode's backbone gradient lose its cosine?  Runs the full-size reference fixture (tests/golden/
full_inv3_720x1280_b1.npz: one clip, Inception-v3, 720x1280) through the HIP path in bf16 and in fp32 storage and prints, layer by layer in
network order, the cosine of every conv-weight gradient: bf16 vs the fixture's reference values (sampled for the big tensors), bf16 vs the
HIP fp32 run (whole tensors), fp32 vs the fixture.  Usage (GPU box): python tools/bf16_grad_cosine.py > gpurun_out/r05_bf16_grad_cosine.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_din_model import _cos, _probe_idx, _run_full_case   # noqa: E402


def main():
    gpu = torch.device("cuda", 0)
    path = os.path.join(ROOT, "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "full_inv3_720x1280_b1.npz")
    # gradient MAPS, tensor by tensor in the order the reverse pass completes them: the fp32 run keeps every map, the bf16 run is compared with it
    from din_amd import nhwc
    maps32, order, rows_m = {}, [], []

    def tap32(tid, name, buf):
        if tid not in maps32:
            maps32[tid] = buf.detach().float().clone()
            order.append((tid, name))
    nhwc.GRAD_TAP = tap32
    _run_full_case(gpu, path, "fp32")
    seen = set()

    def tap16(tid, name, buf):
        if tid in seen or tid not in maps32:
            return
        seen.add(tid)
        a, b = buf.detach().float().reshape(-1), maps32[tid].reshape(-1)
        rows_m.append((name, tuple(buf.shape), float((a.double() @ b.double()) / (a.double().norm() * b.double().norm() + 1e-300)),
                       float(a.double().norm() / (b.double().norm() + 1e-300))))
    nhwc.GRAD_TAP = tap16
    z, lg16, loss16, named16, _ = _run_full_case(gpu, path, "bf16")
    nhwc.GRAD_TAP = None
    maps32.clear()
    print("# gradient maps (tensor gradient when the reverse pass reaches its first producer), loss -> image order:")
    print(f"# {'first producer reached':44s} {'map [nb, h, w, c]':>24s}  cos(bf16, fp32)   |g|_bf16/|g|_fp32")
    for name, shp, c, r in rows_m:
        print(f"  {name:44s} {str(shp):>24s}  {c:12.5f}   {r:10.4f}")

    g16 = {k: v.grad.detach().double().cpu() for k, v in named16.items() if v.grad is not None}
    del named16
    torch.cuda.empty_cache()
    z, lg32, loss32, named32, _ = _run_full_case(gpu, path, "fp32")
    g32 = {k: v.grad.detach().double().cpu() for k, v in named32.items() if v.grad is not None}
    print(f"# {os.path.basename(path)}: loss bf16 {loss16:.6f} fp32 {loss32:.6f} reference {float(z['loss']):.6f}")
    print(f"# {'parameter':58s} {'numel':>9s}  cos(bf16,ref)  cos(bf16,fp32)  cos(fp32,ref)   |g|_bf16/|g|_fp32")

    def cos_ref(g, name):
        if f"g.{name}" in z.files:
            return float(_cos(g[name], z[f"g.{name}"]))
        if f"gs.{name}" in z.files:
            fl = g[name].reshape(-1)
            return float(_cos(fl[_probe_idx(fl.numel())], z[f"gs.{name}"]))
        return float("nan")

    rows = []
    for name in g32:                                            # registration order = network order (stem first)
        if not (name.endswith("conv.weight") or not name.startswith("backbone.")):
            continue
        a, b = g16[name].reshape(-1), g32[name].reshape(-1)
        c1632 = float(a @ b / (a.norm() * b.norm() + 1e-300))
        rows.append((name, a.numel(), cos_ref(g16, name), c1632, cos_ref(g32, name), float(a.norm() / (b.norm() + 1e-300))))
    for name, n, c16r, c1632, c32r, ratio in rows:
        print(f"{name:60s} {n:9d}  {c16r:12.5f}  {c1632:13.5f}  {c32r:12.6f}   {ratio:8.4f}")
    body = [r for r in rows if r[0].startswith("backbone.")]
    worst = sorted(body, key=lambda r: r[3])[:8]
    print("# lowest cos(bf16, fp32) conv weights:", [(r[0], round(r[3], 4)) for r in worst])


if __name__ == "__main__":
    main()
