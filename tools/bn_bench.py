#!/usr/bin/env python3
"""Micro-benchmark of the four batch-statistics BatchNorm stream kernels (csrc/bn.hip) on the views the Inception-v3 trunk gives them at
96 frames of 720x1280 (bench.py --bn-mode batch): GB/s of algorithmic traffic per launch against the ~6.3 TB/s a device copy reaches.
usage: python tools/bn_bench.py [--iters 20]"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L  # noqa: E402

SHAPES = [  # (name, rows, c, ld of the gradient view, coff)
    ("Conv2d_1a 359x639x32", 96 * 359 * 639, 32, 32, 0),
    ("Conv2d_2b 357x637x64", 96 * 357 * 637, 64, 64, 0),
    ("Conv2d_3b 176x316x80", 96 * 176 * 316, 80, 80, 0),
    ("Conv2d_4a 174x314x192", 96 * 174 * 314, 192, 192, 0),
    ("Mixed_5b.branch5x5_1 87x157x48 (view of 288)", 96 * 87 * 157, 48, 288, 64),
    ("Mixed_5c.branch3x3dbl_3 87x157x96 (view of 288)", 96 * 87 * 157, 96, 288, 128),
    ("Mixed_6c.branch7x7_2 43x78x160", 96 * 43 * 78, 160, 160, 0),
    ("Mixed_6e.branch1x1 43x78x192 (view of 768)", 96 * 43 * 78, 192, 768, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib = L.load()
    dt, tdt = L.DIN_BF16, torch.bfloat16
    print(f"{'view':52s} {'stats':>9s} {'apply':>9s} {'bwd_stats':>9s} {'bwd_apply':>9s}   (GB/s of algorithmic bytes; us per launch)")
    tot = [0.0] * 4
    for name, rows, c, ldg, coff in SHAPES:
        y = torch.randn(rows, c, device="cuda").to(tdt)
        z = torch.empty(rows, ldg, dtype=tdt, device="cuda")
        gz = torch.randn(rows, ldg, device="cuda").to(tdt)
        dy = torch.empty(rows, c, dtype=tdt, device="cuda")
        ws = torch.empty(lib.din_bn_workspace(rows, c) // 8, dtype=torch.float64, device="cuda")
        gam, bet, rm, rv = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        a_, b_, mean, rstd, dg, db = (torch.empty(c, device="cuda") for _ in range(6))
        P = lambda t: t.data_ptr()                                                                  # noqa: E731
        nparts = lib.din_bn_parts(rows)

        def stats():
            L.check(lib.din_bn_stats(P(y), dt, rows, c, c, 0, None, P(ws), None))

        def fin():
            L.check(lib.din_bn_finalize(P(ws), nparts, rows, c, P(gam), P(bet), 1e-3, 0.1, P(rm), P(rv), P(a_), P(b_), P(mean), P(rstd), None, None))

        def apply():
            L.check(lib.din_bn_apply(P(y), dt, rows, c, c, 0, P(a_), P(b_), 1, P(z), ldg, coff, None))

        def bstats():
            L.check(lib.din_bn_bwd_stats(P(gz), ldg, coff, P(y), c, 0, dt, rows, c, P(mean), P(rstd), P(ws), None))

        def bapply():
            L.check(lib.din_bn_bwd_apply(P(gz), ldg, coff, P(y), c, 0, dt, rows, c, P(gam), P(mean), P(rstd), P(ws), P(dy), c, 0, P(dg), P(db), None))
        stats(); fin()
        res = []
        for fn, nbytes in ((stats, rows * c * 2), (apply, rows * c * 4), (bstats, rows * c * 4), (bapply, rows * c * 6)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            res.append((nbytes / us / 1e3, us))
        for i, (_, us) in enumerate(res):
            tot[i] += us
        print(f"{name:52s} " + " ".join(f"{g:5.0f}/{u:<6.0f}" for g, u in res))
        del y, z, gz, dy
    print("sum of the listed launches (us):", " ".join(f"{t:9.0f}" for t in tot))


if __name__ == "__main__":
    main()
