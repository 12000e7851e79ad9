#!/usr/bin/env python3
"""Pool / resize kernels on the Inception shapes of the default workload (bf16, 96 frames): time and effective HBM rate (bytes in + out once)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L


def bench(fn, iters=int(os.environ.get("DIN_POOL_ITERS", "20"))):      # DIN_POOL_ITERS=2000: steady-state clocks (tools/clock_probe.sh)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    lib = L.load()
    nb = 96
    bf = torch.bfloat16
    # multiscale fuse: Mixed_6e 43x78x768 -> channels [288, 1056) of the 87x157 fused map
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = nb, 43, 78, 768, 87, 157
    d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = 1, 1, 0, 768, 0, 1056, 288, L.DIN_BF16
    x = torch.randn(nb, 43, 78, 768, device="cuda").to(bf)
    out = torch.empty(nb, 87, 157, 1056, device="cuda", dtype=bf)
    gx = torch.empty_like(x)
    by = (x.numel() + nb * 87 * 157 * 768) * 2
    for mode in ("0", "1"):
        os.environ["DIN_BILINEAR_CELLS"] = mode
        ms = bench(lambda: L.check(lib.din_bilinear_fwd(C.byref(d), x.data_ptr(), out.data_ptr(), None)))
        print(f"bilinear fwd cells={mode}: {ms * 1e3:8.1f} us  {by / ms / 1e9:6.2f} TB/s")
    for mode in ("0", "1"):
        os.environ["DIN_BILINEAR_HOIST"] = mode
        ms = bench(lambda: L.check(lib.din_bilinear_bwd(C.byref(d), out.data_ptr(), gx.data_ptr(), x.data_ptr(), 0, None)))
        print(f"bilinear bwd hoist={mode}: {ms * 1e3:8.1f} us  {(by + x.numel() * 2) / ms / 1e9:6.2f} TB/s (incl. mask read)")
    # max-pools 3x3 / 2: after Conv2d_2b (64 ch, 357x637), after Conv2d_4a (192 ch, 176x316), Mixed_6a pool branch (288 ch, 87x157)
    for (c, h, w) in ((64, 357, 637), (192, 176, 316), (288, 87, 157)):
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        p = L.PoolDesc()
        p.nb, p.h, p.w, p.c, p.oh, p.ow = nb, h, w, c, oh, ow
        p.k, p.stride, p.pad, p.ldi, p.cioff, p.ldo, p.cooff, p.dtype = 3, 2, 0, c, 0, c, 0, L.DIN_BF16
        xi = torch.randn(nb, h, w, c, device="cuda").relu().to(bf)
        yo = torch.empty(nb, oh, ow, c, device="cuda", dtype=bf)
        am = torch.empty(nb, oh, ow, c, device="cuda", dtype=torch.uint8)
        gi = torch.empty_like(xi)
        b1 = xi.numel() * 2 + yo.numel() * 3
        for rows in ("0", "1"):                                   # element-per-thread (+ strip) kernels / row kernels
            os.environ["DIN_MAXPOOL_ROWS"] = rows
            ms = bench(lambda: L.check(lib.din_maxpool_fwd(C.byref(p), xi.data_ptr(), yo.data_ptr(), am.data_ptr(), None)))
            print(f"maxpool fwd rows={rows} {c:3d}ch {h}x{w}: {ms * 1e3:8.1f} us  {b1 / ms / 1e9:6.2f} TB/s")
            ms = bench(lambda: L.check(lib.din_maxpool_bwd(C.byref(p), xi.data_ptr(), am.data_ptr(), yo.data_ptr(), gi.data_ptr(), 1, 0, None)))
            print(f"maxpool bwd rows={rows} {c:3d}ch {h}x{w}: {ms * 1e3:8.1f} us  {b1 / ms / 1e9:6.2f} TB/s")
        os.environ.pop("DIN_MAXPOOL_ROWS")
    # 3x3 / 1 / pad 1 average pools behind the commuted branch_pool 1x1 convs: (channels, h, w, pixel stride of the destination view)
    for (c, h, w, ldo) in ((32, 87, 157, 256), (64, 87, 157, 288), (64, 87, 157, 1056), (192, 43, 78, 768)):
        p = L.PoolDesc()
        p.nb, p.h, p.w, p.c, p.oh, p.ow = nb, h, w, c, h, w
        p.k, p.stride, p.pad, p.ldi, p.cioff, p.ldo, p.cooff, p.dtype = 3, 1, 1, c, 0, ldo, ldo - c, L.DIN_BF16
        xi = torch.randn(nb, h, w, c, device="cuda").to(bf)
        yo = torch.empty(nb, h, w, ldo, device="cuda", dtype=bf)
        bias = torch.randn(c, device="cuda")
        q = L.PoolDesc()        # backward: dout is the strided view, din the dense conv output gradient
        q.nb, q.h, q.w, q.c, q.oh, q.ow = nb, h, w, c, h, w
        q.k, q.stride, q.pad, q.ldi, q.cioff, q.ldo, q.cooff, q.dtype = 3, 1, 1, c, 0, ldo, ldo - c, L.DIN_BF16
        b1 = xi.numel() * 4
        for cap in ("32768",):
            ms = bench(lambda: L.check(lib.din_avgpool_fwd(C.byref(p), xi.data_ptr(), yo.data_ptr(), bias.data_ptr(), L.CONV_BIAS | L.CONV_RELU, None)))
            print(f"avgpool fwd cap={cap:10s} {c:3d}ch {h}x{w} ldo {ldo}: {ms * 1e3:8.1f} us  {b1 / ms / 1e9:6.2f} TB/s")
            ms = bench(lambda: L.check(lib.din_avgpool_bwd(C.byref(q), yo.data_ptr(), xi.data_ptr(), None, 0, None)))
            print(f"avgpool bwd cap={cap:10s} {c:3d}ch {h}x{w} ldo {ldo}: {ms * 1e3:8.1f} us  {b1 / ms / 1e9:6.2f} TB/s")


if __name__ == "__main__":
    main()
