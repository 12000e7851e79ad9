#!/usr/bin/env python3
"""conv1x1_stream_kernel against the 128-pixel implicit-GEMM kernel on the Inception 1x1 shapes of the default workload (bf16, 96 frames):
forward (bias + ReLU), plain dgrad, and the fused multi-source dgrads of Mixed_5b / 5c / 5d (mask + no accumulate)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def desc(nb, h, w, cin, cout, ldi=None, ldo=None):
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, h, w, cout
    d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1
    d.ph = d.pw = 0
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = ldi or cin, 0, ldo or cout, 0, L.DIN_BF16
    return d


def main():
    lib = L.load()
    bf = torch.bfloat16
    nb = 96
    rows = []
    # (name, h, w, cin, cout)
    for name, h, w, cin, cout in (("Conv2d_3b", 176, 316, 64, 80), ("5b heads 192->176", 87, 157, 192, 176), ("5c heads 256->176", 87, 157, 256, 176),
                                  ("5d heads 288->176", 87, 157, 288, 176), ("5b pool 192->32", 87, 157, 192, 32), ("5d pool 288->64", 87, 157, 288, 64),
                                  ("6a dbl_1 288->64", 87, 157, 288, 64)):
        d = desc(nb, h, w, cin, cout)
        x = torch.randn(nb, h, w, cin, device="cuda").to(bf)
        wt = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05)
        bias = torch.randn(cout, device="cuda")
        wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), dtype=bf, device="cuda")
        L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpk.data_ptr(), 0, None))
        wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=bf, device="cuda")
        L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpt.data_ptr(), 1, None))
        out = torch.empty(nb, h, w, cout, device="cuda", dtype=bf)
        gz = torch.randn(nb, h, w, cout, device="cuda").to(bf)
        dx = torch.empty(nb, h, w, cin, device="cuda", dtype=bf)
        ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
        res = {}
        for mode in ("0", "2"):
            os.environ["DIN_CONV_STREAM"] = mode
            f = bench(lambda: L.check(lib.din_conv_fwd(C.byref(d), x.data_ptr(), wpk.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                       L.CONV_BIAS | L.CONV_RELU, ws.data_ptr(), 1 << 20, None)))
            o1 = out.clone()
            b = bench(lambda: L.check(lib.din_conv_dgrad(C.byref(d), gz.data_ptr(), wpt.data_ptr(), dx.data_ptr(), x.data_ptr(), cin, 0,
                                                         L.CONV_MASK, ws.data_ptr(), 1 << 20, None)))
            res[mode] = (f, b, o1, dx.clone())
        fb = (x.numel() + out.numel()) * 2
        bb = (gz.numel() + 2 * dx.numel()) * 2
        err_f = float((res["0"][2].float() - res["2"][2].float()).abs().max()); err_b = float((res["0"][3].float() - res["2"][3].float()).abs().max())
        print(f"{name:22s} fwd {res['0'][0]:7.1f} -> {res['2'][0]:7.1f} us ({fb / res['2'][0] / 1e6:5.2f} TB/s)   dgrad+mask {res['0'][1]:7.1f} -> {res['2'][1]:7.1f} us "
              f"({bb / res['2'][1] / 1e6:5.2f} TB/s)   max |diff| {err_f:.3g} {err_b:.3g}", flush=True)
    # fused multi-source dgrads: sources (branch_pool, 3x3dbl_1, 5x5_1, 1x1) -> the block input
    h, w = 87, 157
    for name, cin, couts in (("Mixed_5b", 192, (32, 64, 48, 64)), ("Mixed_5c", 256, (64, 64, 48, 64)), ("Mixed_5d", 288, (64, 64, 48, 64))):
        srcs = (L.ConvSrc * 4)()
        keep = []
        for i, co in enumerate(couts):
            d = desc(nb, h, w, cin, co)
            wt = torch.randn(co, cin, 1, 1, device="cuda") * 0.05
            wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=bf, device="cuda")
            L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpt.data_ptr(), 1, None))
            gz = torch.randn(nb, h, w, co, device="cuda").to(bf)
            keep += [wpt, gz]
            srcs[i].dout, srcs[i].wpk_t, srcs[i].cout, srcs[i].ldo, srcs[i].cooff = gz.data_ptr(), wpt.data_ptr(), co, co, 0
        xm = torch.randn(nb, h, w, cin, device="cuda").to(bf)
        dx = torch.empty(nb, h, w, cin, device="cuda", dtype=bf)
        res = {}
        for mode in ("0", "2"):
            os.environ["DIN_CONV_STREAM"] = mode
            t = bench(lambda: L.check(lib.din_conv1x1_dgrad_multi(4, srcs, L.DIN_BF16, nb, h, w, cin, cin, 0, dx.data_ptr(), xm.data_ptr(), cin, 0,
                                                                   L.CONV_MASK, None)))
            res[mode] = (t, dx.clone())
        by = (sum(couts) + 2 * cin) * nb * h * w * 2
        err = float((res["0"][1].float() - res["2"][1].float()).abs().max())
        print(f"{name} multi dgrad {res['0'][0]:7.1f} -> {res['2'][0]:7.1f} us ({by / res['2'][0] / 1e6:5.2f} TB/s)   max |diff| {err:.3g}", flush=True)


if __name__ == "__main__":
    main()
