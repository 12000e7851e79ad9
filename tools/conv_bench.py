#!/usr/bin/env python3
"""Single-layer microbenchmark of the conv kernels through the C ABI (used for tuning + rocprofv3 --pmc runs)."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L

LAYERS = {  # name: (nb, h, w, cin, cout, k, s, p)
    "vgg_conv1_2": (12, 720, 1280, 64, 64, 3, 1, 1),
    "vgg_conv2_2": (12, 360, 640, 128, 128, 3, 1, 1),
    "vgg_conv3_2": (12, 180, 320, 256, 256, 3, 1, 1),
    "vgg_conv4_2": (12, 90, 160, 512, 512, 3, 1, 1),
    "vgg_conv5_2": (12, 45, 80, 512, 512, 3, 1, 1),
    "inc_5b_1x1": (96, 87, 157, 192, 64, 1, 1, 0),
    "inc_5d_3x3": (96, 87, 157, 96, 96, 3, 1, 1),
    "inc_6b_1x1": (96, 43, 78, 768, 192, 1, 1, 0),
    "inc_6c_1x7": (96, 43, 78, 160, 160, (1, 7), 1, (0, 3)),
    "inc_2b_3x3": (96, 357, 637, 32, 64, 3, 1, 1),
    "inc_2a_3x3": (96, 359, 639, 32, 32, 3, 1, 0),
    "inc_5b_5x5": (96, 87, 157, 48, 64, 5, 1, 2),
    "inc_3b_1x1": (96, 178, 318, 64, 80, 1, 1, 0),
    "inc_5d_1x1_288": (96, 87, 157, 288, 64, 1, 1, 0),
    "inc_6a_3x3dbl1": (96, 87, 157, 288, 64, 1, 1, 0),
    "inc_6e_1x1_768": (96, 43, 78, 768, 192, 1, 1, 0),
    "inc_4a_3x3": (96, 178, 318, 80, 192, 3, 1, 0),
    "inc_6e_7x1": (96, 43, 78, 192, 192, (7, 1), 1, (3, 0)),
    "inc_6a_3x3": (96, 87, 157, 288, 384, 3, 2, 0),
    "k_1x1_384_288": (96, 43, 78, 384, 288, 1, 1, 0),
    "k_1x1_288_384": (96, 43, 78, 288, 384, 1, 1, 0),
    "k_1x1_288_384_s2": (96, 87, 157, 288, 384, 1, 2, 0),
    "inc_6a_dbl3": (96, 87, 157, 96, 96, 3, 2, 0),
    "inc_3b_1x1_80": (96, 178, 318, 64, 80, 1, 1, 0),
    "k_7x1_64": (96, 43, 78, 64, 192, (7, 1), 1, (3, 0)),
    "k_7x1_384": (96, 43, 78, 384, 192, (7, 1), 1, (3, 0)),
    "k_7x1_768": (96, 43, 78, 768, 192, (7, 1), 1, (3, 0)),
    "k_1x1_192": (96, 43, 78, 192, 192, 1, 1, 0),
    "k_1x1_1536": (96, 43, 78, 1536, 192, 1, 1, 0),
    "inc_6e_1x7": (96, 43, 78, 192, 192, (1, 7), 1, (0, 3)),
    "inc_6b_1x7": (96, 43, 78, 128, 128, (1, 7), 1, (0, 3)),
    "inc_6b_7x1": (96, 43, 78, 128, 128, (7, 1), 1, (3, 0)),
    "inc_6c_7x1": (96, 43, 78, 160, 160, (7, 1), 1, (3, 0)),
    "inc_6c_7x1_192": (96, 43, 78, 160, 192, (7, 1), 1, (3, 0)),
    "k_1x1_768_64": (96, 43, 78, 768, 64, 1, 1, 0),
    "k_1x1_1536_64": (96, 43, 78, 1536, 64, 1, 1, 0),
    "inc_5b_entry_176": (96, 87, 157, 192, 176, 1, 1, 0),     # Mixed_5b sibling group 192 -> 64 + 48 + 64
    "inc_5c_entry_176": (96, 87, 157, 256, 176, 1, 1, 0),
    "inc_5d_entry_176": (96, 87, 157, 288, 176, 1, 1, 0),
    "inc_5c_pool_64": (96, 87, 157, 256, 64, 1, 1, 0),
    "k_1x1_256_240": (96, 87, 157, 256, 240, 1, 1, 0),        # dgrad: the shape of the Mixed_5c block-entry data gradient (one source)
    "inc_6e_entry_576": (96, 43, 78, 768, 576, 1, 1, 0),      # the sibling group 768 -> 192 + 192 + 192 as one bank
    "inc_6b_entry_448": (96, 43, 78, 768, 448, 1, 1, 0),
    "k_1x1_768_768": (96, 43, 78, 768, 768, 1, 1, 0),         # dgrad: the shape of the Mixed_6e block-entry data gradient (one source)
    "k_1x1_768_768_b4": (12, 43, 78, 768, 768, 1, 1, 0),
    "inc_6e_1x1_768_b4": (12, 43, 78, 768, 192, 1, 1, 0),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="vgg_conv3_2")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--which", default="fwd", choices=["fwd", "dgrad", "wgrad"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--flags", type=int, default=8, help="dgrad epilogue flags (8 = ReLU mask, 4 = accumulate)")
    ap.add_argument("--no-dbias", action="store_true", help="wgrad without the bias gradient (what do its per-workgroup atomics cost?)")
    ap.add_argument("--const", action="store_true", help="constant operands (low toggle rate): shows how much the clock sags on random data")
    ap.add_argument("--relu", action="store_true", help="half of the activations / gradients zero, as behind a ReLU (what the mid-network layers see)")
    a = ap.parse_args()
    lib = L.load()
    torch.manual_seed(0)                                  # (same operands in every process: the checksums of an A/B pair are comparable)
    nb, h, w, cin, cout, k, s, p = LAYERS[a.layer]
    k = (k, k) if isinstance(k, int) else k
    p = (p, p) if isinstance(p, int) else p
    oh = (h + 2 * p[0] - k[0]) // s + 1
    ow = (w + 2 * p[1] - k[1]) // s + 1
    dt = L.DIN_BF16 if a.dtype == "bf16" else L.DIN_F32
    tdt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, oh, ow, cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = k[0], k[1], s, s, p[0], p[1], 1, 1
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin, 0, cout, 0, dt
    x = torch.randn(nb, h, w, cin, device="cuda").to(tdt)
    gy = torch.randn(nb, oh, ow, cout, device="cuda").to(tdt)
    wt = torch.randn(cout, cin, k[0], k[1], device="cuda") * 0.05
    bias = torch.zeros(cout, device="cuda")
    if a.relu:
        x = torch.relu(x); gy = gy * (torch.rand_like(gy, dtype=torch.float32) > 0.5).to(tdt)
    if a.const:
        x.fill_(1.0); gy.fill_(1.0); wt.fill_(0.03125)
    y = torch.empty(nb, oh, ow, cout, device="cuda", dtype=tdt)
    dx = torch.empty_like(x)
    dw = torch.empty_like(wt)
    wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), dtype=tdt, device="cuda")
    wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=tdt, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpk.data_ptr(), 0, None))
    L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpt.data_ptr(), 1, None))
    which = {"fwd": 0, "dgrad": 1, "wgrad": 2}[a.which]
    wsb = lib.din_conv_workspace_bytes(C.byref(d), which)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    def run():
        if a.which == "fwd":
            L.check(lib.din_conv_fwd(C.byref(d), x.data_ptr(), wpk.data_ptr(), bias.data_ptr(), y.data_ptr(), 3, ws.data_ptr(), wsb, None))
        elif a.which == "dgrad":
            L.check(lib.din_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx.data_ptr(), x.data_ptr(), cin, 0, a.flags, ws.data_ptr(), wsb, None))
        else:
            L.check(lib.din_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), None if a.no_dbias else bias.data_ptr(), None, None, None, 0, ws.data_ptr(), wsb, None))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = 2.0 * nb * oh * ow * cout * cin * k[0] * k[1]
    chk = {"fwd": y, "dgrad": dx, "wgrad": dw}[a.which].float().abs().sum().item()
    print(f"{a.layer:14s} {a.dtype} {a.which:5s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s  (M={nb*oh*ow}, K={cin*k[0]*k[1]}, N={cout})  |out|_1 = {chk:.6e}", flush=True)

if __name__ == "__main__":
    main()
