#!/usr/bin/env python3
"""Weight-gradient kernel variants on the Inception layer shapes (bf16, through the C ABI): the round-1 ring kernel (DIN_WGRAD_PIPE=0), the
software-pipelined 32x32x16 kernel (default) and its atomic-accumulate epilogue (DIN_WGRAD_ATOMIC=1), at the 32-clip (96 frames) and 4-clip
(12 frames) batch.  Prints time incl. the slice reduce, TFLOP/s and the max-rel difference of dW against the ring kernel's."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L

LAYERS = {  # name: (h, w, cin, cout, k, s, p)
    "Conv2d_4a_3x3": (178, 318, 80, 192, (3, 3), 1, (0, 0)),
    "Mixed_6a.branch3x3": (87, 157, 288, 384, (3, 3), 2, (0, 0)),
    "Mixed_6e.7x7dbl_3 (1x7)": (43, 78, 192, 192, (1, 7), 1, (0, 3)),
    "Mixed_6e.7x7dbl_2 (7x1)": (43, 78, 192, 192, (7, 1), 1, (3, 0)),
    "Mixed_6c.7x7_3 (7x1 160)": (43, 78, 160, 192, (7, 1), 1, (3, 0)),
    "Mixed_6e.branch1x1": (43, 78, 768, 192, (1, 1), 1, (0, 0)),
    "Mixed_6e.7x7_1+dbl_1": (43, 78, 768, 384, (1, 1), 1, (0, 0)),
    "Mixed_6b.7x7_1+dbl_1": (43, 78, 768, 256, (1, 1), 1, (0, 0)),
    "Mixed_6c.7x7_2 (1x7 160)": (43, 78, 160, 160, (1, 7), 1, (0, 3)),
    "Mixed_6c.7x7dbl_2 (7x1 160)": (43, 78, 160, 160, (7, 1), 1, (3, 0)),
    "Mixed_6c.7x7_1+dbl_1": (43, 78, 768, 320, (1, 1), 1, (0, 0)),
    "Mixed_6b.7x7_2 (1x7 128)": (43, 78, 128, 128, (1, 7), 1, (0, 3)),
    "Mixed_5d.5x5_1+3x3dbl_1": (87, 157, 288, 112, (1, 1), 1, (0, 0)),
    "Mixed_5c.3x3dbl_3": (87, 157, 96, 96, (3, 3), 1, (1, 1)),
}
MODES = [("ring", {"DIN_WGRAD_PIPE": "0"}), ("pipe-r1tiles", {"DIN_WGRAD_PIPE": "3", "DIN_WGRAD_ATOMIC": "0"}),
         ("pipe", {"DIN_WGRAD_PIPE": "1", "DIN_WGRAD_ATOMIC": "0"})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--frames", type=int, nargs="*", default=[96, 12])
    a = ap.parse_args()
    lib = L.load()
    for nb in a.frames:
        for name, (h, w, cin, cout, k, s, p) in LAYERS.items():
            oh, ow = (h + 2 * p[0] - k[0]) // s + 1, (w + 2 * p[1] - k[1]) // s + 1
            d = L.ConvDesc()
            d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, oh, ow, cout
            d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = k[0], k[1], s, s, p[0], p[1], 1, 1
            d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin, 0, cout, 0, L.DIN_BF16
            g = torch.Generator(device="cuda").manual_seed(1)
            x = torch.randn(nb, h, w, cin, device="cuda", generator=g).relu().bfloat16()
            gy = torch.randn(nb, oh, ow, cout, device="cuda", generator=g).bfloat16()
            fl = 2.0 * nb * oh * ow * cout * cin * k[0] * k[1]
            ref = None
            line = f"{name:26s} nb={nb:3d} "
            for mode, env in MODES:
                os.environ.update(env)
                bm, bn = C.c_int32(0), C.c_int32(0)
                lib.din_conv_kernel_tile(C.byref(d), 2, C.byref(bm), C.byref(bn))
                dw = torch.empty(cout, cin, k[0], k[1], device="cuda")
                db = torch.empty(cout, device="cuda")
                wsb = lib.din_conv_workspace_bytes(C.byref(d), 2)
                ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
                run = lambda: L.check(lib.din_conv_wgrad(C.byref(d), x.data_ptr(), gy.data_ptr(), dw.data_ptr(), db.data_ptr(), None, None, None,
                                                          0, ws.data_ptr(), wsb, None))
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
                if ref is None:
                    ref = (dw.clone(), db.clone())
                    err = ""
                else:
                    ew = ((dw - ref[0]).abs().max() / ref[0].abs().max()).item()
                    eb = ((db - ref[1]).abs().max() / ref[1].abs().max()).item()
                    err = f" (dW {ew:.1e} db {eb:.1e})"
                line += f"| {mode} [{bm.value}x{bn.value % 1000}{'p' if bn.value >= 2000 else 'r' if bn.value >= 1000 else ''}] {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF{err} "
            print(line, flush=True)


if __name__ == "__main__":
    main()
