#!/usr/bin/env python3
"""conv_wgrad_halo_kernel (dW stationary in registers, halo tiles) against the general weight-gradient kernels on the narrow Inception layers
(bf16, through the C ABI): DIN_WGRAD_HALO=0 vs 2, 96 and 12 frames.  Time includes the slab reduce; prints the max-rel difference of dW / db."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L

LAYERS = {  # name: (h, w, cin, cout, k, p)
    "Mixed_5x.branch5x5_2": (87, 157, 48, 64, (5, 5), (2, 2)),
    "Mixed_5x.branch3x3dbl_2": (87, 157, 64, 96, (3, 3), (1, 1)),
    "Mixed_5x.branch3x3dbl_3": (87, 157, 96, 96, (3, 3), (1, 1)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--frames", type=int, nargs="*", default=[96, 12])
    a = ap.parse_args()
    lib = L.load()
    for nb in a.frames:
        for name, (h, w, cin, cout, k, p) in LAYERS.items():
            d = L.ConvDesc()
            d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, h, w, cout
            d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = k[0], k[1], 1, 1, p[0], p[1], 1, 1
            d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin, 0, cout, 0, L.DIN_BF16
            g = torch.Generator(device="cuda").manual_seed(1)
            x = torch.randn(nb, h, w, cin, device="cuda", generator=g).relu().bfloat16()
            gy = torch.randn(nb, h, w, cout, device="cuda", generator=g).bfloat16()
            fl = 2.0 * nb * h * w * cout * cin * k[0] * k[1]
            ref, line = None, f"{name:26s} nb={nb:3d} "
            for mode in ("0", "2"):
                os.environ["DIN_WGRAD_HALO"] = mode
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
                err = ""
                if ref is None:
                    ref = (dw.clone(), db.clone())
                else:
                    err = f" (dW {((dw - ref[0]).abs().max() / ref[0].abs().max()).item():.1e} db {((db - ref[1]).abs().max() / ref[1].abs().max()).item():.1e})"
                line += f"| halo={mode} [code {bm.value}, {bn.value}] {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF{err} "
            print(line, flush=True)


if __name__ == "__main__":
    main()
