#!/usr/bin/env python3
"""din_colsum (bias / BN-shift gradient of the commuted branch_pool convs) on the default workload's seven shapes: time and HBM rate."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L

def main():
    lib = L.load()
    iters = int(os.environ.get("DIN_PROBE_ITERS", "2000"))
    tot = 0.0
    for (m, c, ld, coff) in ((96 * 87 * 157, 32, 256, 224), (96 * 87 * 157, 64, 288, 224), (96 * 87 * 157, 64, 288, 224),
                             (96 * 43 * 78, 192, 768, 576), (96 * 43 * 78, 192, 768, 576), (96 * 43 * 78, 192, 768, 576), (96 * 43 * 78, 192, 768, 576)):
        gs = [torch.randn(m, ld, device="cuda").to(torch.bfloat16) for _ in range(3)]     # rotated: the slice of ONE tensor would sit in the 256 MB cache
        g = gs[0]
        out = torch.empty(c, device="cuda")
        state = [0]
        def run():
            state[0] = (state[0] + 1) % 3
            L.check(lib.din_colsum(gs[state[0]].data_ptr(), L.DIN_BF16, m, c, ld, coff, out.data_ptr(), None))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        ref = gs[state[0]][:, coff:coff + c].float().sum(0)
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        tot += ms
        print(f"colsum M={m} c={c} ld={ld}: {ms * 1e3:7.1f} us  {m * c * 2 / ms / 1e9:5.2f} TB/s  rel err {err:.1e}", flush=True)
    print(f"sum of the seven: {tot * 1e3:.1f} us")

if __name__ == "__main__":
    main()
