import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
import os as _os
_os.environ.setdefault("DIN_OPTIONS_FROM_ENV", "1")   # tuning tool: DIN_* variables of this process become library options (din_set_option) at load
from din_amd import _lib as L
from tools.pool_bench import bench
lib = L.load(); nb = 96; bf = torch.bfloat16
for (c, h, w, ldo, coff) in ((192, 43, 78, 768, 576), (192, 43, 78, 192, 0), (192, 43, 78, 768, 0), (64, 87, 157, 288, 224), (64, 87, 157, 64, 0)):
    p = L.PoolDesc()
    p.nb, p.h, p.w, p.c, p.oh, p.ow = nb, h, w, c, h, w
    p.k, p.stride, p.pad, p.ldi, p.cioff, p.ldo, p.cooff, p.dtype = 3, 1, 1, c, 0, ldo, coff, L.DIN_BF16
    xi = torch.randn(nb, h, w, c, device="cuda").to(bf)
    yo = torch.empty(nb, h, w, ldo, device="cuda", dtype=bf)
    bias = torch.randn(c, device="cuda")
    for fl, b in ((L.CONV_BIAS | L.CONV_RELU, bias.data_ptr()), (0, None)):
        ms = bench(lambda: L.check(lib.din_avgpool_fwd(C.byref(p), xi.data_ptr(), yo.data_ptr(), b, fl, None)))
        print(f"fwd {c}ch {h}x{w} ldo {ldo} off {coff} flags {fl}: {ms*1e3:7.1f} us")
