#!/usr/bin/env python3
"""Does running the HBM-bound stem in FRAME CHUNKS keep producer -> consumer tensors inside the 256 MB last-level cache?
Conv2d_2a (32 -> 32, 3x3) followed by Conv2d_2b (32 -> 64, 3x3 pad 1) on 96 frames of 359 x 639 (reference backbone/backbone.py:45-46), forward
only: all 96 frames per launch (what the step does) against chunks of C frames (2a on chunk i, then 2b on chunk i: 2b reads what 2a just wrote).
Same kernels, same bytes; only the order changes.   usage: python tools/chunk_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from din_amd import _lib as L

lib = L.load()
NB, H, W = 96, 359, 639
bf = torch.bfloat16


def desc(nb, h, w, cin, cout, pad):
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, h + 2 * pad - 2, w + 2 * pad - 2, cout
    d.kh = d.kw = 3
    d.sh = d.sw = d.dh = d.dw = 1
    d.ph = d.pw = pad
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin, 0, cout, 0, L.DIN_BF16
    return d


def packed(d, w):
    wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), dtype=bf, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), w.data_ptr(), None, wpk.data_ptr(), 0, None))
    return wpk


torch.manual_seed(0)
x = torch.randn(NB, H, W, 32, device="cuda").to(bf)
w2a, w2b = torch.randn(32, 32, 3, 3, device="cuda") * 0.1, torch.randn(64, 32, 3, 3, device="cuda") * 0.1
b2a, b2b = torch.zeros(32, device="cuda"), torch.zeros(64, device="cuda")
y2a = torch.empty(NB, H - 2, W - 2, 32, device="cuda", dtype=bf)
y2b = torch.empty(NB, H - 2, W - 2, 64, device="cuda", dtype=bf)
pk2a, pk2b = packed(desc(NB, H, W, 32, 32, 0), w2a), packed(desc(NB, H - 2, W - 2, 32, 64, 1), w2b)


def run(chunk):
    da, db = desc(chunk, H, W, 32, 32, 0), desc(chunk, H - 2, W - 2, 32, 64, 1)
    for f0 in range(0, NB, chunk):
        L.check(lib.din_conv_fwd(C.byref(da), x[f0].data_ptr(), pk2a.data_ptr(), b2a.data_ptr(), y2a[f0].data_ptr(), 3, None, 0, None))
        L.check(lib.din_conv_fwd(C.byref(db), y2a[f0].data_ptr(), pk2b.data_ptr(), b2b.data_ptr(), y2b[f0].data_ptr(), 3, None, 0, None))


ref = None
for chunk in (96, 48, 24, 12, 8, 6, 4, 2):
    for _ in range(2):
        run(chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(chunk)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    chk = float(y2b.float().abs().sum())
    ref = chk if ref is None else ref
    gb = NB * ((H * W * 32 + 2 * (H - 2) * (W - 2) * 32 + (H - 2) * (W - 2) * 64) * 2) / 1e9
    print(f"chunk {chunk:3d} frames: {ms * 1e3:8.1f} us for 2a + 2b over 96 frames  ({gb / ms:6.2f} TB/s of algorithmic bytes; 2a output per chunk "
          f"{chunk * (H - 2) * (W - 2) * 32 * 2 / 1e6:6.1f} MB)  checksum {'same' if chk == ref else 'DIFFERENT'}", flush=True)
