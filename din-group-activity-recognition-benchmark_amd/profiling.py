"""Per-launch timing of the conv launches for bench.py (measurement side, not on the product path).

`install()` sets nhwc.LAUNCH_TIMER; while `PROFILE` is a list every conv launch appends
(kind, kernel name as rocprofv3 prints it, algorithmic FLOPs, dtype, start event, end event, layer name), the events recorded on the
stream the kernel is launched on.  With `PROFILE_ONLY` set, only launches of that kernel are bracketed, so a timed step that needs the
dominant kernel's launch times pays for ~20 event pairs instead of ~190."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import nhwc

PROFILE = None
PROFILE_ONLY = None      # str: while PROFILE is a list, bracket only the launches of this kernel (as named below) with events


def _conv_flops(d) -> float:
    return 2.0 * d.nb * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw


class LaunchTimer:
    """Brackets one conv launch with HIP events when profiling.PROFILE is a list; names the kernel the launch resolves to exactly as rocprofv3
    prints it (din_conv_kernel_tile / din_conv_kernel_variant).  With PROFILE_ONLY set, launches of other kernels are left alone, so a
    timed step that only needs the dominant kernel's launch times pays for ~20 event pairs instead of ~190."""
    def __init__(self, kind, d, name=""):
        self.kind, self.d, self.name = kind, d, name
        self.rec = False

    def _variant(self) -> str:
        d = self.d
        bm, bn = C.c_int32(0), C.c_int32(0)
        which = {"fwd": 0, "dgrad": 1, "wgrad": 2}[self.kind]
        L.load().din_conv_kernel_tile(C.byref(d), which, C.byref(bm), C.byref(bn))
        tn = "unsigned short" if d.dtype == L.DIN_BF16 else "float"
        if self.kind == "wgrad":
            if bm.value == 0:
                return f"conv_wgrad_small_kernel<..., {bn.value}, ...>"
            if bn.value >= 2000:
                waves = int(os.environ.get("DIN_WGRAD_PIPE_WAVES", "16"))      # wave grid 2 x WN (conv_wgrad_pipe.hip launch_wgrad_pipe)
                wn = 8 if waves == 16 else (2 if waves == 4 and bm.value <= 192 else 4)
                return f"conv_wgrad_pipe_kernel<{bm.value}, {bn.value - 2000}, {'true' if d.ow >= 32 else 'false'}, {wn}>"
            if bn.value >= 1000:
                return f"conv_wgrad_ring_kernel<{bm.value}, {bn.value - 1000}>"
            return f"conv_wgrad_bf16_kernel<{bm.value}>" if d.dtype == L.DIN_BF16 else "conv_wgrad_f32_kernel"
        if bm.value == 0:
            return f"conv_small_kernel<..., {bn.value}, ...>"
        if bm.value == 1:
            return f"conv_halo_kernel<{bn.value}, ...>"
        if bm.value == 2 and not (self.kind == "dgrad" and "+" in self.name):
            return f"conv_gather_pipe_kernel<{bn.value}>"
        # the exact instantiation: <T, BM, BN, WM, WN, KCS, NS, MULTI, FASTK>
        fl = C.c_int32(0)
        multi = self.kind == "dgrad" and "+" in self.name
        if not multi:
            L.load().din_conv_kernel_variant(C.byref(d), which, C.byref(fl))
        BM, BN = (128 if bm.value == 2 else bm.value), bn.value      # (multi-source launches stay on the 128-pixel kernel)
        if BM == 256:
            geo = "4, 1, 4, 4" if BN == 64 else ("2, 2, 8, 2" if BN in (96, 160) else "4, 2, 8, 2")
        elif multi:
            geo = "4, 2, 8, 2" if (BN % 64 == 0 and d.dtype == L.DIN_BF16) else "2, 2, 8, 2"
        else:
            geo = "4, 2, 8, 2" if fl.value & 2 else "2, 2, 8, 2"     # (bit 1: the 8-wave instantiation, incl. 128 x 96 for strided dgrads)
        return (f"conv_gather_fast_kernel<{tn}, {BM}, {BN}, {geo}, {'true' if multi else 'false'}, "
                f"{'true' if fl.value & 1 else 'false'}>")

    def __enter__(self):
        if PROFILE is not None:
            self.variant = self._variant()
            if PROFILE_ONLY is None or self.variant == PROFILE_ONLY:
                self.rec = True
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.rec:
            self.e1.record(torch.cuda.current_stream())
            PROFILE.append((self.kind, self.variant, _conv_flops(self.d), int(self.d.dtype), self.e0, self.e1, self.name))
        return False




def install() -> None:
    nhwc.LAUNCH_TIMER = LaunchTimer


def uninstall() -> None:
    nhwc.LAUNCH_TIMER = None
