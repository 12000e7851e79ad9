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
HBM = None               # list while the HBM-bound launch groups are surveyed: (group, kernel entry point, algorithmic bytes, start event, end event)


def _conv_flops(d) -> float:
    over = getattr(d, "flops_override", None)              # a grouped launch (nhwc.flush_wq) carries the sum over its layers
    return float(over) if over else 2.0 * d.nb * d.oh * d.ow * d.cout * d.cin * d.kh * d.kw


def _esz(dt: int) -> int:
    return 4 if dt == L.DIN_F32 else 2


def _stem_bytes(kind: str, d) -> float:
    """algorithmic HBM bytes of one launch of the halo-tiled stem kernels (input once + output once; the dgrad also reads
    the ReLU mask = the layer's stored input, the weight gradient reads input + output gradient)"""
    e = _esz(d.dtype)
    x = d.nb * d.h * d.w * (3 if d.in_u8 else d.cin * e)               # uint8 frames: 3 bytes per pixel
    y = d.nb * d.oh * d.ow * d.cout * e
    if kind == "fwd":
        return float(x + y)
    if kind == "dgrad":
        return float(y + 2 * x)                                        # gradient in, ReLU mask in, gradient out
    return float(x + y)


def _pool_bytes(name: str, args) -> float:
    d = args[0]._obj
    e = _esz(d.dtype)
    big, small = d.nb * d.h * d.w * d.c, d.nb * d.oh * d.ow * d.c
    if name == "din_maxpool_fwd":                                      # (d, in, out, argmax, stream)
        return float(e * (big + small) + (small if args[3] else 0))
    # din_maxpool_bwd(d, in, argmax, dout, din, relu_mask, accumulate, stream): with the arg-max map the input is not read
    return float(e * (small + big) + (small if args[2] else e * big) + (e * big if args[6] else 0))


def _roi_bytes(name: str, args) -> float:
    if name == "din_roi_align_fwd":                                    # (fm, dt, nb, hf, wf, c, ldf, gh, gw, boxes, ind, m, k, out, ...)
        _fm, dt, nb, hf, wf, c, _ldf, _gh, _gw, _b, _i, m, k = args[:13]
        return float(nb * hf * wf * c * _esz(dt) + m * c * k * k * 4)   # the stored map at most once + the crops
    if name == "din_roi_crop_grad_transpose":                          # (dout, m, c, k, out, stream)
        _d, m, c, k = args[:4]
        return float(2 * m * c * k * k * 4)
    # din_roi_align_bwd_nhwc(dout, dout_c, dout_coff, transposed, nb, hf, wf, c, gh, gw, boxes, ind, m, k, fm_mask, dtype, ldf, gfm, ldg, st)
    nb, hf, wf, c = args[4:8]
    m, k, mask, dt = args[12], args[13], args[14], args[15]
    return float(nb * hf * wf * c * _esz(dt) * (2 if mask else 1) + m * c * k * k * 4)


def _walk_bytes(name: str, args) -> float:
    if name == "din_walk_fwd":                                         # (x, pred, cp, b, t, n, c, kh, kw, ...)
        cp, b, t, n, c, kh, kw = args[2:9]
        return float(4 * b * t * n * (2 * c + cp + 5 * kh * kw))       # x in, z out, predictions in, relation weights + corner indices out
    cp = args[2]
    b, t, n, c = args[5:9]
    return float(4 * b * t * n * (3 * c + 2 * cp))                     # x, gz in; dx out; predictions in, their gradient out


_HBM_CALLS = {"din_maxpool_fwd": ("max-pools", _pool_bytes), "din_maxpool_bwd": ("max-pools", _pool_bytes),
              "din_roi_align_fwd": ("RoIAlign", _roi_bytes), "din_roi_crop_grad_transpose": ("RoIAlign", _roi_bytes),
              "din_roi_align_bwd_nhwc": ("RoIAlign", _roi_bytes),
              "din_walk_fwd": ("DIN walk", _walk_bytes), "din_walk_bwd": ("DIN walk", _walk_bytes)}


class hbm_survey:
    """`with hbm_survey() as rec:` -- every launch of the HBM-bound groups (stem halo kernels through the conv timer, max-pools, RoIAlign,
    DIN walk through wrapped library entry points) is bracketed with HIP events on the launch stream and recorded with its ALGORITHMIC
    bytes.  Measurement side only: the wrappers are removed on exit."""
    def __enter__(self):
        global HBM
        HBM = []
        lib = L.load()
        self.saved = {}
        for name, (group, fbytes) in _HBM_CALLS.items():
            fn = getattr(lib, name)
            self.saved[name] = fn

            def wrapped(*args, _fn=fn, _name=name, _group=group, _fb=fbytes):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
                rc = _fn(*args)
                e1.record(torch.cuda.current_stream())
                vals = [a.value if isinstance(a, (C.c_void_p, C.c_int, C.c_int64)) else a for a in args]
                HBM.append((_group, _name, _fb(_name, vals), e0, e1))
                return rc
            setattr(lib, name, wrapped)
        return self

    def __exit__(self, *exc):
        global HBM
        lib = L.load()
        for name, fn in self.saved.items():
            setattr(lib, name, fn)
        self.records, HBM = HBM, None
        return False

    def summary(self, event_overhead_ms: float = 0.0, peak_gbs: float = 8000.0):
        out = {}
        for group, _name, nbytes, e0, e1 in self.records:
            rec = out.setdefault(group, [0.0, 0.0, 0])
            rec[0] += nbytes
            rec[1] += max(e0.elapsed_time(e1) - event_overhead_ms, 1e-4) * 1e-3
            rec[2] += 1
        return {g: {"bound": "hbm", "achieved": round(v[0] / v[1] / 1e9, 1), "peak": peak_gbs, "unit": "GB/s",
                    "frac": round(v[0] / v[1] / 1e9 / peak_gbs, 4), "launches": v[2], "time_ms": round(v[1] * 1e3, 3),
                    "algorithmic_bytes": round(v[0])} for g, v in out.items()}


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
            if self.name.startswith("1x1multi:"):
                return f"conv_wgrad_1x1_multi_kernel<{d.cin // 8}>"
            if self.name.startswith("group:"):                         # din_conv_wgrad_group: the sixteen-wave pipe instantiation of the items' tile
                return f"conv_wgrad_pipe_group_kernel<{bm.value}, {bn.value - 2000}, {'true' if d.ow >= 32 else 'false'}, 8>"
            if bm.value == 3:
                return f"conv_wgrad_halo_kernel<..., {bn.value}, ...>"
            if bm.value == 0:
                return f"conv_wgrad_small_kernel<..., {bn.value}, ...>"
            if bn.value >= 2000:
                waves = int(L.get_option("DIN_WGRAD_PIPE_WAVES") or "16")      # wave grid 2 x WN (conv_wgrad_pipe.hip launch_wgrad_pipe)
                wn = 8 if waves == 16 else (2 if waves == 4 and bm.value <= 192 else 4)
                return f"conv_wgrad_pipe_kernel<{bm.value}, {bn.value - 2000}, {'true' if d.ow >= 32 else 'false'}, {wn}>"
            if bn.value >= 1000:
                return f"conv_wgrad_ring_kernel<{bm.value}, {bn.value - 1000}>"
            return f"conv_wgrad_bf16_kernel<{bm.value}>" if d.dtype == L.DIN_BF16 else "conv_wgrad_f32_kernel"
        if bm.value == 4 and not (self.kind == "dgrad" and "+" in self.name):
            return f"conv1x1_stream_kernel<{bn.value}>"
        if bm.value == 5:
            # NKS as launch_conv1x1_regw counts it: every source padded to whole 64-channel stages (a multi-source dgrad carries its sources'
            # channel counts as d.src_couts: 192 + 160 + 160 + 192 is 24 k-steps, not the 22 of the summed reduction)
            srcs = getattr(d, "src_couts", None) or ((d.cin if self.kind == "fwd" else d.cout),)
            nks = sum((c + 63) // 64 * 2 for c in srcs)
            if nks in (6, 8, 10, 20, 24):
                return f"conv1x1_regw_kernel<{nks}, ...>"
            bm.value, bn.value = 128, 192                             # (the multi-source launch does not fit the register-resident form: tile kernel)
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
        # <T, BM, BN, WM, WN, KCS, NS, MULTI, FASTK, XSRC, LANEK> as rocprofv3 prints it
        return (f"conv_gather_fast_kernel<{tn}, {BM}, {BN}, {geo}, {'true' if multi else 'false'}, "
                f"{'true' if fl.value & 1 else 'false'}, false, {'true' if fl.value & 4 else 'false'}>")

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
            if HBM is not None and self.variant.startswith(("conv_small_kernel", "conv_wgrad_small_kernel")):
                HBM.append(("stem (Conv2d_1a-2b halo kernels)", self.variant, _stem_bytes(self.kind, self.d), self.e0, self.e1))
        return False




def install() -> None:
    nhwc.LAUNCH_TIMER = LaunchTimer
    nhwc.TIMING_ACTIVE = lambda: PROFILE is not None      # per-launch events only make sense on one stream: bracketed steps run single-stream


def uninstall() -> None:
    nhwc.LAUNCH_TIMER = None
    nhwc.TIMING_ACTIVE = lambda: False
