"""ctypes binding of libdin_hip.so (C ABI declared in include/din_hip.h).

The product path has NO fallback: if the shared library is missing or was built for another
target this module raises, and every op raises on non-GPU tensors.  The library travels in-tree
(built by `__graft_entry__.build()` / `make -C .../csrc`), never from site-packages.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from typing import Dict, List

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIN_LIB_PATH") or os.path.join(_HERE, "libdin_hip.so")     # (DIN_LIB_PATH: experiment builds, tools/ only)
CSRC_DIR = os.path.join(_HERE, "csrc")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "din_hip.h")

DIN_F32, DIN_BF16 = 0, 1
ABI_VERSION = 9
CONV_BIAS, CONV_RELU, CONV_ACCUM, CONV_MASK = 1, 2, 4, 8


class DinError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nb", "h", "w", "cin", "oh", "ow", "cout", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw",
        "ldi", "cioff", "ldo", "cooff", "dtype", "in_u8")]


class PoolDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "nb", "h", "w", "c", "oh", "ow", "k", "stride", "pad", "ldi", "cioff", "ldo", "cooff", "dtype")]


class PackDesc(C.Structure):
    _fields_ = [("w", C.c_uint64), ("scale", C.c_uint64), ("out", C.c_uint64)] + [(n, C.c_int32) for n in (
        "cout", "cin", "kh", "kw", "rows", "rows_pad", "inner", "inner_pad", "kelems", "transposed", "dtype", "reserved")]


class ConvSrc(C.Structure):
    _fields_ = [("dout", C.c_void_p), ("wpk_t", C.c_void_p), ("cout", C.c_int32), ("ldo", C.c_int32), ("cooff", C.c_int32)]


class ConvWSrc(C.Structure):
    _fields_ = [("dout", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("scale", C.c_void_p), ("w", C.c_void_p), ("wdot", C.c_void_p),
                ("cout", C.c_int32), ("ldo", C.c_int32), ("cooff", C.c_int32)]


class ConvWgradItem(C.Structure):                  # din_conv_wgrad_item: the arguments of one din_conv_wgrad call
    _fields_ = [("desc", ConvDesc), ("in_", C.c_void_p), ("dout", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p), ("scale", C.c_void_p),
                ("w", C.c_void_p), ("wdot", C.c_void_p), ("accumulate", C.c_int32), ("reserved", C.c_int32)]


_P, _I, _L, _F, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
_CD, _PD = C.POINTER(ConvDesc), C.POINTER(PoolDesc)

# name -> (restype, argtypes); mirrors include/din_hip.h one to one
SIGNATURES: Dict[str, tuple] = {
    "din_abi_version": (_I, []),
    "din_last_error_string": (C.c_char_p, []),
    "din_set_option": (_I, [C.c_char_p, C.c_char_p]),
    "din_get_option": (_I, [C.c_char_p, C.c_char_p, _I]),
    "din_build_arch": (C.c_char_p, []),
    "din_prep_images_f32": (_I, [_P, _P, _L, _P]),
    "din_prep_images_nhwc": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "din_conv_packed_elems": (_L, [_CD, _I]),
    "din_conv_pack_weights": (_I, [_CD, _P, _P, _P, _I, _P]),
    "din_conv_pack_desc": (_I, [_CD, _P, _P, _P, _I, C.POINTER(PackDesc)]),
    "din_conv_pack_multi": (_I, [_P, _P, _P, _I, _I, _P]),
    "din_conv_kernel_tile": (_I, [_CD, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "din_conv_kernel_variant": (_I, [_CD, _I, C.POINTER(C.c_int32)]),
    "din_conv_workspace_bytes": (_L, [_CD, _I]),
    "din_conv_accepts_u8": (_I, [_P]),
    "din_conv_fwd": (_I, [_CD, _P, _P, _P, _P, _I, _P, _L, _P]),
    "din_conv_fwd2": (_I, [_CD, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "din_conv_dgrad": (_I, [_CD, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P]),
    "din_conv1x1_dgrad_multi": (_I, [_I, C.POINTER(ConvSrc), _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
    "din_conv_dgrad_x_fused": (_I, [_CD]),
    "din_conv_dgrad_x": (_I, [_CD, _P, _P, _P, _P, _I, _I, _I, C.POINTER(ConvSrc), _P, _L, _P]),
    "din_conv1x1_wgrad_multi_workspace": (_L, [_I, C.POINTER(ConvWSrc), _I, _L, _I]),
    "din_conv1x1_wgrad_multi": (_I, [_I, C.POINTER(ConvWSrc), _I, _L, _I, _I, _I, _P, _I, _P, _L, _P]),
    "din_conv_wgrad_group_key": (_I, [_CD]),
    "din_conv_wgrad_group_workspace": (_L, [_I, C.POINTER(ConvWgradItem)]),
    "din_conv_wgrad_group": (_I, [_I, C.POINTER(ConvWgradItem), _P, _L, _P]),
    "din_conv_wgrad": (_I, [_CD, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _P]),
    "din_colsum": (_I, [_P, _I, _L, _I, _I, _I, _P, _P]),
    "din_bn_fold": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _P]),
    "din_bn_fold_bwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _I, _P]),
    "din_bn_fold_multi": (_I, [_P, _P, _I, _I, _F, _P, _P, _P]),
    "din_bn_fold_bwd_multi": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P, _P]),
    "din_bn_stats": (_I, [_P, _I, _L, _I, _I, _I, _P, _P, _P]),
    "din_bn_parts": (_I, [_L]),
    "din_bn_workspace": (_L, [_L, _I]),
    "din_bn_reduce": (_I, [_P, _I, _I, _P]),
    "din_bn_finalize": (_I, [_P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "din_bn_apply": (_I, [_P, _I, _L, _I, _I, _I, _P, _P, _I, _P, _I, _I, _P]),
    "din_bn_bwd_stats": (_I, [_P, _I, _I, _P, _I, _I, _I, _L, _I, _P, _P, _P, _P]),
    "din_bn_bwd_apply": (_I, [_P, _I, _I, _P, _I, _I, _I, _L, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "din_maxpool_fwd": (_I, [_PD, _P, _P, _P, _P]),
    "din_maxpool_bwd": (_I, [_PD, _P, _P, _P, _P, _I, _I, _P]),
    "din_avgpool_fwd": (_I, [_PD, _P, _P, _P, _I, _P]),
    "din_avgpool_bwd": (_I, [_PD, _P, _P, _P, _I, _P]),
    "din_bilinear_fwd": (_I, [_PD, _P, _P, _P]),
    "din_bilinear_bwd": (_I, [_PD, _P, _P, _P, _I, _P]),
    "din_roi_align_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _P, _P]),
    "din_roi_align_bwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "din_roi_crop_grad_transpose": (_I, [_P, _I, _I, _I, _P, _P]),
    "din_roi_align_bwd_nhwc": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _P, _I, _P]),
    "din_grad_cast_mask": (_I, [_P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _I, _P]),
    "din_boxes_frame_index": (_I, [_P, _I, _I, _P]),
    "din_layernorm_fwd": (_I, [_P, _P, _P, _P, _F, _P, _P, _L, _L, _I, _F, _U64, _P, _P]),
    "din_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _F, _U64, _P, _P]),
    "din_counter_add": (_I, [_P, _U64, _P]),
    "din_walk_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "din_walk_bwd": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "din_head_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "din_head_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "din_axpby": (_I, [_P, _P, _P, _F, _F, _L, _P]),
    "din_mask_actors": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "din_scale_by_param": (_I, [_P, _P, _I, _P, _I, _L, _P]),
    "din_dot_accum": (_I, [_P, _P, _P, _I, _L, _P]),
    "din_cast": (_I, [_P, _I, _P, _I, _L, _P]),
    "din_nhwc_to_nchw_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "din_nchw_f32_to_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "din_ctx_scores": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "din_softmax_rows": (_I, [_P, _L, _I, _P]),
    "din_softmax_rows_bwd": (_I, [_P, _P, _L, _I, _P]),
    "din_ctx_apply": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "din_ctx_keys_grad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "din_add_position": (_I, [_P, _I, _P, _P, _L, _L, _P]),
    "din_add_position_bwd": (_I, [_P, _P, _I, _P, _L, _I, _P]),
    "din_act_dropout_fwd": (_I, [_P, _P, _L, _I, _F, _U64, _P, _P]),
    "din_act_dropout_bwd": (_I, [_P, _P, _P, _L, _I, _F, _U64, _P, _P]),
    "din_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    "din_adam_step_multi": (_I, [_P, _P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _I, _F, _P]),
}

_lib = None


def header_symbols() -> List[str]:
    """Every function name include/din_hip.h declares (used by the CPU symbol test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(din_[a-z0-9_]+)\s*\(", text)))


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into the in-tree libdin_hip.so (hipcc cross-compiles w/o a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j", str(os.cpu_count() or 4)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise DinError("building libdin_hip.so failed (see output above)")
    return LIB_PATH


def load():
    """Load the library (once).  Raises DinError loudly -- there is no CPU/eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DinError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(or `make -C {CSRC_DIR}`); the DIN hot path has no fallback implementation")
    # torch first: its wheel bundles its own libamdhip64; if this library pulled in /opt/rocm's copy before torch loaded its own, the
    # process would hold two HIP runtimes and the second one finds no device ("no ROCm-capable device is detected")
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DinError(f"libdin_hip.so does not export {name}; stale build?") from e
        fn.restype, fn.argtypes = res, args
    if lib.din_abi_version() != ABI_VERSION:
        raise DinError(f"libdin_hip.so ABI version {lib.din_abi_version()} != {ABI_VERSION} (stale build?)")
    if lib.din_build_arch() != b"gfx950":
        raise DinError("libdin_hip.so was not built for gfx950")
    _lib = lib
    if os.environ.get("DIN_OPTIONS_FROM_ENV") == "1":
        # tuning tools only (tools/gpu/lease.sh, tools/conv_bench.py A/B runs): forward the DIN_* variables of THIS process once, at load
        # time, as library options.  Nothing else ever carries the environment into the library.
        for k, v in os.environ.items():
            if k.startswith("DIN_") and k not in ("DIN_OPTIONS_FROM_ENV", "DIN_LIB_PATH"):
                lib.din_set_option(k.encode(), v.encode())
    return lib


def set_option(name: str, value) -> None:
    """Select a kernel variant for the tests / tuning tools (include/din_hip.h: din_set_option); value None = back to the shipped choice."""
    check(load().din_set_option(name.encode(), None if value is None else str(value).encode()), f"set_option({name})")


def get_option(name: str):
    buf = C.create_string_buffer(256)
    rc = load().din_get_option(name.encode(), buf, 256)
    if rc < 0:
        check(rc, f"get_option({name})")
    return buf.value.decode() if rc == 1 else None


def host_flag(name: str, default: bool) -> bool:
    """A switch of the PYTHON host layer (graph fusions in nhwc.py, graph variants in backbone/backbone.py), kept in the SAME option table
    as the library's kernel-selection switches (din_set_option / din_get_option): production sets none, tests and tuning tools set them
    through the C ABI, nothing reads the process environment.  "0" = off, anything else = on, unset = `default`."""
    if _lib is None and not os.path.exists(LIB_PATH):
        return default                                     # (the library is missing: every op will say so; graph construction need not)
    v = get_option(name)
    return default if v is None else v != "0"


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().din_last_error_string().decode("utf-8", "replace")
        raise DinError(f"{what or 'libdin_hip'} failed (code {rc}): {msg}")
