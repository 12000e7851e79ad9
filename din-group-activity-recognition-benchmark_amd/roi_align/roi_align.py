"""RoIAlign(crop_h, crop_w)(featuremap, boxes, box_ind) -- drop-in for the third-party `roi_align.roi_align.RoIAlign`
the reference imports at infer_model.py:3 (longcw/RoIAlign.pytorch: TF crop_and_resize with transform_fpcoor=True).

Accepts either the reference's NCHW fp32 feature map (API parity; converted once to pixel-major) or a pixel-major NHWC
buffer produced by din_amd backbones (`nhwc=True`, the fast path).  Output: fp32 [M, C, crop_h, crop_w].
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib as L
from ..nhwc import _ptr, _stream
from ..ops import RoIAlignFunction, RoIAlignMultiScaleFunction


class _NCHWToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = L.load()
        x = x.float().contiguous()
        nb, c, h, w = x.shape
        out = torch.empty((nb, h, w, c), dtype=torch.float32, device=x.device)
        L.check(lib.din_nchw_f32_to_nhwc(_ptr(x), nb, h, w, c, _ptr(out), L.DIN_F32, c, 0, _stream()), "nchw_to_nhwc")
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        g = g.contiguous()
        nb, h, w, c = g.shape
        out = torch.empty((nb, c, h, w), dtype=torch.float32, device=g.device)
        L.check(lib.din_nhwc_to_nchw_f32(_ptr(g), L.DIN_F32, nb, h, w, c, c, 0, _ptr(out), _stream()), "nhwc_to_nchw")
        return out


class RoIAlign(nn.Module):
    def __init__(self, crop_height, crop_width, extrapolation_value=0, transform_fpcoor=True):
        super().__init__()
        if crop_height != crop_width:
            raise NotImplementedError("the DIN path only uses square crops (config.py crop_size = 5, 5)")
        if extrapolation_value != 0 or not transform_fpcoor:
            raise NotImplementedError("only extrapolation_value=0, transform_fpcoor=True (the reference's defaults)")
        self.crop_height, self.crop_width = crop_height, crop_width

    def forward(self, featuremap, boxes, box_ind, nhwc: bool = False, channels: int = None, relu_masked: bool = False):
        if nhwc:
            fm, c = featuremap, channels if channels is not None else featuremap.shape[-1]
        else:
            fm, c = _NCHWToNHWC.apply(featuremap), featuremap.shape[1]
            relu_masked = False
        return RoIAlignFunction.apply(fm, boxes, box_ind, self.crop_height, c, relu_masked)

    def forward_multiscale(self, maps, boxes, box_ind, grid):
        """maps: [(NHWC buffer, channels, relu_masked), ...] at their own resolutions; boxes in `grid` = (OH, OW) pixels.  Equals
        forward(torch.cat([F.interpolate(m, grid, mode='bilinear', align_corners=True) for m in maps], 1), ...) -- the multi-scale fuse of
        infer_model.py:165-180 -- without materialising the resized / concatenated map."""
        fms, chans, masked = zip(*maps)
        return RoIAlignMultiScaleFunction.apply(boxes, box_ind, self.crop_height, tuple(grid), tuple(chans), tuple(masked), *fms)
