"""Context_PositionEmbeddingSine -- drop-in for the class Dynamic_TCE_volleyball uses from the reference's
infer_module/positional_encoding.py:50-92.

The embedding depends only on (OH, OW): it is built once per map size on the host (a [OH,OW,2*num_pos_feats] table, pixel-major
like the backbone buffers) and added to the context map by one HBM-bound kernel (din_add_position), whose backward hands the
gradient back to the backbone graph in the backbone's storage type.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops


class Context_PositionEmbeddingSine(nn.Module):
    def __init__(self, context_downscale_ratio, num_pos_feats, temperature=10000, normalize=False, scale=None):
        super().__init__()
        self.context_downscale_ratio = context_downscale_ratio
        self.num_pos_feats = int(num_pos_feats)            # the reference passes 512 / 2 = 256.0 (infer_model.py:293)
        self.temperature = temperature
        self.normalize = normalize
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.scale = 2 * math.pi if scale is None else scale
        self._tables = {}

    def table(self, oh: int, ow: int, device) -> torch.Tensor:
        """pos [OH, OW, 2C] fp32: channels [0, C) encode y, [C, 2C) encode x; even channels sin, odd channels cos (:79-89)"""
        key = (oh, ow, str(device))
        if key not in self._tables:
            c = self.num_pos_feats
            y = torch.arange(1, oh + 1, dtype=torch.float32) * self.context_downscale_ratio       # cumsum of ones (:73-74)
            x = torch.arange(1, ow + 1, dtype=torch.float32) * self.context_downscale_ratio
            if self.normalize:
                eps = 1e-6
                y = y / (y[-1:] + eps) * self.scale
                x = x / (x[-1:] + eps) * self.scale
            i = torch.arange(c, dtype=torch.float32)
            dim_t = self.temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / c)
            py, px = y[:, None] / dim_t, x[:, None] / dim_t

            def interleave(a):
                return torch.stack((a[:, 0::2].sin(), a[:, 1::2].cos()), dim=2).flatten(1)

            py, px = interleave(py), interleave(px)
            pos = torch.cat((py[:, None, :].expand(oh, ow, c), px[None, :, :].expand(oh, ow, c)), dim=2)
            self._tables[key] = pos.contiguous().to(device)
        return self._tables[key]

    def forward(self, context: torch.Tensor, nhwc: bool = False, relu_masked: bool = False) -> torch.Tensor:
        """context NCHW fp32 (reference API) -> NCHW; or, nhwc=True, a backbone buffer [BT,OH,OW,C] -> fp32 [BT,OH,OW,C]"""
        if not nhwc:
            bt, c, oh, ow = context.shape
            x = context.permute(0, 2, 3, 1).contiguous()
            return ops.AddPositionFunction.apply(x, self.table(oh, ow, context.device), False).permute(0, 3, 1, 2)
        bt, oh, ow, c = context.shape
        assert c == 2 * self.num_pos_feats, f"context has {c} channels, the embedding {2 * self.num_pos_feats}"
        return ops.AddPositionFunction.apply(context, self.table(oh, ow, context.device), relu_masked)
