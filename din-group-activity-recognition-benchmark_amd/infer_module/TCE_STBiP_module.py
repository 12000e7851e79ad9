"""Embedded-feature context encoding (TCE) -- drop-in for the two classes Dynamic_TCE_volleyball uses from the reference's
infer_module/TCE_STBiP_module.py: EmbfeatureContextEncodingTransformer (:224-286) and MultiHeadLayerEmbfeatureContextEncoding (:289-313).

Same constructor signatures and state_dict keys (`CET.<j>.downsample2`, `.emb_roi`, `.layernorm1`, `.FFN.0`, `.FFN.3`, `.layernorm2`;
layers above the first: `.downsample`).  The heads of one layer run together: ONE 1x1 contraction 512 -> heads*128 over the context pixels
(keys = values), ONE Linear NFB -> heads*128 over the boxes (queries), and the attention of all heads in one launch set
(ops.ContextAttentionFunction: scores, row softmax, weighted sum; HBM-bound streaming of the keys).  LayerNorm / FFN run per head
on the existing fused kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class EmbfeatureContextEncodingTransformer(nn.Module):
    """one attention head: parameter container + the per-head tail (LayerNorm(ctx + q) -> + FFN -> LayerNorm)"""

    def __init__(self, num_features_context, NFB, K, N, layer_id, num_heads_per_layer, context_dropout_ratio=0.1):
        super().__init__()
        self.num_features_context = num_features_context
        if layer_id == 1:
            self.downsample2 = nn.Conv2d(512, num_features_context, kernel_size=1, stride=1)
            self.emb_roi = nn.Linear(NFB, num_features_context, bias=True)
        elif layer_id > 1:
            self.downsample = nn.Conv2d(512, num_features_context, kernel_size=1, stride=1)
            self.emb_roi = nn.Linear(num_features_context * num_heads_per_layer, num_features_context, bias=True)
            nn.init.kaiming_normal_(self.downsample.weight)
        self.N, self.K = N, K
        self.dropout = nn.Dropout(context_dropout_ratio)                 # holders of p; the masks are made in the kernels
        self.layernorm1 = nn.LayerNorm(num_features_context)
        self.FFN = nn.Sequential(
            nn.Linear(num_features_context, num_features_context, bias=True),
            nn.ReLU(inplace=True),
            nn.Dropout(context_dropout_ratio),
            nn.Linear(num_features_context, num_features_context, bias=True))
        self.layernorm2 = nn.LayerNorm(num_features_context)
        self.att_map = None

    @property
    def key_conv(self):
        return self.downsample2 if hasattr(self, "downsample2") else self.downsample

    def tail(self, ctx: torch.Tensor, emb: torch.Tensor, seeds) -> torch.Tensor:
        """ctx, emb [rows, 128] -> [rows, 128] (:277-282)"""
        p = self.dropout.p if self.training else 0.0
        if p > 0.0:
            ctx = ops.ActDropoutFunction.apply(ctx, False, p, seeds[0])
        x = ops.layer_norm(ctx, self.layernorm1.weight, self.layernorm1.bias, res=emb)
        f = ops.linear(x, self.FFN[0].weight, self.FFN[0].bias)
        f = ops.ActDropoutFunction.apply(f, True, p, seeds[1])
        f = ops.linear(f, self.FFN[3].weight, self.FFN[3].bias)
        return ops.layer_norm(f, self.layernorm2.weight, self.layernorm2.bias, res=x)


class MultiHeadLayerEmbfeatureContextEncoding(nn.Module):
    def __init__(self, num_heads_per_layer, num_layers, num_features_context, NFB, K, N, context_dropout_ratio=0.1):
        super().__init__()
        self.CET = nn.ModuleList()
        for i in range(num_layers):
            for _j in range(num_heads_per_layer):
                self.CET.append(EmbfeatureContextEncodingTransformer(num_features_context, NFB, K, N, i + 1, num_heads_per_layer,
                                                                     context_dropout_ratio))
        self.num_layers = num_layers
        self.num_heads_per_layer = num_heads_per_layer
        self.num_features_context = num_features_context
        self._step = 0
        self.seed_base = 0

    def forward(self, roi_feature: torch.Tensor, image_feature: torch.Tensor, nhwc: bool = False) -> torch.Tensor:
        """roi_feature [BT*N, NFB]; image_feature [BT, 512, OH, OW] (reference layout) or, nhwc=True, fp32 [BT, OH, OW, 512].
        -> [BT*N, heads * 128]"""
        if not nhwc:
            image_feature = image_feature.permute(0, 2, 3, 1).contiguous()
        image_feature = image_feature.float()
        bt, oh, ow, cin = image_feature.shape
        h, c = self.num_heads_per_layer, self.num_features_context
        n = roi_feature.shape[0] // bt
        self._step += 1
        pix = image_feature.reshape(bt * oh * ow, cin)
        for i in range(self.num_layers):
            heads = [self.CET[i * h + j] for j in range(h)]
            # keys = values of every head: one contraction over the context pixels (:265-270)
            wk = torch.cat([m.key_conv.weight.reshape(c, cin) for m in heads], 0)
            bk = torch.cat([m.key_conv.bias for m in heads], 0)
            kf = ops.linear(pix, wk, bk).reshape(bt, oh * ow, h * c)
            # queries of every head (:266 / :268)
            wq = torch.cat([m.emb_roi.weight for m in heads], 0)
            bq = torch.cat([m.emb_roi.bias for m in heads], 0)
            q = ops.linear(roi_feature, wq, bq)                                           # [BT*N, h*c]
            ctx, att = ops.ContextAttentionFunction.apply(q.reshape(bt, n, h * c), kf, h)  # :271-277
            ctx = ctx.reshape(bt * n, h * c)
            outs = []
            for j, m in enumerate(heads):
                m.att_map = att[:, j]                                                     # [BT, N, OH*OW] (:274)
                seeds = [ops.mask_seed(self.seed_base + 7919 * (2 * (i * h + j) + k + 1), self._step) for k in range(2)]
                outs.append(m.tail(ctx[:, j * c:(j + 1) * c], q[:, j * c:(j + 1) * c], seeds))
            roi_feature = torch.cat(outs, dim=1)                                          # :310
        return roi_feature
