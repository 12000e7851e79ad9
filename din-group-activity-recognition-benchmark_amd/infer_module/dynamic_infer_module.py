"""Dynamic Relation / Dynamic Walk inference -- drop-in for the reference's infer_module/dynamic_infer_module.py.

Same classes, constructor signatures, forward contract `(ft [B,T,N,C], ft_infer_MAD [B,T,N,k2,C])` and state_dict
keys (`hidden_weight.weight`, `p_conv.<ratio>.{weight,bias}`, `scale_conv.<ratio>.{weight,bias}`, `beta`,
`DIMlist.<i>.*`, `DPI_1/hier_LN/DPI_2`).  The arithmetic runs on MI355X:
  * p_conv + scale_conv share ONE implicit-GEMM MFMA contraction over the T x N grid (3*k2 output channels);
  * softmax over k2, floor/clamp corners, bilinear coefficients, 4-corner gathers and the weighted aggregation are
    ONE fused LDS-tiled kernel (din_walk_fwd), its backward another (din_walk_bwd);
  * hidden_weight is an MFMA contraction.
`ft_infer_MAD` is only materialised when `return_mad=True` (both reference callers discard it: infer_model.py:199).

Reference behaviours kept on purpose: Q2 channel order, Q3 clamp double-count, Q4 detached floor,
Q5 zero init, Q8/Q9 sub-gradients.  `dynamic_sampling=False` (plain_infer_ratio, :154-181) and `parallel_inference=True`
(parallel_infer, :285-341) run their per-ratio arithmetic through the same walk kernel (lattice-gather mode / person_mat_shape
clamps); the reference's forward then dies on an unbound `ft_infer_MAD` (Q1, :151) -- here the MAD slot of the returned tuple is None.
Hierarchical uses the intended semantics.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class Dynamic_Person_Inference(nn.Module):
    def __init__(self, in_dim, person_mat_shape, stride=1, kernel_size=(3, 3), dynamic_sampling=False,
                 sampling_ratio=[1], group=1, scale_factor=False, beta_factor=False, parallel_inference=False,
                 cfg=None, return_mad=False):
        super().__init__()
        if stride != 1 or group != 1:
            raise NotImplementedError("only stride=1, group=1 is ever used by the reference (config.py:84,88)")
        if parallel_inference and not (dynamic_sampling and scale_factor):
            raise AssertionError("parallel_inference needs dynamic_sampling and scale_factor (dynamic_infer_module.py:130)")
        self.dynamic_sampling, self.parallel_inference = bool(dynamic_sampling), bool(parallel_inference)
        self.T, self.N = person_mat_shape
        self.kernel_size = tuple(kernel_size)
        self.sampling_ratio = list(sampling_ratio)
        self.scale_factor, self.beta_factor = scale_factor, beta_factor
        self.return_mad = return_mad
        self.cfg = cfg
        kh, kw = self.kernel_size
        self.hidden_weight = nn.Linear(in_dim, in_dim, bias=False)
        nn.init.kaiming_normal_(self.hidden_weight.weight)
        if beta_factor:
            self.beta = nn.Parameter(torch.ones(len(self.sampling_ratio)))
        if dynamic_sampling:                                   # (:47-48: the offset predictors exist only with dynamic sampling)
            self.p_conv = nn.ModuleDict()
        if scale_factor:
            self.scale_conv = nn.ModuleDict()
        for r in self.sampling_ratio:
            pad = ((kh - 1) // 2 * r, (kw - 1) // 2 * r)
            if dynamic_sampling:
                pc = nn.Conv2d(in_dim, 2 * kh * kw, self.kernel_size, dilation=r, padding=pad)
                nn.init.zeros_(pc.weight), nn.init.zeros_(pc.bias)
                self.p_conv[str(r)] = pc
            if scale_factor:
                sc = nn.Conv2d(in_dim, kh * kw, self.kernel_size, dilation=r, padding=pad)
                nn.init.zeros_(sc.weight), nn.init.zeros_(sc.bias)
                self.scale_conv[str(r)] = sc

    def _ratio(self, x, r, n_per_clip=None):
        kh, kw = self.kernel_size
        k2 = kh * kw
        if not self.dynamic_sampling:
            # plain_infer_ratio (:154-181): features AT the lattice points, weighted by the relation softmax (or averaged)
            if self.scale_factor:
                sc = self.scale_conv[str(r)]
                w = torch.cat([sc.weight.new_zeros((2 * k2,) + tuple(sc.weight.shape[1:])), sc.weight], 0)     # offset channels unused
                b = torch.cat([sc.bias.new_zeros(2 * k2), sc.bias], 0)
                pred = ops.GridConvFunction.apply(x, w, b, r)
            else:
                pred = x.new_zeros(x.shape[:3] + (2 * k2,))
            z, a, idx, mad = ops.DynamicWalkFunction.apply(x, pred, kh, kw, r, self.scale_factor, False, n_per_clip, True, None)
            return z, None, a, idx
        pc = self.p_conv[str(r)]
        if self.scale_factor:
            sc = self.scale_conv[str(r)]
            w = torch.cat([pc.weight, sc.weight], 0)
            b = torch.cat([pc.bias, sc.bias], 0)
        else:
            w, b = pc.weight, pc.bias
        pred = ops.GridConvFunction.apply(x, w, b, r)                       # [B,T,N,pad4(3*k2)]
        if self.parallel_inference:
            # parallel_infer (:285-341): relation-weighted lattice gather + MEAN over k2 of the dynamic walk, the latter clamped with
            # person_mat_shape: indices to (T + 2r - 1, N + 2r - 1), positions to (T + 2r, N + 2r)   (:307-317)
            if n_per_clip is not None:
                raise NotImplementedError("parallel_inference with per-clip actor counts")
            zs, a, idx, _ = ops.DynamicWalkFunction.apply(x, pred, kh, kw, r, True, False, None, True, None)
            clamp = (self.T + 2 * r - 1, self.N + 2 * r - 1, self.T + 2 * r, self.N + 2 * r)
            zw, _, idx, _ = ops.DynamicWalkFunction.apply(x, pred, kh, kw, r, False, False, None, False, clamp)
            return ops.AxpbyFunction.apply(zs, zw, 1.0, 1.0), None, a, idx
        z, a, idx, mad = ops.DynamicWalkFunction.apply(x, pred, kh, kw, r, self.scale_factor, self.return_mad, n_per_clip)
        return z, mad, a, idx

    def forward(self, person_features, n_per_clip=None):
        """n_per_clip (optional int32 [B] on the device): clip b is the T x n_per_clip[b] grid in the first columns of its slab and
        `person_features` is zero beyond it (Dynamic_collective; the reference calls the module once per clip, infer_model.py:1286-1293)"""
        x = person_features.contiguous()
        agg, mad = None, None
        nr = len(self.sampling_ratio)
        for i, r in enumerate(self.sampling_ratio):
            z, mad, _a, _idx = self._ratio(x, r, n_per_clip)
            if self.beta_factor:
                z = ops.ScaleByParamFunction.apply(z, self.beta, i)
                agg = z if agg is None else ops.AxpbyFunction.apply(agg, z, 1.0, 1.0)
            else:
                agg = z if agg is None else ops.AxpbyFunction.apply(agg, z, 1.0, 1.0)
        if not self.beta_factor and nr > 1:
            agg = ops.AxpbyFunction.apply(agg, agg, 1.0 / nr, 0.0)
        out = ops.linear(agg, self.hidden_weight.weight, None)
        return out, (mad if self.return_mad else None)


class Multi_Dynamic_Inference(nn.Module):
    def __init__(self, in_dim, person_mat_shape, stride=1, kernel_size=[(3, 3)], dynamic_sampling=False,
                 sampling_ratio=[1], group=1, scale_factor=False, beta_factor=False, parallel_inference=False,
                 num_DIM=1, cfg=None, return_mad=False):
        super().__init__()
        self.DIMlist = nn.ModuleList([
            Dynamic_Person_Inference(in_dim, person_mat_shape, stride, kernel_size[i], dynamic_sampling, sampling_ratio,
                                     group, scale_factor, beta_factor, parallel_inference, cfg, return_mad)
            for i in range(num_DIM)])

    def forward(self, person_features):
        out, mad = None, None
        for m in self.DIMlist:
            o, mad = m(person_features)
            out = o if out is None else ops.AxpbyFunction.apply(out, o, 1.0, 1.0)
        return out, mad


class Hierarchical_Dynamic_Inference(nn.Module):
    """DPI_1 -> LayerNorm(person_mat_shape + (C,)) -> ReLU -> dropout -> DPI_2.  The reference crashes here (tuple fed to LayerNorm,
    :492-493); this is the intended dataflow.  The reference's dropout is `F.dropout(x)` with the functional defaults (:495): p = 0.5 and
    training=True, i.e. ALWAYS on, also under model.eval().  That is kept as the default; `hier_dropout_p` (ctor argument, or
    `cfg.hier_dropout_p` when a cfg is given) makes it explicit -- 0.0 gives the deterministic module the golden vectors were taken with."""

    def __init__(self, in_dim, person_mat_shape, stride=1, kernel_size=[(3, 3)], dynamic_sampling=False,
                 sampling_ratio=[1], group=1, scale_factor=False, beta_factor=False, parallel_inference=False, cfg=None,
                 hier_dropout_p=None):
        super().__init__()
        assert len(kernel_size) == 2
        mk = lambda k: Dynamic_Person_Inference(in_dim, person_mat_shape, stride, k, dynamic_sampling, sampling_ratio,
                                                group, scale_factor, beta_factor, parallel_inference, cfg)
        self.DPI_1 = mk(kernel_size[0])
        self.hier_LN = nn.LayerNorm(tuple(person_mat_shape) + (in_dim,))
        self.DPI_2 = mk(kernel_size[1])
        if hier_dropout_p is None:
            hier_dropout_p = getattr(cfg, "hier_dropout_p", 0.5) if cfg is not None else 0.5
        self.hier_dropout_p = float(hier_dropout_p)
        self._step = 0              # dropout-mask counter (saved / restored by train_net with the checkpoint, see ops.mask_seed)

    def forward(self, person_features):
        h, _ = self.DPI_1(person_features)
        self._step += 1
        h = ops.layer_norm(h, self.hier_LN.weight, self.hier_LN.bias, relu=True, drop_p=self.hier_dropout_p,
                           seed=ops.mask_seed(0x9E3779B1, self._step))
        return self.DPI_2(h)
