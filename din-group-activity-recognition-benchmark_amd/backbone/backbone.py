"""MyVGG16 / MyInception_v3 -- drop-in for the reference's backbone/backbone.py:10-99 on MI355X.

Same class names, constructor arguments, `forward(x) -> list of NCHW feature maps` contract and state_dict keys
(`features.<i>.{weight,bias}` / torchvision Inception naming), but the conv stacks execute as one NHWC graph of
hand-written gfx950 kernels (din_amd.nhwc).  `forward_nhwc(images)` is the fast path used by Dynamic_volleyball:
raw 0..255 images in (uint8 welcome), prep_images fused into the loader, pixel-major buffers out, no layout copies.

Weights: there is no network here, so `pretrained=True` only records the request; real torchvision / stage-1
weights drop in through load_state_dict (key names are identical).
"""
from __future__ import annotations

import os

import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from .. import _lib as L
from ..nhwc import Graph, GraphBuilder, NHWCGraphFunction, View
from ..ops import NHWCToNCHWFunction

VGG16_TABLE = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


def _dt(compute_dtype: str) -> int:
    if compute_dtype in ("fp32", "float32", "f32"):
        return L.DIN_F32
    if compute_dtype in ("bf16", "bfloat16"):
        return L.DIN_BF16
    raise ValueError(f"compute_dtype must be 'fp32' or 'bf16', got {compute_dtype!r}")


def _cpad_image(dt: int) -> int:
    return 4 if dt == L.DIN_F32 else 8


class _GraphBackbone(nn.Module):
    """Common machinery: lazily builds (and caches) the NHWC graph for an input size and runs it."""

    def __init__(self, compute_dtype: str = "fp32"):
        super().__init__()
        self.compute_dtype = compute_dtype
        self._graphs: Dict[Tuple[int, int, int], Graph] = {}

    # subclasses: build_graph(h, w, dt) -> Graph ; named params must match graph.param_names() with prefix removed
    def build_graph(self, h: int, w: int, dt: int) -> Graph:
        raise NotImplementedError

    def graph_for(self, h: int, w: int) -> Tuple[Graph, int]:
        dt = _dt(self.compute_dtype)
        key = (h, w, dt)
        if key not in self._graphs:
            self._graphs[key] = self.build_graph(h, w, dt)
        return self._graphs[key], dt

    def _ordered_params(self, graph: Graph) -> List[torch.Tensor]:
        # cached per graph: walking named_parameters() / named_buffers() of ~300 modules is milliseconds of host time per step -- as much as the
        # whole forward dispatch on the 4-clip step.  `_apply` (.to / .cuda / .float: buffers are REPLACED there) drops the cache; any other
        # rebinding (load_state_dict(assign=True), module.weight = nn.Parameter(...), pruning re-registration, a direct write to
        # module._parameters[...]) is caught by the identity check below: every cached tensor must still be the object its
        # owning module holds under that name (~300 dict lookups, ~25 us).  Nothing is hooked process-wide any more.
        cache = self.__dict__.setdefault("_ordered_cache", {})
        hit = cache.get(id(graph))
        if hit is not None and all(owner.get(key) is t for (owner, key), t in zip(hit[0], hit[1])):
            return hit[1]
        slots, tensors = [], []
        for name in graph.param_names():
            mod_path, _, attr = name.rpartition(".")
            mod = self.get_submodule(mod_path) if mod_path else self
            owner = mod._parameters if attr in mod._parameters else mod._buffers
            slots.append((owner, attr))
            tensors.append(owner[attr])
        cache[id(graph)] = (slots, tensors)
        return tensors

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_ordered_cache", None)
        self.__dict__.pop("_bn_modules", None)
        return super()._apply(fn, *args, **kwargs)

    def forward_nhwc(self, images: torch.Tensor, prenormalised: bool = False):
        """images [NB,3,H,W] (uint8 or float, 0..255 unless prenormalised).  Returns (list of NHWC buffers, graph)."""
        graph, dt = self.graph_for(images.shape[2], images.shape[3])
        # BatchNorm mode as in torch: a BatchNorm2d module in training mode normalises with batch statistics and updates its running
        # statistics; `model.apply(set_bn_eval)` (train_net_dynamic.py:17-20) puts the modules in eval mode -> running statistics, folded
        bns = getattr(self, "_bn_modules", None)
        if bns is None:                                           # (cached: walking ~300 modules per step is host time on the small-batch step)
            bns = self._bn_modules = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        bn_train = any(m.training for m in bns)
        if bn_train and not all(m.training for m in bns):
            raise L.DinError("BatchNorm modules of one backbone must all be in the same mode (all train or all eval)")
        outs = NHWCGraphFunction.apply(graph, dt, images, prenormalised, bn_train, *self._ordered_params(graph))
        if bn_train:
            with torch.no_grad():                                 # one launch for all ~94 counters instead of one each
                torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
        if not isinstance(outs, tuple):
            outs = (outs,)
        return list(outs), graph

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        """API parity with the reference: x is already normalised (prep_images), returns NCHW fp32 maps."""
        bufs, graph = self.forward_nhwc(x, prenormalised=True)
        outs = []
        for buf, (tid, coff, c) in zip(bufs, self.output_views(graph)):
            outs.append(NHWCToNCHWFunction.apply(buf, coff, c, graph.tensors[tid].relu_masked))
        return outs

    def output_views(self, graph: Graph):
        return [(t, 0, graph.tensors[t].c) for t in graph.output_tids]


class MyVGG16(_GraphBackbone):
    """reference backbone/backbone.py:88-99 (torchvision vgg16().features, table 'D')."""

    def __init__(self, pretrained: bool = False, compute_dtype: str = "fp32"):
        super().__init__(compute_dtype)
        self.pretrained_requested = pretrained
        layers, cin = [], 3
        for v in VGG16_TABLE:
            if v == "M":
                layers.append(nn.Identity())             # placeholder keeps torchvision's Sequential indices
            else:
                conv = nn.Conv2d(cin, v, kernel_size=3, padding=1)
                nn.init.normal_(conv.weight, 0.0, math.sqrt(2.0 / (cin * 9)))
                nn.init.zeros_(conv.bias)
                layers += [conv, nn.Identity()]
                cin = v
        self.features = nn.Sequential(*layers)

    def build_graph(self, h, w, dt) -> Graph:
        gb = GraphBuilder(h, w, _cpad_image(dt))
        v = gb.full(gb.g.input_tid)
        idx = 0
        for item in VGG16_TABLE:
            if item == "M":
                v = gb.pool("maxpool", v, 2, 2, 0)
                idx += 1
            else:
                v = gb.conv(f"features.{idx}", v, item, (3, 3), (1, 1), (1, 1), relu=True, bias=True)
                idx += 2
        gb.g.output_tids = [v.tid]
        return gb.g


class _BasicConv2d(nn.Module):
    """parameter holder with torchvision's names: .conv.weight, .bn.{weight,bias,running_mean,running_var}"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=k, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)
        nn.init.normal_(self.conv.weight, 0.0, math.sqrt(2.0 / (cin * k[0] * k[1])))


def _inception_specs():
    """(module path, cin, cout, k, s, p) for every BasicConv2d up to Mixed_6e (torchvision Inception3)."""
    def a(pre, cin, pf):
        return [(pre + "branch1x1", cin, 64, (1, 1), (1, 1), (0, 0)), (pre + "branch5x5_1", cin, 48, (1, 1), (1, 1), (0, 0)),
                (pre + "branch5x5_2", 48, 64, (5, 5), (1, 1), (2, 2)), (pre + "branch3x3dbl_1", cin, 64, (1, 1), (1, 1), (0, 0)),
                (pre + "branch3x3dbl_2", 64, 96, (3, 3), (1, 1), (1, 1)), (pre + "branch3x3dbl_3", 96, 96, (3, 3), (1, 1), (1, 1)),
                (pre + "branch_pool", cin, pf, (1, 1), (1, 1), (0, 0))]

    def b(pre, cin):
        return [(pre + "branch3x3", cin, 384, (3, 3), (2, 2), (0, 0)), (pre + "branch3x3dbl_1", cin, 64, (1, 1), (1, 1), (0, 0)),
                (pre + "branch3x3dbl_2", 64, 96, (3, 3), (1, 1), (1, 1)), (pre + "branch3x3dbl_3", 96, 96, (3, 3), (2, 2), (0, 0))]

    def c(pre, cin, c7):
        return [(pre + "branch1x1", cin, 192, (1, 1), (1, 1), (0, 0)), (pre + "branch7x7_1", cin, c7, (1, 1), (1, 1), (0, 0)),
                (pre + "branch7x7_2", c7, c7, (1, 7), (1, 1), (0, 3)), (pre + "branch7x7_3", c7, 192, (7, 1), (1, 1), (3, 0)),
                (pre + "branch7x7dbl_1", cin, c7, (1, 1), (1, 1), (0, 0)), (pre + "branch7x7dbl_2", c7, c7, (7, 1), (1, 1), (3, 0)),
                (pre + "branch7x7dbl_3", c7, c7, (1, 7), (1, 1), (0, 3)), (pre + "branch7x7dbl_4", c7, c7, (7, 1), (1, 1), (3, 0)),
                (pre + "branch7x7dbl_5", c7, 192, (1, 7), (1, 1), (0, 3)), (pre + "branch_pool", cin, 192, (1, 1), (1, 1), (0, 0))]

    s = [("Conv2d_1a_3x3", 3, 32, (3, 3), (2, 2), (0, 0)), ("Conv2d_2a_3x3", 32, 32, (3, 3), (1, 1), (0, 0)),
         ("Conv2d_2b_3x3", 32, 64, (3, 3), (1, 1), (1, 1)), ("Conv2d_3b_1x1", 64, 80, (1, 1), (1, 1), (0, 0)),
         ("Conv2d_4a_3x3", 80, 192, (3, 3), (1, 1), (0, 0))]
    s += a("Mixed_5b.", 192, 32) + a("Mixed_5c.", 256, 64) + a("Mixed_5d.", 288, 64) + b("Mixed_6a.", 288)
    s += c("Mixed_6b.", 768, 128) + c("Mixed_6c.", 768, 160) + c("Mixed_6d.", 768, 160) + c("Mixed_6e.", 768, 192)
    return s


class MyInception_v3(_GraphBackbone):
    """reference backbone/backbone.py:10-85: Inception-v3 truncated after Mixed_6e, outputs [Mixed_5d, Mixed_6e].

    BatchNorm follows the modules' mode like torch: eval() (the reference's `set_bn_eval`, train_net_dynamic.py:17-20) = running
    statistics, folded into the packed filters, gamma / beta still receive gradients; train() (the reference's stage-2 default,
    config.py:80) = batch statistics over the B*T frames of this process + running-statistics update (csrc/bn.hip)."""

    def __init__(self, transform_input: bool = False, pretrained: bool = False, compute_dtype: str = "fp32"):
        super().__init__(compute_dtype)
        if transform_input:
            raise NotImplementedError("transform_input=True is never used by the DIN path (infer_model.py:32)")
        self.transform_input = transform_input
        self.pretrained_requested = pretrained
        self._specs = _inception_specs()
        for path, cin, cout, k, _s, _p in self._specs:
            mod = self
            parts = path.split(".")
            for part in parts[:-1]:
                if not hasattr(mod, part):
                    setattr(mod, part, nn.Module())
                mod = getattr(mod, part)
            setattr(mod, parts[-1], _BasicConv2d(cin, cout, k))
        # DIN_ROI_COMPOSE=0: materialise the multi-scale fuse [5d | resize(6e)] (infer_model.py:165-172) inside the graph, as round 1 did.
        # Default: the graph ends at Mixed_5d / Mixed_6e and RoIAlign samples Mixed_6e through its virtual resize (ops.RoIAlignMultiScale).
        self.materialise_fuse = not L.host_flag("DIN_ROI_COMPOSE", True)

    def build_graph(self, h, w, dt) -> Graph:
        gb = GraphBuilder(h, w, _cpad_image(dt))
        spec = {s[0]: s for s in self._specs}

        def bc(name, src: View, dst=None, pooled=None) -> View:
            _, _cin, cout, k, s, p = spec[name]
            return gb.conv(name, src, cout, k, s, p, relu=True, bn=True, dst=dst, pooled=pooled)

        # branch_pool = BasicConv2d_1x1(avg_pool2d(x, 3, 1, 1)) (torchvision InceptionA/C.forward): run as conv1x1 -> avgpool so the
        # pool moves cout (32..192) instead of cin (192..768) channels.  DIN_POOL_COMMUTE=0 keeps the reference's op order.
        commute = L.host_flag("DIN_POOL_COMMUTE", True)

        def branch_pool(name, src: View, dst: View, mid=None):
            if commute:
                _, _cin, cout, k, s, p = spec[name]
                return gb.conv(name, src, cout, k, s, p, relu=True, bn=True, dst=dst, pooled=(3, 1, 1), mid=mid)
            return bc(name, gb.pool("avgpool", src, 3, 1, 1), dst)

        v = gb.full(gb.g.input_tid)
        v = bc("Conv2d_1a_3x3", v)
        v = bc("Conv2d_2a_3x3", v)
        v = bc("Conv2d_2b_3x3", v)
        v = gb.pool("maxpool", v, 3, 2, 0)
        v = bc("Conv2d_3b_1x1", v)
        v = bc("Conv2d_4a_3x3", v)
        v = gb.pool("maxpool", v, 3, 2, 0)
        ts = gb.g.tensors[v.tid]
        h5, w5 = ts.h, ts.w
        fused_tid = None
        for blk, pf in (("Mixed_5b.", 32), ("Mixed_5c.", 64), ("Mixed_5d.", 64)):
            ctot = 64 + 64 + 96 + pf
            if blk == "Mixed_5d." and self.materialise_fuse:
                # Mixed_5d writes straight into the fused multi-scale tensor [5d (288) | resize(6e) (768)]
                fused_tid = gb.tensor(h5, w5, ctot + 768)
                out_tid, base = fused_tid, 0
            else:
                out_tid, base = gb.tensor(h5, w5, ctot), 0
            # the three 1x1 convs that read the block input are laid out for ONE forward launch (nhwc.Graph.fwd_groups): consecutive ops,
            # the two temporaries adjacent views of one tensor.  Separately they are three 64-wide launches bound by re-reading the input.
            # The branch_pool conv as a fourth, raw-stored sibling (din_conv_fwd2 craw): no gain on the 128-wide tiles of round 2 (208 / 240 filters
            # needed two tiles instead of one 192-wide), but the round-5 kernel runs 176 and 208 / 240 filters alike as two classes of 128:
            # the pool conv's own launch goes away (+0.15 % end to end; DIN_FUSE_POOL=0 restores it)
            pool4 = commute and L.host_flag("DIN_FUSE_POOL", True)
            tmp_tid = gb.tensor(h5, w5, 48 + 64 + (pf if pool4 else 0))
            bc(blk + "branch1x1", v, View(out_tid, base, 64))
            t5 = bc(blk + "branch5x5_1", v, View(tmp_tid, 0, 48))
            t3 = bc(blk + "branch3x3dbl_1", v, View(tmp_tid, 48, 64))
            if pool4:
                branch_pool(blk + "branch_pool", v, View(out_tid, base + 224, pf), mid=View(tmp_tid, 112, pf))
            gb.fuse_forward(4 if pool4 else 3)
            bc(blk + "branch5x5_2", t5, View(out_tid, base + 64, 64))
            t = bc(blk + "branch3x3dbl_2", t3)
            bc(blk + "branch3x3dbl_3", t, View(out_tid, base + 128, 96))
            if not pool4:
                branch_pool(blk + "branch_pool", v, View(out_tid, base + 224, pf))
            v = View(out_tid, base, ctot)
        v5d = v
        # Mixed_6a (InceptionB)
        h6, w6 = (h5 - 3) // 2 + 1, (w5 - 3) // 2 + 1
        out_tid = gb.tensor(h6, w6, 768)
        bc("Mixed_6a.branch3x3", v, View(out_tid, 0, 384))
        t = bc("Mixed_6a.branch3x3dbl_1", v)
        t = bc("Mixed_6a.branch3x3dbl_2", t)
        bc("Mixed_6a.branch3x3dbl_3", t, View(out_tid, 384, 96))
        gb.pool("maxpool", v, 3, 2, 0, View(out_tid, 480, 288))
        gb.g.tensors[out_tid].relu_masked = True     # pool branch of ReLU outputs: masking by (y>0) is exact
        v = View(out_tid, 0, 768)
        for blk in ("Mixed_6b.", "Mixed_6c.", "Mixed_6d.", "Mixed_6e."):
            out_tid = gb.tensor(h6, w6, 768)
            c7 = spec[blk + "branch7x7_1"][2]
            bc(blk + "branch1x1", v, View(out_tid, 0, 192))
            fuse6 = L.host_flag("DIN_FUSE_FWD6", True)
            # round 5: with the filters resident in registers (conv1x1_regw_kernel: classes of 192 filters on one XCD share the pixel stream)
            # the commuted branch_pool conv rides as a FOURTH, raw-stored sibling: one pass over the block input instead of two
            # (DIN_FUSE_POOL6=0 restores the separate launch)
            pool6 = fuse6 and commute and L.host_flag("DIN_FUSE_POOL6", True)
            if fuse6:
                tmp_tid = gb.tensor(h6, w6, 2 * c7 + (192 if pool6 else 0))
                t7 = bc(blk + "branch7x7_1", v, View(tmp_tid, 0, c7))
                td = bc(blk + "branch7x7dbl_1", v, View(tmp_tid, c7, c7))
                if pool6:
                    branch_pool(blk + "branch_pool", v, View(out_tid, 576, 192), mid=View(tmp_tid, 2 * c7, 192))
                gb.fuse_forward(4 if pool6 else 3)
            else:
                t7 = bc(blk + "branch7x7_1", v)
                td = bc(blk + "branch7x7dbl_1", v)
            t = bc(blk + "branch7x7_2", t7)
            bc(blk + "branch7x7_3", t, View(out_tid, 192, 192))
            t = bc(blk + "branch7x7dbl_2", td)
            t = bc(blk + "branch7x7dbl_3", t)
            t = bc(blk + "branch7x7dbl_4", t)
            bc(blk + "branch7x7dbl_5", t, View(out_tid, 384, 192))
            if not pool6:
                branch_pool(blk + "branch_pool", v, View(out_tid, 576, 192))
            v = View(out_tid, 0, 768)
        if self.materialise_fuse:
            # multiscale fuse (infer_model.py:165-172): resize Mixed_6e to the Mixed_5d grid into channels [288, 1056)
            gb.bilinear(v, View(fused_tid, 288, 768))
            gb.g.tensors[fused_tid].relu_masked = True   # bilinear of non-negative maps: zero output <=> all taps zero
            gb.g.output_tids = [fused_tid, v.tid]
        else:
            gb.g.output_tids = [v5d.tid, v.tid]
        self._v5d = v5d
        return gb.g

    def output_views(self, graph: Graph):
        t5d, t6e = graph.output_tids                     # (t5d is the fused tensor when the fuse is materialised: channels [0, 288) are Mixed_5d)
        return [(t5d, 0, 288), (t6e, 0, 768)]
