"""CPU oracle for the DIN stage-2 hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch-CPU fp32/fp64 tensor algebra) of the
reference algorithm for the path named by BASELINE.json `north_star`.  It is the *checker*:
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product package (`din_amd`) never imports anything from `oracle/`.

Pinning status (see DESIGN.md "Oracle"):
  * prep / DIN (Dynamic Relation + Dynamic Walk, fwd and autograd bwd) / trunk+head wiring:
    PINNED against golden vectors captured by importing the reference's own Python modules
    in the build container (tools/gen_golden.py -> tests/golden/*.npz).
  * RoIAlign (longcw/RoIAlign.pytorch, un-vendored, no version pin: reference Dockerfile:6)
    and the torchvision 0.4 VGG16 / Inception-v3 layer tables (un-vendored): restated from
    their published algorithms; the reference holds no tests/vectors for them
    -> "parity unpinned" for those two rows (self-consistency fixtures only).

Every function cites the reference file:line (relative to the reference root) it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------
# Row P -- image normalisation (utils.py:8-19)
# ----------------------------------------------------------------------------------------
def prep_images(images: Tensor) -> Tensor:
    """y = ((x / 255) - 0.5) * 2, three separate roundings like utils.py:14-17."""
    y = images / 255.0
    y = y - 0.5
    return y * 2.0


# ----------------------------------------------------------------------------------------
# Row V -- VGG16 `features` (backbone/backbone.py:88-99 -> torchvision vgg16 table 'D')
# ----------------------------------------------------------------------------------------
VGG16_TABLE: Tuple = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M",
                      512, 512, 512, "M", 512, 512, 512, "M")


def vgg16_conv_indices() -> List[int]:
    """Sequential indices of the conv layers inside `vgg.features` (0,2,5,7,10,...)."""
    idx, out = 0, []
    for v in VGG16_TABLE:
        if v == "M":
            idx += 1
        else:
            out.append(idx)
            idx += 2
    return out


def vgg16_features(x: Tensor, p: Params, prefix: str = "backbone.features.") -> List[Tensor]:
    """conv3x3(s1,p1)+bias -> ReLU, MaxPool 2x2 s2 after each block.  Returns [fm]."""
    idx = 0
    for v in VGG16_TABLE:
        if v == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, p[f"{prefix}{idx}.weight"], p[f"{prefix}{idx}.bias"], padding=1))
            idx += 2
    return [x]


def vgg16_param_shapes(prefix: str = "backbone.features.") -> Dict[str, Tuple[int, ...]]:
    shapes, cin = {}, 3
    for i, v in zip(vgg16_conv_indices(), [c for c in VGG16_TABLE if c != "M"]):
        shapes[f"{prefix}{i}.weight"] = (v, cin, 3, 3)
        shapes[f"{prefix}{i}.bias"] = (v,)
        cin = v
    return shapes


# ----------------------------------------------------------------------------------------
# Row I -- Inception-v3 up to Mixed_6e (backbone/backbone.py:10-85 -> torchvision inception)
# ----------------------------------------------------------------------------------------
# Each BasicConv2d = Conv(bias=False) + BatchNorm(eps=1e-3) + ReLU.
# spec: name -> (cin, cout, (kh,kw), (sh,sw), (ph,pw))
def _inception_a(prefix: str, cin: int, pool_features: int):
    return [
        (prefix + "branch1x1", cin, 64, (1, 1), (1, 1), (0, 0)),
        (prefix + "branch5x5_1", cin, 48, (1, 1), (1, 1), (0, 0)),
        (prefix + "branch5x5_2", 48, 64, (5, 5), (1, 1), (2, 2)),
        (prefix + "branch3x3dbl_1", cin, 64, (1, 1), (1, 1), (0, 0)),
        (prefix + "branch3x3dbl_2", 64, 96, (3, 3), (1, 1), (1, 1)),
        (prefix + "branch3x3dbl_3", 96, 96, (3, 3), (1, 1), (1, 1)),
        (prefix + "branch_pool", cin, pool_features, (1, 1), (1, 1), (0, 0)),
    ]


def _inception_b(prefix: str, cin: int):
    return [
        (prefix + "branch3x3", cin, 384, (3, 3), (2, 2), (0, 0)),
        (prefix + "branch3x3dbl_1", cin, 64, (1, 1), (1, 1), (0, 0)),
        (prefix + "branch3x3dbl_2", 64, 96, (3, 3), (1, 1), (1, 1)),
        (prefix + "branch3x3dbl_3", 96, 96, (3, 3), (2, 2), (0, 0)),
    ]


def _inception_c(prefix: str, cin: int, c7: int):
    return [
        (prefix + "branch1x1", cin, 192, (1, 1), (1, 1), (0, 0)),
        (prefix + "branch7x7_1", cin, c7, (1, 1), (1, 1), (0, 0)),
        (prefix + "branch7x7_2", c7, c7, (1, 7), (1, 1), (0, 3)),
        (prefix + "branch7x7_3", c7, 192, (7, 1), (1, 1), (3, 0)),
        (prefix + "branch7x7dbl_1", cin, c7, (1, 1), (1, 1), (0, 0)),
        (prefix + "branch7x7dbl_2", c7, c7, (7, 1), (1, 1), (3, 0)),
        (prefix + "branch7x7dbl_3", c7, c7, (1, 7), (1, 1), (0, 3)),
        (prefix + "branch7x7dbl_4", c7, c7, (7, 1), (1, 1), (3, 0)),
        (prefix + "branch7x7dbl_5", c7, 192, (1, 7), (1, 1), (0, 3)),
        (prefix + "branch_pool", cin, 192, (1, 1), (1, 1), (0, 0)),
    ]


def inception_v3_conv_specs(prefix: str = "backbone."):
    s = [
        (prefix + "Conv2d_1a_3x3", 3, 32, (3, 3), (2, 2), (0, 0)),
        (prefix + "Conv2d_2a_3x3", 32, 32, (3, 3), (1, 1), (0, 0)),
        (prefix + "Conv2d_2b_3x3", 32, 64, (3, 3), (1, 1), (1, 1)),
        (prefix + "Conv2d_3b_1x1", 64, 80, (1, 1), (1, 1), (0, 0)),
        (prefix + "Conv2d_4a_3x3", 80, 192, (3, 3), (1, 1), (0, 0)),
    ]
    s += _inception_a(prefix + "Mixed_5b.", 192, 32)
    s += _inception_a(prefix + "Mixed_5c.", 256, 64)
    s += _inception_a(prefix + "Mixed_5d.", 288, 64)
    s += _inception_b(prefix + "Mixed_6a.", 288)
    s += _inception_c(prefix + "Mixed_6b.", 768, 128)
    s += _inception_c(prefix + "Mixed_6c.", 768, 160)
    s += _inception_c(prefix + "Mixed_6d.", 768, 160)
    s += _inception_c(prefix + "Mixed_6e.", 768, 192)
    return s


def inception_v3_param_shapes(prefix: str = "backbone.") -> Dict[str, Tuple[int, ...]]:
    shapes = {}
    for name, cin, cout, k, _s, _p in inception_v3_conv_specs(prefix):
        shapes[name + ".conv.weight"] = (cout, cin, k[0], k[1])
        for b in ("weight", "bias", "running_mean", "running_var"):
            shapes[name + ".bn." + b] = (cout,)
    return shapes


def _basic_conv(x: Tensor, p: Params, name: str, stride, padding, bn_train: bool) -> Tensor:
    x = F.conv2d(x, p[name + ".conv.weight"], None, stride=stride, padding=padding)
    x = F.batch_norm(x, p[name + ".bn.running_mean"].detach(), p[name + ".bn.running_var"].detach(),
                     p[name + ".bn.weight"], p[name + ".bn.bias"], training=bn_train, eps=1e-3)
    return F.relu(x)


def inception_v3_features(x: Tensor, p: Params, prefix: str = "backbone.",
                          bn_train: bool = False) -> List[Tensor]:
    """Returns [Mixed_5d, Mixed_6e] like backbone.py:35-85 (transform_input=False)."""
    specs = {s[0]: s for s in inception_v3_conv_specs(prefix)}

    def bc(t, name):
        _, _, _, k, s, pad = specs[prefix + name]
        return _basic_conv(t, p, prefix + name, s, pad, bn_train)

    x = bc(x, "Conv2d_1a_3x3")
    x = bc(x, "Conv2d_2a_3x3")
    x = bc(x, "Conv2d_2b_3x3")
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    x = bc(x, "Conv2d_3b_1x1")
    x = bc(x, "Conv2d_4a_3x3")
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    for blk in ("Mixed_5b.", "Mixed_5c.", "Mixed_5d."):
        b1 = bc(x, blk + "branch1x1")
        b5 = bc(bc(x, blk + "branch5x5_1"), blk + "branch5x5_2")
        b3 = bc(bc(bc(x, blk + "branch3x3dbl_1"), blk + "branch3x3dbl_2"), blk + "branch3x3dbl_3")
        bp = bc(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1), blk + "branch_pool")
        x = torch.cat([b1, b5, b3, bp], 1)
    out0 = x
    blk = "Mixed_6a."
    b3 = bc(x, blk + "branch3x3")
    bd = bc(bc(bc(x, blk + "branch3x3dbl_1"), blk + "branch3x3dbl_2"), blk + "branch3x3dbl_3")
    bp = F.max_pool2d(x, kernel_size=3, stride=2)
    x = torch.cat([b3, bd, bp], 1)
    for blk in ("Mixed_6b.", "Mixed_6c.", "Mixed_6d.", "Mixed_6e."):
        b1 = bc(x, blk + "branch1x1")
        b7 = bc(bc(bc(x, blk + "branch7x7_1"), blk + "branch7x7_2"), blk + "branch7x7_3")
        bd = x
        for i in range(1, 6):
            bd = bc(bd, blk + f"branch7x7dbl_{i}")
        bp = bc(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1), blk + "branch_pool")
        x = torch.cat([b1, b7, bd, bp], 1)
    return [out0, x]


# ----------------------------------------------------------------------------------------
# Row M -- multiscale fuse (infer_model.py:165-172)
# ----------------------------------------------------------------------------------------
def bilinear_resize_align_corners(x: Tensor, oh: int, ow: int) -> Tensor:
    """F.interpolate(mode='bilinear', align_corners=True) restated explicitly:
    src = dst * (in-1)/(out-1); lerp between floor and floor+1 (clamped)."""
    n, c, ih, iw = x.shape
    sy = (ih - 1) / (oh - 1) if oh > 1 else 0.0
    sx = (iw - 1) / (ow - 1) if ow > 1 else 0.0
    yy = torch.arange(oh, dtype=x.dtype) * x.new_tensor(sy)
    xx = torch.arange(ow, dtype=x.dtype) * x.new_tensor(sx)
    y0 = yy.floor().long().clamp(0, ih - 1)
    x0 = xx.floor().long().clamp(0, iw - 1)
    y1 = (y0 + 1).clamp(max=ih - 1)
    x1 = (x0 + 1).clamp(max=iw - 1)
    ly = (yy - y0.to(x.dtype)).view(1, 1, oh, 1)
    lx = (xx - x0.to(x.dtype)).view(1, 1, 1, ow)
    top = x[:, :, y0][:, :, :, x0] * (1 - lx) + x[:, :, y0][:, :, :, x1] * lx
    bot = x[:, :, y1][:, :, :, x0] * (1 - lx) + x[:, :, y1][:, :, :, x1] * lx
    return top * (1 - ly) + bot * ly


def multiscale_fuse(outputs: Sequence[Tensor], oh: int, ow: int) -> Tensor:
    assert tuple(outputs[0].shape[2:4]) == (oh, ow)
    feats = []
    for f in outputs:
        if tuple(f.shape[2:4]) != (oh, ow):
            f = bilinear_resize_align_corners(f, oh, ow)
        feats.append(f)
    return torch.cat(feats, dim=1)


# ----------------------------------------------------------------------------------------
# Row X -- box -> frame index (infer_model.py:155-157)
# ----------------------------------------------------------------------------------------
def boxes_frame_index(bt: int, n: int) -> Tensor:
    return torch.arange(bt, dtype=torch.int32).repeat_interleave(n)


# ----------------------------------------------------------------------------------------
# Row R -- RoIAlign = TF crop_and_resize with transform_fpcoor=True
# (third-party longcw/RoIAlign.pytorch; call site infer_model.py:178-180)  PARITY UNPINNED
# ----------------------------------------------------------------------------------------
def roi_align_sample_grid(boxes: Tensor, hf: int, wf: int, k: int):
    """Per-box sample coordinates in_y[M,K], in_x[M,K] in feature pixels, using exactly the
    fp32 operation order of SURVEY row R.  Returns (in_y, in_x)."""
    f32 = torch.float32
    b = boxes.to(f32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    kf = torch.tensor(float(k), dtype=f32)
    sw = (x2 - x1) / kf
    sh = (y2 - y1) / kf
    wm1 = torch.tensor(float(wf - 1), dtype=f32)
    hm1 = torch.tensor(float(hf - 1), dtype=f32)
    nx0 = (x1 + sw / 2 - 0.5) / wm1
    ny0 = (y1 + sh / 2 - 0.5) / hm1
    nw = sw * torch.tensor(float(k - 1), dtype=f32) / wm1
    nh = sh * torch.tensor(float(k - 1), dtype=f32) / hm1
    y1n, x1n, y2n, x2n = ny0, nx0, ny0 + nh, nx0 + nw
    idx = torch.arange(k, dtype=f32)
    if k > 1:
        hs = (y2n - y1n) * hm1 / torch.tensor(float(k - 1), dtype=f32)
        ws = (x2n - x1n) * wm1 / torch.tensor(float(k - 1), dtype=f32)
        in_y = (y1n * hm1)[:, None] + idx[None, :] * hs[:, None]
        in_x = (x1n * wm1)[:, None] + idx[None, :] * ws[:, None]
    else:
        in_y = (0.5 * (y1n + y2n) * hm1)[:, None].expand(-1, 1)
        in_x = (0.5 * (x1n + x2n) * wm1)[:, None].expand(-1, 1)
    return in_y, in_x


def roi_align(fm: Tensor, boxes: Tensor, box_ind: Tensor, k: int, return_index: bool = False):
    """fm [NB,C,Hf,Wf]; boxes [M,4]=(x1,y1,x2,y2) feature px; box_ind [M] int32.
    -> [M,C,K,K].  Differentiable w.r.t. fm only (no grad to boxes).
    Integer decisions (top/bottom/left/right index, out-of-range flags) are returned when
    `return_index` so tests can compare them bit-exactly with the HIP kernel."""
    nb, c, hf, wf = fm.shape
    m = boxes.shape[0]
    in_y, in_x = roi_align_sample_grid(boxes.detach(), hf, wf, k)       # fp32 [M,K]
    oob_y = (in_y < 0) | (in_y > float(hf - 1))
    oob_x = (in_x < 0) | (in_x > float(wf - 1))
    top = in_y.floor()
    bot = in_y.ceil()
    left = in_x.floor()
    right = in_x.ceil()
    ly = (in_y - top).to(fm.dtype)
    lx = (in_x - left).to(fm.dtype)
    ti = top.long().clamp(0, hf - 1)
    bi = bot.long().clamp(0, hf - 1)
    li = left.long().clamp(0, wf - 1)
    ri = right.long().clamp(0, wf - 1)
    frames = box_ind.long()
    flat_fm = fm.reshape(nb, c, hf * wf)

    def g(yi, xi):                                                       # -> [M,C,K,K]
        # gather per source frame (never materialises fm[box_ind]: M copies of the map)
        lin = (yi[:, :, None] * wf + xi[:, None, :]).reshape(m, k * k)
        out_ = fm.new_zeros((m, c, k * k))
        for f in torch.unique(frames).tolist():
            sel = (frames == f).nonzero(as_tuple=True)[0]
            if 0 <= f < nb:
                vals = flat_fm[f][:, lin[sel].reshape(-1)]               # [C, n_sel*K*K]
                out_ = out_.index_copy(0, sel, vals.reshape(c, sel.numel(), k * k).permute(1, 0, 2))
        return out_.reshape(m, c, k, k)

    tl, tr, bl, br = g(ti, li), g(ti, ri), g(bi, li), g(bi, ri)
    lxb = lx[:, None, None, :]
    lyb = ly[:, None, :, None]
    topv = tl + (tr - tl) * lxb
    botv = bl + (br - bl) * lxb
    out = topv + (botv - topv) * lyb
    dead = (oob_y[:, :, None] | oob_x[:, None, :])[:, None]
    out = torch.where(dead, torch.zeros((), dtype=out.dtype), out)
    if return_index:
        return out, dict(top=ti.int(), bot=bi.int(), left=li.int(), right=ri.int(),
                         oob_y=oob_y, oob_x=oob_x)
    return out


# ----------------------------------------------------------------------------------------
# Rows D0-D4 -- Dynamic Relation + Dynamic Walk (infer_module/dynamic_infer_module.py)
# ----------------------------------------------------------------------------------------
def din_lattice(kernel: Tuple[int, int], ratio: int):
    """(ky_k, kx_k) lattice offsets, row-major over the kernel (:385-392)."""
    kh, kw = kernel
    ys = [(-((kh - 1) * ratio) // 2) + i * ratio for i in range(kh)]
    xs = [(-((kw - 1) * ratio) // 2) + i * ratio for i in range(kw)]
    # python floor-division on the negative number mirrors `-(field-1)//2` at :388-389
    ky = [y for y in ys for _ in xs]
    kx = [x for _ in ys for x in xs]
    return ky, kx


def din_sampling(offset: Tensor, t: int, n: int, kernel: Tuple[int, int], ratio: int):
    """offset [B,T,N,2*k2] (y block first: :199, Q2) -> integer corners + clamped position.
    Returns dict(ly,ry,lx,rx int64 [B,T,N,k2]; py,px float [B,T,N,k2]) per :208-229."""
    kh, kw = kernel
    k2 = kh * kw
    pt, pl = (kh - 1) // 2 * ratio, (kw - 1) // 2 * ratio
    hp, wp = t + 2 * pt, n + 2 * pl
    ky, kx = din_lattice(kernel, ratio)
    dt = offset.dtype
    base_y = (torch.arange(t, dtype=dt) + pt).view(1, t, 1, 1) + torch.tensor(ky, dtype=dt).view(1, 1, 1, k2)
    base_x = (torch.arange(n, dtype=dt) + pl).view(1, 1, n, 1) + torch.tensor(kx, dtype=dt).view(1, 1, 1, k2)
    py0 = base_y + offset[..., :k2]
    px0 = base_x + offset[..., k2:]
    fy = py0.detach().floor()
    fx = px0.detach().floor()
    ly = fy.clamp(0, hp - 1)
    ry = (fy + 1).clamp(0, hp - 1)
    lx = fx.clamp(0, wp - 1)
    rx = (fx + 1).clamp(0, wp - 1)
    py = py0.clamp(0, hp - 1)
    px = px0.clamp(0, wp - 1)
    return dict(ly=ly, ry=ry, lx=lx, rx=rx, py=py, px=px, hp=hp, wp=wp, pt=pt, pl=pl)


def din_ratio_forward(x: Tensor, p_w: Tensor, p_b: Tensor, s_w: Optional[Tensor], s_b: Optional[Tensor],
                      kernel: Tuple[int, int], ratio: int, want_aux: bool = False):
    """One (module, ratio) pass: x [B,T,N,C] -> Z [B,T,N,C], S [B,T,N,k2,C]  (:184-282)."""
    b, t, n, c = x.shape
    kh, kw = kernel
    k2 = kh * kw
    pt, pl = (kh - 1) // 2 * ratio, (kw - 1) // 2 * ratio
    xc = x.permute(0, 3, 1, 2)                                                 # [B,C,T,N]
    offset = F.conv2d(xc, p_w, p_b, padding=(pt, pl), dilation=ratio).permute(0, 2, 3, 1)
    g = din_sampling(offset, t, n, kernel, ratio)
    hp, wp = g["hp"], g["wp"]
    pad = F.pad(xc, (pl, pl, pt, pt)).permute(0, 2, 3, 1).reshape(b, hp * wp, c)  # P[b,y*wp+x,:]

    def fetch(cy, cx):
        lin = (cy.long() * wp + cx.long()).reshape(b, t * n * k2, 1).expand(b, t * n * k2, c)
        return pad.gather(1, lin).reshape(b, t, n, k2, c)

    def w(cy, cx):
        return (1 - (g["py"] - cy).abs()) * (1 - (g["px"] - cx).abs())

    corners = ((g["ly"], g["lx"]), (g["ry"], g["rx"]), (g["ry"], g["lx"]), (g["ly"], g["rx"]))
    s = None
    for cy, cx in corners:                                                     # no de-dup (Q3)
        term = fetch(cy, cx) * w(cy, cx).unsqueeze(-1)
        s = term if s is None else s + term
    if s_w is not None:
        logits = F.conv2d(xc, s_w, s_b, padding=(pt, pl), dilation=ratio).permute(0, 2, 3, 1)
        a = torch.softmax(logits, dim=-1)
        z = (s * a.unsqueeze(-1)).sum(3)
    else:
        a = None
        z = s.mean(3)
    if want_aux:
        return z, s, dict(offset=offset, a=a, **g)
    return z, s


def _din_padded(x: Tensor, kernel: Tuple[int, int], ratio: int):
    b, t, n, c = x.shape
    kh, kw = kernel
    pt, pl = (kh - 1) // 2 * ratio, (kw - 1) // 2 * ratio
    hp, wp = t + 2 * pt, n + 2 * pl
    xc = x.permute(0, 3, 1, 2)
    pad = F.pad(xc, (pl, pl, pt, pt)).permute(0, 2, 3, 1).reshape(b, hp * wp, c)
    return xc, pad, pt, pl, hp, wp


def din_plain_ratio_forward(x: Tensor, s_w: Optional[Tensor], s_b: Optional[Tensor], kernel: Tuple[int, int], ratio: int) -> Tensor:
    """plain_infer_ratio (:154-181): the features AT the k2 lattice points pos_0 + pos_k (no offsets, no interpolation),
    weighted by the relation softmax (scale_factor) or averaged."""
    b, t, n, c = x.shape
    kh, kw = kernel
    k2 = kh * kw
    xc, pad, pt, pl, hp, wp = _din_padded(x, kernel, ratio)
    ky, kx = din_lattice(kernel, ratio)
    cy = (torch.arange(t) + pt).view(1, t, 1, 1) + torch.tensor(ky).view(1, 1, 1, k2)          # integers inside the padded grid
    cx = (torch.arange(n) + pl).view(1, 1, n, 1) + torch.tensor(kx).view(1, 1, 1, k2)
    lin = (cy * wp + cx).expand(b, t, n, k2).reshape(b, t * n * k2, 1).expand(b, t * n * k2, c)
    ft = pad.gather(1, lin).reshape(b, t, n, k2, c)
    if s_w is not None:
        a = torch.softmax(F.conv2d(xc, s_w, s_b, padding=(pt, pl), dilation=ratio).permute(0, 2, 3, 1), dim=-1)
        return (ft * a.unsqueeze(-1)).sum(3)
    return ft.mean(3)


def din_parallel_ratio_forward(x: Tensor, p_w: Tensor, p_b: Tensor, s_w: Tensor, s_b: Tensor, kernel: Tuple[int, int], ratio: int,
                               mat_shape: Tuple[int, int]) -> Tensor:
    """parallel_infer (:285-341): relation-weighted lattice gather + the MEAN over k2 of the dynamic walk.  The walk half clamps with
    person_mat_shape (self.T, self.N) and 2 * ratio whatever the kernel: corner indices to [0, T + 2r - 1] / [0, N + 2r - 1], the
    position to [0, T + 2r] / [0, N + 2r] (:307-317) -- restated as written; index maxima are additionally held inside the padded map
    (where the reference would read outside its tensor)."""
    b, t, n, c = x.shape
    kh, kw = kernel
    k2 = kh * kw
    z_scale = din_plain_ratio_forward(x, s_w, s_b, kernel, ratio)
    xc, pad, pt, pl, hp, wp = _din_padded(x, kernel, ratio)
    offset = F.conv2d(xc, p_w, p_b, padding=(pt, pl), dilation=ratio).permute(0, 2, 3, 1)
    ky, kx = din_lattice(kernel, ratio)
    dt = offset.dtype
    py0 = (torch.arange(t, dtype=dt) + pt).view(1, t, 1, 1) + torch.tensor(ky, dtype=dt).view(1, 1, 1, k2) + offset[..., :k2]
    px0 = (torch.arange(n, dtype=dt) + pl).view(1, 1, n, 1) + torch.tensor(kx, dtype=dt).view(1, 1, 1, k2) + offset[..., k2:]
    tm, nm = mat_shape
    iy, ix = min(tm + 2 * ratio - 1, hp - 1), min(nm + 2 * ratio - 1, wp - 1)
    fy, fx = py0.detach().floor(), px0.detach().floor()
    ly, ry, lx, rx = fy.clamp(0, iy), (fy + 1).clamp(0, iy), fx.clamp(0, ix), (fx + 1).clamp(0, ix)
    py, px = py0.clamp(0, tm + 2 * ratio), px0.clamp(0, nm + 2 * ratio)

    def fetch(cy, cx):
        lin = (cy.long() * wp + cx.long()).reshape(b, t * n * k2, 1).expand(b, t * n * k2, c)
        return pad.gather(1, lin).reshape(b, t, n, k2, c)

    def w(cy, cx):
        return (1 - (py - cy).abs()) * (1 - (px - cx).abs())

    s = None
    for cy, cx in ((ly, lx), (ry, rx), (ry, lx), (ly, rx)):
        term = fetch(cy, cx) * w(cy, cx).unsqueeze(-1)
        s = term if s is None else s + term
    return z_scale + s.mean(3)


def din_person_inference(x: Tensor, p: Params, prefix: str, kernel: Tuple[int, int],
                         ratios: Sequence[int], scale_factor: bool = True,
                         beta_factor: bool = False, dynamic_sampling: bool = True, parallel_inference: bool = False,
                         mat_shape: Tuple[int, int] = (10, 12)) -> Tuple[Tensor, Tensor]:
    """Dynamic_Person_Inference.forward (:121-151).  dynamic_sampling=False / parallel_inference=True: the reference computes the
    per-ratio features and then dies on the unbound `ft_infer_MAD` (:151); the recipe is its own per-ratio methods followed by the
    combination at :137-147, MAD = None."""
    zs, s_last = [], None
    for r in ratios:
        sw = p.get(f"{prefix}scale_conv.{r}.weight") if scale_factor else None
        sb = p.get(f"{prefix}scale_conv.{r}.bias") if scale_factor else None
        if parallel_inference:
            z = din_parallel_ratio_forward(x, p[f"{prefix}p_conv.{r}.weight"], p[f"{prefix}p_conv.{r}.bias"], sw, sb, kernel, r, mat_shape)
        elif not dynamic_sampling:
            z = din_plain_ratio_forward(x, sw, sb, kernel, r)
        else:
            z, s_last = din_ratio_forward(x, p[f"{prefix}p_conv.{r}.weight"], p[f"{prefix}p_conv.{r}.bias"],
                                          sw, sb, kernel, r)
        zs.append(z)
    zst = torch.stack(zs, dim=4)
    if beta_factor:
        agg = (zst * p[f"{prefix}beta"]).sum(-1)
    else:
        agg = zst.mean(4)
    return agg @ p[f"{prefix}hidden_weight.weight"].t(), s_last


def din_multi_inference(x: Tensor, p: Params, prefix: str, kernels: Sequence[Tuple[int, int]],
                        ratios: Sequence[int], scale_factor=True, beta_factor=False):
    """Multi_Dynamic_Inference.forward (:436-443): sum of num_DIM independent modules."""
    out, mad = None, None
    for i, k in enumerate(kernels):
        o, mad = din_person_inference(x, p, f"{prefix}DIMlist.{i}.", tuple(k), ratios, scale_factor, beta_factor)
        out = o if out is None else out + o
    return out, mad


def din_hierarchical_inference(x: Tensor, p: Params, prefix: str, kernels, ratios,
                               scale_factor=True, beta_factor=False, dropout_mask: Optional[Tensor] = None):
    """Hierarchical_Dynamic_Inference (:491-498) with the intended semantics (reference is
    broken as shipped: SURVEY section 0 bug 2): DPI_1 -> LN -> ReLU -> dropout(p=.5) -> DPI_2."""
    h, _ = din_person_inference(x, p, prefix + "DPI_1.", tuple(kernels[0]), ratios, scale_factor, beta_factor)
    w, bb = p[prefix + "hier_LN.weight"], p[prefix + "hier_LN.bias"]
    h = F.relu(F.layer_norm(h, w.shape, w, bb, 1e-5))
    if dropout_mask is not None:
        h = h * dropout_mask
    return din_person_inference(h, p, prefix + "DPI_2.", tuple(kernels[1]), ratios, scale_factor, beta_factor)


def din_param_shapes(prefix: str, c: int, kernel: Tuple[int, int], ratios: Sequence[int],
                     scale_factor=True, beta_factor=False) -> Dict[str, Tuple[int, ...]]:
    kh, kw = kernel
    k2 = kh * kw
    shapes = {prefix + "hidden_weight.weight": (c, c)}
    for r in ratios:
        shapes[f"{prefix}p_conv.{r}.weight"] = (2 * k2, c, kh, kw)
        shapes[f"{prefix}p_conv.{r}.bias"] = (2 * k2,)
        if scale_factor:
            shapes[f"{prefix}scale_conv.{r}.weight"] = (k2, c, kh, kw)
            shapes[f"{prefix}scale_conv.{r}.bias"] = (k2,)
    if beta_factor:
        shapes[prefix + "beta"] = (len(ratios),)
    return shapes


# ----------------------------------------------------------------------------------------
# Rows E, L, H + whole-network wiring (infer_model.py:141-234)
# ----------------------------------------------------------------------------------------
class OracleCfg:
    """The cfg fields the path reads (config.py:10-104); defaults = BASELINE config[0]."""
    def __init__(self, **kw):
        self.backbone = "vgg16"
        self.image_size = (720, 1280)
        self.out_size = (22, 40)
        self.emb_features = 512
        self.crop_size = (5, 5)
        self.num_boxes = 12
        self.num_frames = 3
        self.num_features_boxes = 1024
        self.num_activities = 8
        self.ST_kernel_size = [(3, 3)]
        self.sampling_ratio = [1]
        self.num_DIM = 1
        self.scale_factor = True
        self.beta_factor = False
        self.lite_dim = None
        self.hierarchical_inference = False
        self.train_dropout_prob = 0.3
        self.head_mode = "vgg16"       # which residual/LN order (infer_model.py:203-216)
        self.collective = False        # Dynamic_collective parameter layout (bare DPI, dpi_nl [T,C])
        for k, v in kw.items():
            setattr(self, k, v)


def model_param_shapes(cfg: OracleCfg) -> Dict[str, Tuple[int, ...]]:
    t, n = cfg.num_frames, cfg.num_boxes
    d, k, nfb = cfg.emb_features, cfg.crop_size[0], cfg.num_features_boxes
    shapes = {}
    if cfg.backbone == "vgg16":
        shapes.update(vgg16_param_shapes())
    elif cfg.backbone == "inv3":
        shapes.update(inception_v3_param_shapes())
    shapes["fc_emb_1.weight"] = (nfb, k * k * d)
    shapes["fc_emb_1.bias"] = (nfb,)
    shapes["nl_emb_1.weight"] = (nfb,)
    shapes["nl_emb_1.bias"] = (nfb,)
    c = cfg.lite_dim if cfg.lite_dim else nfb
    if cfg.lite_dim:
        shapes["point_conv.weight"] = (c, nfb, 1, 1)
        shapes["point_conv.bias"] = (c,)
        shapes["point_ln.weight"] = (t, n, c)
        shapes["point_ln.bias"] = (t, n, c)
    if getattr(cfg, "collective", False):
        k0 = cfg.ST_kernel_size[0] if isinstance(cfg.ST_kernel_size, list) else cfg.ST_kernel_size
        shapes.update(din_param_shapes("DPI.", c, tuple(k0), cfg.sampling_ratio, cfg.scale_factor, cfg.beta_factor))
        shapes["dpi_nl.weight"] = (t, c)
        shapes["dpi_nl.bias"] = (t, c)
        shapes["fc_activities.weight"] = (cfg.num_activities, c)
        shapes["fc_activities.bias"] = (cfg.num_activities,)
        return shapes
    if cfg.hierarchical_inference:
        for i, sub in enumerate(("DPI.DPI_1.", "DPI.DPI_2.")):
            shapes.update(din_param_shapes(sub, c, tuple(cfg.ST_kernel_size[i]), cfg.sampling_ratio,
                                           cfg.scale_factor, cfg.beta_factor))
        shapes["DPI.hier_LN.weight"] = (t, n, c)
        shapes["DPI.hier_LN.bias"] = (t, n, c)
    else:
        for i in range(cfg.num_DIM):
            shapes.update(din_param_shapes(f"DPI.DIMlist.{i}.", c, tuple(cfg.ST_kernel_size[i]),
                                           cfg.sampling_ratio, cfg.scale_factor, cfg.beta_factor))
    shapes["dpi_nl.weight"] = (t, n, c)
    shapes["dpi_nl.bias"] = (t, n, c)
    shapes["fc_activities.weight"] = (cfg.num_activities, c)
    shapes["fc_activities.bias"] = (cfg.num_activities,)
    return shapes


def embed_boxes(box_feats: Tensor, p: Params) -> Tensor:
    """fc_emb_1 -> LayerNorm([NFB]) -> ReLU (infer_model.py:184-186)."""
    y = F.linear(box_feats, p["fc_emb_1.weight"], p["fc_emb_1.bias"])
    return F.relu(F.layer_norm(y, (y.shape[-1],), p["nl_emb_1.weight"], p["nl_emb_1.bias"], 1e-5))


def lite_projection(x: Tensor, p: Params) -> Tensor:
    """1x1 conv NFB->lite + LayerNorm([T,N,lite]) + ReLU (infer_model.py:188-193)."""
    w = p["point_conv.weight"].reshape(p["point_conv.weight"].shape[0], -1)
    y = F.linear(x, w, p["point_conv.bias"])
    lw, lb = p["point_ln.weight"], p["point_ln.bias"]
    return F.relu(F.layer_norm(y, lw.shape, lw, lb, 1e-5))


def head(graph: Tensor, x: Tensor, p: Params, mode: str = "vgg16",
         dropout_mask: Optional[Tensor] = None) -> Tensor:
    """infer_model.py:203-232.  graph, x: [B,T,N,C] -> activity scores [B,A]."""
    lw, lb = p["dpi_nl.weight"], p["dpi_nl.bias"]
    if mode == "res18":
        s = F.relu(F.layer_norm(graph, lw.shape, lw, lb, 1e-5)) + x
    else:
        s = F.relu(F.layer_norm(graph + x, lw.shape, lw, lb, 1e-5))
    if dropout_mask is not None:
        s = s * dropout_mask
    pooled = s.max(dim=2).values                                        # [B,T,C]
    b, t, c = pooled.shape
    sc = F.linear(pooled.reshape(b * t, c), p["fc_activities.weight"], p["fc_activities.bias"])
    return sc.reshape(b, t, -1).mean(1)


def backbone_features(cfg: OracleCfg, images_flat: Tensor, p: Params, bn_train: bool = False) -> Tensor:
    x = prep_images(images_flat)
    if cfg.backbone == "vgg16":
        outs = vgg16_features(x, p)
    elif cfg.backbone == "inv3":
        outs = inception_v3_features(x, p, bn_train=bn_train)
    else:
        raise ValueError(cfg.backbone)
    return multiscale_fuse(outs, *cfg.out_size)


def dynamic_volleyball_forward(cfg: OracleCfg, p: Params, images: Tensor, boxes: Tensor,
                               dropout_mask: Optional[Tensor] = None,
                               return_intermediates: bool = False):
    """Dynamic_volleyball.forward (infer_model.py:141-234).  images [B,T,3,H,W] 0..255,
    boxes [B,T,N,4] feature px.  -> {'activities': [B,A]}."""
    b, t = images.shape[:2]
    n = cfg.num_boxes
    h, w = cfg.image_size
    k = cfg.crop_size[0]
    fm = backbone_features(cfg, images.reshape(b * t, 3, h, w), p)
    idx = boxes_frame_index(b * t, n)
    crops = roi_align(fm, boxes.reshape(b * t * n, 4), idx, k)          # [BTN,D,K,K]
    x = embed_boxes(crops.reshape(b, t, n, -1), p)
    if cfg.lite_dim:
        x = lite_projection(x, p)
    if cfg.hierarchical_inference:
        graph, _ = din_hierarchical_inference(x, p, "DPI.", cfg.ST_kernel_size, cfg.sampling_ratio,
                                              cfg.scale_factor, cfg.beta_factor)
    else:
        graph, _ = din_multi_inference(x, p, "DPI.", cfg.ST_kernel_size, cfg.sampling_ratio,
                                       cfg.scale_factor, cfg.beta_factor)
    scores = head(graph, x, p, cfg.head_mode, dropout_mask)
    if return_intermediates:
        return {"activities": scores}, dict(fm=fm, crops=crops, x=x, graph=graph)
    return {"activities": scores}


# ----------------------------------------------------------------------------------------
# SURVEY 8(f)-4 -- Dynamic_TCE_volleyball (infer_model.py:237-468): the DIN trunk with a context-encoding transformer in front
# ----------------------------------------------------------------------------------------
TCE_HEADS, TCE_FEATURES = 4, 128                # num_heads_context, num_features_context (infer_model.py:244-245)


def context_position_embedding(oh: int, ow: int, downscale: float = 16.0, num_pos_feats: int = 256,
                               temperature: float = 10000.0, dtype=torch.float32) -> Tensor:
    """Context_PositionEmbeddingSine(16, 512 / 2) (infer_module/positional_encoding.py:50-92; built at infer_model.py:293): the
    sine embedding [2 * num_pos_feats, OH, OW] that is ADDED to the context map.  Channels [0, 256) encode y, [256, 512) encode x;
    inside each half channel 2i is sin, 2i+1 is cos of (coordinate * 16) / T^(2i / 256); coordinates are 1-based (a cumsum of ones)."""
    y = torch.arange(1, oh + 1, dtype=torch.float32) * downscale                          # [OH]
    x = torch.arange(1, ow + 1, dtype=torch.float32) * downscale                          # [OW]
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)   # [C]
    py = y[:, None] / dim_t                                                               # [OH, C]
    px = x[:, None] / dim_t                                                               # [OW, C]
    def interleave(a):                                                                    # sin on even, cos on odd channels
        return torch.stack((a[:, 0::2].sin(), a[:, 1::2].cos()), dim=2).flatten(1)
    py, px = interleave(py), interleave(px)
    pos = torch.cat((py[:, None, :].expand(oh, ow, num_pos_feats), px[None, :, :].expand(oh, ow, num_pos_feats)), dim=2)
    return pos.permute(2, 0, 1).contiguous().to(dtype)                                    # [2C, OH, OW]


def tce_param_shapes(nfb: int, prefix: str = "multilayer_head_embfeature_context_encoding.") -> Dict[str, Tuple[int, ...]]:
    """one layer of TCE_HEADS EmbfeatureContextEncodingTransformer heads (TCE_STBiP_module.py:224-249, 289-297)"""
    c = TCE_FEATURES
    shapes = {}
    for j in range(TCE_HEADS):
        q = f"{prefix}CET.{j}."
        shapes[q + "downsample2.weight"] = (c, 512, 1, 1)
        shapes[q + "downsample2.bias"] = (c,)
        shapes[q + "emb_roi.weight"] = (c, nfb)
        shapes[q + "emb_roi.bias"] = (c,)
        for ln in ("layernorm1", "layernorm2"):
            shapes[q + ln + ".weight"] = (c,)
            shapes[q + ln + ".bias"] = (c,)
        for lin in ("FFN.0", "FFN.3"):
            shapes[q + lin + ".weight"] = (c, c)
            shapes[q + lin + ".bias"] = (c,)
    return shapes


def tce_context_encoding(roi_feature: Tensor, context: Tensor, p: Params,
                         prefix: str = "multilayer_head_embfeature_context_encoding.", return_attention: bool = False):
    """MultiHeadLayerEmbfeatureContextEncoding, one layer (TCE_STBiP_module.py:252-286, 300-313), dropout off.
    roi_feature [BT*N, NFB] (or [..., NFB]); context [BT, 512, OH, OW] (position embedding already added) -> [BT*N, HEADS * 128].
    Per head: keys = values = 1x1 conv of the context (512 -> 128); query = Linear(NFB -> 128) of the box embedding; attention over
    the OH*OW pixels of the box's own frame; LayerNorm(context + query); + FFN; LayerNorm."""
    bt, _, oh, ow = context.shape
    c = TCE_FEATURES
    roi = roi_feature.reshape(-1, roi_feature.shape[-1])
    n = roi.shape[0] // bt
    outs, atts = [], []
    for j in range(TCE_HEADS):
        q_ = f"{prefix}CET.{j}."
        img = F.conv2d(context, p[q_ + "downsample2.weight"], p[q_ + "downsample2.bias"])             # [BT,128,OH,OW]
        emb = F.linear(roi, p[q_ + "emb_roi.weight"], p[q_ + "emb_roi.bias"])                        # [BT*N,128]
        keys = img.reshape(bt, c, oh * ow)                                                             # [BT,128,P]
        a = torch.bmm(emb.reshape(bt, n, c), keys)                                                     # [BT,N,P]
        att = F.softmax(a, dim=2)
        ctx = torch.bmm(att, keys.transpose(1, 2)).reshape(bt * n, c)                                  # [BT*N,128]
        x = F.layer_norm(ctx + emb, (c,), p[q_ + "layernorm1.weight"], p[q_ + "layernorm1.bias"], 1e-5)
        f = F.linear(F.relu(F.linear(x, p[q_ + "FFN.0.weight"], p[q_ + "FFN.0.bias"])), p[q_ + "FFN.3.weight"], p[q_ + "FFN.3.bias"])
        x = F.layer_norm(x + f, (c,), p[q_ + "layernorm2.weight"], p[q_ + "layernorm2.bias"], 1e-5)
        outs.append(x)
        atts.append(att)
    out = torch.cat(outs, dim=1)
    return (out, atts) if return_attention else out


def tce_model_param_shapes(cfg: OracleCfg) -> Dict[str, Tuple[int, ...]]:
    """Dynamic_TCE_volleyball.__init__ (infer_model.py:241-351): the DIN modules, dpi_nl and (non-lite) fc_activities work on
    context_dim = in_dim + 4 * 128 channels"""
    t, n = cfg.num_frames, cfg.num_boxes
    d, k, nfb = cfg.emb_features, cfg.crop_size[0], cfg.num_features_boxes
    assert cfg.backbone == "vgg16", "the reference's Dynamic_TCE_volleyball.forward only has res18 / vgg16 head branches (infer_model.py:430-442)"
    assert not cfg.lite_dim, "with lite_dim the reference concatenates lite_dim + 512 channels but sizes fc_activities for lite_dim: it cannot run"
    shapes = dict(vgg16_param_shapes())
    shapes["fc_emb_1.weight"] = (nfb, k * k * d)
    shapes["fc_emb_1.bias"] = (nfb,)
    shapes["nl_emb_1.weight"] = (nfb,)
    shapes["nl_emb_1.bias"] = (nfb,)
    shapes.update(tce_param_shapes(nfb))
    c = nfb + TCE_HEADS * TCE_FEATURES
    if cfg.hierarchical_inference:
        for i, sub in enumerate(("DPI.DPI_1.", "DPI.DPI_2.")):
            shapes.update(din_param_shapes(sub, c, tuple(cfg.ST_kernel_size[i]), cfg.sampling_ratio, cfg.scale_factor, cfg.beta_factor))
        shapes["DPI.hier_LN.weight"] = (t, n, c)
        shapes["DPI.hier_LN.bias"] = (t, n, c)
    else:
        for i in range(cfg.num_DIM):
            shapes.update(din_param_shapes(f"DPI.DIMlist.{i}.", c, tuple(cfg.ST_kernel_size[i]), cfg.sampling_ratio, cfg.scale_factor,
                                           cfg.beta_factor))
    shapes["dpi_nl.weight"] = (t, n, c)
    shapes["dpi_nl.bias"] = (t, n, c)
    shapes["fc_activities.weight"] = (cfg.num_activities, c)
    shapes["fc_activities.bias"] = (cfg.num_activities,)
    return shapes


def tce_synth_params(cfg: OracleCfg, seed: int) -> Params:
    """Seeded parameters of the TCE fixtures (tools/gen_golden.py::tce_case and the tests share this recipe): synth_params for the trunk /
    DIN / head; for the transformer LayerNorm gains around 1, small random biases, fan-in scaled weights."""
    shapes = tce_model_param_shapes(cfg)
    p = synth_params(shapes, seed=seed + 3, din_std=0.02)
    g_ = torch.Generator().manual_seed(seed + 13)
    for k in shapes:
        if "context_encoding" in k:
            if "layernorm" in k and k.endswith("weight"):
                p[k] = 0.75 + 0.5 * torch.rand(shapes[k], generator=g_)
            elif k.endswith("bias"):
                p[k] = 0.1 * torch.randn(shapes[k], generator=g_)
            else:
                fan_in = 1
                for d_ in shapes[k][1:]:
                    fan_in *= d_
                p[k] = torch.randn(shapes[k], generator=g_) * (1.0 / fan_in) ** 0.5
    return p


def dynamic_tce_volleyball_forward(cfg: OracleCfg, p: Params, images: Tensor, boxes: Tensor, return_intermediates: bool = False):
    """Dynamic_TCE_volleyball.forward (infer_model.py:370-468), eval mode: the Dynamic_volleyball trunk up to the box embeddings, then
    [embedding | context encoding of the LAST backbone output + position embedding] (1024 + 512 channels) through DIN and the vgg16 head."""
    b, t = images.shape[:2]
    n = cfg.num_boxes
    h, w = cfg.image_size
    k = cfg.crop_size[0]
    outs = vgg16_features(prep_images(images.reshape(b * t, 3, h, w)), p)
    fm = multiscale_fuse(outs, *cfg.out_size)
    idx = boxes_frame_index(b * t, n)
    crops = roi_align(fm, boxes.reshape(b * t * n, 4), idx, k)
    x = embed_boxes(crops.reshape(b, t, n, -1), p)                                        # [B,T,N,NFB]
    context = outs[-1]
    context = context + context_position_embedding(context.shape[2], context.shape[3], dtype=context.dtype)[None]   # :404-406
    enc = tce_context_encoding(x.reshape(b * t * n, -1), context, p).reshape(b, t, n, -1)   # :408-409
    xc = torch.cat((x, enc), dim=3)                                                       # :410
    if cfg.hierarchical_inference:
        graph, _ = din_hierarchical_inference(xc, p, "DPI.", cfg.ST_kernel_size, cfg.sampling_ratio, cfg.scale_factor, cfg.beta_factor)
    else:
        graph, _ = din_multi_inference(xc, p, "DPI.", cfg.ST_kernel_size, cfg.sampling_ratio, cfg.scale_factor, cfg.beta_factor)
    scores = head(graph, xc, p, "vgg16")                                                  # :436-442, :452-466
    if return_intermediates:
        return {"activities": scores}, dict(x=x, enc=enc, context=context)
    return {"activities": scores}


# ----------------------------------------------------------------------------------------
# Row C -- Dynamic_collective (infer_model.py:1226-1319), intended semantics
# ----------------------------------------------------------------------------------------
def dynamic_collective_forward(cfg: OracleCfg, p: Params, images: Tensor, boxes: Tensor,
                               bboxes_num: Tensor):
    """Variable N per clip.  DPI is a bare Dynamic_Person_Inference (prefix 'DPI.');
    LayerNorm([T,C]) on [N,T,C]; max over actors; fc; mean over T."""
    b, t = images.shape[:2]
    mx = cfg.num_boxes
    h, w = cfg.image_size
    k = cfg.crop_size[0]
    fm = backbone_features(cfg, images.reshape(b * t, 3, h, w), p)
    idx = boxes_frame_index(b * t, mx)
    crops = roi_align(fm, boxes.reshape(b * t * mx, 4), idx, k)
    x = embed_boxes(crops.reshape(b, t, mx, -1), p)
    outs = []
    for bi in range(b):
        nb = int(bboxes_num[bi, 0])
        xb = x[bi:bi + 1, :, :nb]
        g, _ = din_person_inference(xb, p, "DPI.", tuple(cfg.ST_kernel_size[0])
                                    if isinstance(cfg.ST_kernel_size, list) else tuple(cfg.ST_kernel_size),
                                    cfg.sampling_ratio, cfg.scale_factor, cfg.beta_factor)
        s = (g + xb)[0].permute(1, 0, 2)                                # [N,T,C]
        lw, lb = p["dpi_nl.weight"], p["dpi_nl.bias"]
        s = F.relu(F.layer_norm(s, lw.shape, lw, lb, 1e-5))
        pooled = s.max(dim=0).values                                    # [T,C]
        sc = F.linear(pooled, p["fc_activities.weight"], p["fc_activities.bias"]).mean(0, keepdim=True)
        outs.append(sc)
    return {"activities": torch.cat(outs, 0)}


# ----------------------------------------------------------------------------------------
# Row S -- one training step's loss (train_net_dynamic.py:183-193)
# ----------------------------------------------------------------------------------------
def train_step_loss(cfg: OracleCfg, p: Params, images: Tensor, boxes: Tensor, labels: Tensor,
                    dropout_mask: Optional[Tensor] = None) -> Tensor:
    out = dynamic_volleyball_forward(cfg, p, images, boxes, dropout_mask)
    return F.cross_entropy(out["activities"], labels)


# ----------------------------------------------------------------------------------------
# Synthetic inputs / weights -- shared by tests, bench cpu leg, golden generator
# ----------------------------------------------------------------------------------------
def synth_params(shapes: Dict[str, Tuple[int, ...]], seed: int = 3, din_std: float = 0.02,
                 dtype=torch.float32) -> Params:
    """conv: N(0, sqrt(2/fan_in)); Linear: kaiming_normal (same formula); biases 0; LN 1/0;
    BN weight 1, bias 0, mean small noise, var ~1; DIN p_conv/scale_conv N(0, din_std) so the
    dynamic-walk path is exercised (the reference zero-inits them: dynamic_infer_module.py:66-81)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if "p_conv" in name or "scale_conv" in name:
            t = torch.randn(shp, generator=g, dtype=torch.float64) * din_std
        elif name.endswith("running_var"):
            t = 0.5 + torch.rand(shp, generator=g, dtype=torch.float64)
        elif name.endswith("running_mean"):
            t = 0.1 * torch.randn(shp, generator=g, dtype=torch.float64)
        elif ".bn." in name and name.endswith("weight"):
            t = 0.8 + 0.4 * torch.rand(shp, generator=g, dtype=torch.float64)
        elif ".bn." in name and name.endswith("bias"):
            t = 0.1 * torch.randn(shp, generator=g, dtype=torch.float64)
        elif name.endswith("beta"):
            t = torch.ones(shp, dtype=torch.float64)
        elif ("nl_" in name or "_ln" in name or "_LN" in name or "dpi_nl" in name):
            t = torch.ones(shp, dtype=torch.float64) if name.endswith("weight") else torch.zeros(shp, dtype=torch.float64)
        elif name.endswith("bias"):
            t = torch.zeros(shp, dtype=torch.float64)
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            t = torch.randn(shp, generator=g, dtype=torch.float64) * math.sqrt(2.0 / fan_in)
        out[name] = t.to(dtype)
    return out


def synth_scene_images(boxes, h: int, w: int, oh: int, ow: int, seed: int = 0):
    """uint8 frames with the statistics of a photographed scene rather than of white noise: per clip and colour plane a smooth background
    (three sinusoidal fields, spatial periods 30 .. 400 pixels) that drifts by a few pixels from frame to frame (the frames of a Volleyball
    clip are consecutive video frames, reference volleyball.py:239-243), DISTINCT actors -- inside every box (given in feature-map pixels
    like the model's `boxes_in`, [B, T, N, 4]) an oriented texture with the actor's own period, orientation, colour and brightness, the same
    in all frames of the clip -- and sensor noise N(0, 5).  Used by the full-size fixture that holds the bf16 mode's gradient direction on a
    realistic input (tools/gen_golden.py --only full_scene); regenerated from the seed and the boxes on both sides."""
    import numpy as np
    rng = np.random.default_rng(seed)
    bx = boxes.numpy() if hasattr(boxes, "numpy") else np.asarray(boxes)
    b, t, n = bx.shape[:3]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.empty((b, t, 3, h, w), dtype=np.uint8)
    sy, sx = h / float(oh), w / float(ow)
    for bi in range(b):
        comps = [[(rng.uniform(5, 64), rng.uniform(5, 64), rng.uniform(0, 6.28), rng.uniform(0, 6.28), rng.uniform(10, 35)) for _ in range(3)]
                 for _c in range(3)]
        drift = rng.uniform(-3, 3, size=(t, 2)).cumsum(0)
        actors = [(rng.uniform(3.0, 14.0), rng.uniform(0, 3.1416), rng.uniform(40, 215, 3), rng.uniform(25, 70, 3), rng.uniform(0, 6.28)) for _ in range(n)]
        for ti in range(t):
            planes = []
            for c in range(3):
                f = np.full((h, w), 112.0 + 10.0 * c)
                for (py, px, ay, ax, amp) in comps[c]:
                    f += amp * np.sin((yy + drift[ti, 0]) / py + ay) * np.cos((xx + drift[ti, 1]) / px + ax)
                planes.append(f)
            for ai in range(n):
                x1, y1, x2, y2 = bx[bi, ti, ai]
                r0, r1 = int(max(0, np.floor(y1 * sy))), int(min(h, np.ceil(y2 * sy)))
                c0, c1 = int(max(0, np.floor(x1 * sx))), int(min(w, np.ceil(x2 * sx)))
                if r1 <= r0 or c1 <= c0:
                    continue
                period, ang, mean, amp, ph = actors[ai]
                u = ((yy[r0:r1, c0:c1] - r0) * np.cos(ang) + (xx[r0:r1, c0:c1] - c0) * np.sin(ang)) * (6.2832 / period) + ph
                for c in range(3):
                    planes[c][r0:r1, c0:c1] = mean[c] + amp[c] * np.sin(u + 0.7 * c)
            for c in range(3):
                out[bi, ti, c] = np.clip(np.rint(planes[c] + rng.normal(0.0, 5.0, (h, w))), 0, 255).astype(np.uint8)
    return torch.from_numpy(out)


def synth_inputs(b: int, t: int, n: int, h: int, w: int, oh: int, ow: int, num_classes: int = 8,
                 seed: int = 0):
    """uint8-valued images, player-shaped boxes in feature px, labels."""
    import numpy as np
    r0 = np.random.default_rng(seed)
    images = torch.from_numpy(r0.integers(0, 256, size=(b, t, 3, h, w), dtype=np.uint8))
    r1 = np.random.default_rng(seed + 1)
    cx = r1.uniform(0.05, 0.95, (b, t, n)) * ow
    cy = r1.uniform(0.3, 0.9, (b, t, n)) * oh
    bw = r1.uniform(0.03, 0.08, (b, t, n)) * ow
    bh = r1.uniform(0.15, 0.35, (b, t, n)) * oh
    x1 = np.clip(cx - bw / 2, 0, ow)
    x2 = np.clip(cx + bw / 2, 0, ow)
    y1 = np.clip(cy - bh / 2, 0, oh)
    y2 = np.clip(cy + bh / 2, 0, oh)
    boxes = torch.from_numpy(np.stack([x1, y1, x2, y2], -1).astype(np.float32))
    r2 = np.random.default_rng(seed + 2)
    labels = torch.from_numpy(r2.integers(0, num_classes, size=(b,)).astype(np.int64))
    return images, boxes, labels
