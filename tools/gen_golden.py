#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules on CPU.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
Nothing from the reference is copied: we import its modules, feed seeded inputs/weights and
store the numbers it produces.  Missing third-party deps of the reference are stubbed
(SURVEY Appendix B):
  thop, fvcore.nn, skimage, cv2      -> import-only stubs
  torchvision.models.vgg16/inception_v3 -> torch.nn restatements of the published layer tables
  roi_align.roi_align.RoIAlign       -> wrapper over oracle.roi_align (third-party code absent
                                        from the tree => that row stays "parity unpinned")
While generating, every golden tensor is also cross-checked against oracle/din_oracle.py;
the script aborts if the oracle disagrees with the reference.

usage: python tools/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import din_oracle as O  # noqa: E402


# ------------------------------------------------------------------ stubs for missing deps
class _BasicConv2d(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class _IncA(nn.Module):
    def __init__(self, cin, pf):
        super().__init__()
        self.branch1x1 = _BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = _BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = _BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = _BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = _BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = _BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = _BasicConv2d(cin, pf, kernel_size=1)

    def forward(self, x):
        a = self.branch1x1(x)
        b = self.branch5x5_2(self.branch5x5_1(x))
        c = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        d = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([a, b, c, d], 1)


class _IncB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = _BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = _BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = _BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = _BasicConv2d(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        a = self.branch3x3(x)
        b = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        c = F.max_pool2d(x, kernel_size=3, stride=2)
        return torch.cat([a, b, c], 1)


class _IncC(nn.Module):
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = _BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = _BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = _BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = _BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = _BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = _BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = _BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = _BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = _BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = _BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        a = self.branch1x1(x)
        b = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        c = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        d = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([a, b, c, d], 1)


class _Inception3(nn.Module):
    def __init__(self):
        super().__init__()
        self.Conv2d_1a_3x3 = _BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = _BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = _BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = _BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = _BasicConv2d(80, 192, kernel_size=3)
        self.Mixed_5b = _IncA(192, 32)
        self.Mixed_5c = _IncA(256, 64)
        self.Mixed_5d = _IncA(288, 64)
        self.Mixed_6a = _IncB(288)
        self.Mixed_6b = _IncC(768, 128)
        self.Mixed_6c = _IncC(768, 160)
        self.Mixed_6d = _IncC(768, 160)
        self.Mixed_6e = _IncC(768, 192)


class _VGG(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in O.VGG16_TABLE:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)


class _RoIAlignStub(nn.Module):
    def __init__(self, crop_h, crop_w, extrapolation_value=0, transform_fpcoor=True):
        super().__init__()
        assert crop_h == crop_w
        self.k = crop_h

    def forward(self, fm, boxes, box_ind):
        return O.roi_align(fm, boxes, box_ind, self.k)


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ident = lambda *a, **k: None  # noqa: E731
    mod("thop", profile=ident, clever_format=ident)
    mod("fvcore")
    mod("fvcore.nn", activation_count=ident, flop_count=ident, parameter_count=ident,
        parameter_count_table=ident)
    tv = mod("torchvision")
    tvm = mod("torchvision.models", vgg16=lambda pretrained=False: _VGG(),
              inception_v3=lambda pretrained=False: _Inception3(),
              vgg19=ident, resnet18=ident, resnet50=ident, alexnet=ident)
    tvt = mod("torchvision.transforms")
    # torchvision==0.4.0 (README.md:45) transforms.functional.resize(img, size, interpolation=Image.BILINEAR) for a PIL image and a
    # (h, w) size is `img.resize(size[::-1], interpolation)` -- third-party, absent from the tree: restated from the published source
    from PIL import Image as _Image
    tvt.functional = mod("torchvision.transforms.functional",
                         resize=lambda img, size, interpolation=_Image.BILINEAR: img.resize(tuple(size)[::-1], interpolation))
    tv.models, tv.transforms = tvm, tvt
    ra = mod("roi_align")
    ram = mod("roi_align.roi_align", RoIAlign=_RoIAlignStub)
    ra.roi_align = ram
    sk = mod("skimage")
    sk.io = mod("skimage.io")
    sk.transform = mod("skimage.transform")
    mod("cv2")


# ------------------------------------------------------------------ helpers
def seeded(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def close(a, b, tol, what):
    a, b = a.double(), b.double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-30
    assert err <= tol * ref, f"ORACLE != REFERENCE for {what}: max abs err {err:.3e} (scale {ref:.3e})"
    return err / ref


def din_case(name, refmod, b, t, n, c, kernels, ratios, beta, num_dim, dtype, out_dir,
             x_seed, w_seed, store_full=True, din_std=0.05, offset_boost=1.0):
    """Run the reference Multi_Dynamic_Inference / bare DPI fwd+bwd; compare oracle; save."""
    DPI, MDI = refmod.Dynamic_Person_Inference, refmod.Multi_Dynamic_Inference
    torch.manual_seed(0)
    m = MDI(in_dim=c, person_mat_shape=(10, 12), stride=1, kernel_size=kernels, dynamic_sampling=True,
            sampling_ratio=ratios, group=1, scale_factor=True, beta_factor=beta,
            parallel_inference=False, num_DIM=num_dim, cfg=None).to(dtype)
    shapes = {}
    for i in range(num_dim):
        shapes.update(O.din_param_shapes(f"DIMlist.{i}.", c, tuple(kernels[i]), ratios, True, beta))
    p = O.synth_params(shapes, seed=w_seed, din_std=din_std, dtype=dtype)
    for k in p:
        if "p_conv" in k:
            p[k] = p[k] * offset_boost          # push some samples past the clamp range
        if k.endswith("beta"):
            p[k] = 1.0 + 0.25 * seeded(p[k].shape, w_seed + 7, 1.0, dtype)
    missing, unexpected = m.load_state_dict(p, strict=False)
    assert not unexpected and not [k for k in missing if "zero_padding" not in k], (missing, unexpected)
    x = seeded((b, t, n, c), x_seed, 1.0, dtype).requires_grad_(True)
    cot = seeded((b, t, n, c), x_seed + 1, 1.0, dtype)
    out, mad = m(x)
    (out * cot).sum().backward()
    ref_gx = x.grad.detach().clone()
    ref_gp = {k: v.grad.detach().clone() for k, v in m.named_parameters()}

    # oracle on the same numbers
    po = {("DPI." + k): v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.detach().clone().requires_grad_(True)
    oo, omad = O.din_multi_inference(xo, po, "DPI.", kernels, ratios, True, beta)
    (oo * cot).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-11
    errs = dict(out=close(oo, out, tol, name + ".out"), mad=close(omad, mad, tol, name + ".mad"),
                gx=close(xo.grad, ref_gx, 10 * tol, name + ".gx"))
    for k, v in ref_gp.items():
        errs["g_" + k] = close(po["DPI." + k].grad, v, 10 * tol, name + ".g_" + k)

    # integer corner decisions of the LAST module/ratio, re-evaluated from the reference's own
    # conv (same expressions as dynamic_infer_module.py:191,208-223)
    last = m.DIMlist[num_dim - 1]
    r = ratios[-1]
    kh, kw = kernels[num_dim - 1]
    k2 = kh * kw
    with torch.no_grad():
        off = last.p_conv[str(r)](x.detach().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        pos = last._get_pos(off, r)
        lt = pos.floor()
        pt, pl = (kh - 1) // 2 * r, (kw - 1) // 2 * r
        ly = lt[..., :k2].clamp(0, t + 2 * pt - 1).long()
        lx = lt[..., k2:].clamp(0, n + 2 * pl - 1).long()
        ry = (lt[..., :k2] + 1).clamp(0, t + 2 * pt - 1).long()
        rx = (lt[..., k2:] + 1).clamp(0, n + 2 * pl - 1).long()
        scale = torch.softmax(last.scale_conv[str(r)](x.detach().permute(0, 3, 1, 2)).permute(0, 2, 3, 1), -1)
    frac_clamped = float(((lt[..., :k2] < 0) | (lt[..., :k2] + 1 > t + 2 * pt - 1)).double().mean())

    rec = dict(meta=np.array([b, t, n, c, num_dim, int(beta)], dtype=np.int64),
               kernels=np.array(kernels, dtype=np.int64), ratios=np.array(ratios, dtype=np.int64),
               x_seed=np.int64(x_seed), w_seed=np.int64(w_seed), din_std=np.float64(din_std),
               offset_boost=np.float64(offset_boost),
               dtype=np.array(str(dtype).replace("torch.", "")),
               out=out.detach().numpy(), gx=ref_gx.numpy(),
               ly=ly.numpy().astype(np.int32), lx=lx.numpy().astype(np.int32),
               ry=ry.numpy().astype(np.int32), rx=rx.numpy().astype(np.int32),
               offset=off.numpy(), scale=scale.numpy())
    if store_full:
        rec["x"] = x.detach().numpy()
        rec["cot"] = cot.numpy()
        rec["mad"] = mad.detach().numpy()
        for k, v in p.items():
            rec["p." + k] = v.numpy()
    else:
        rec["mad_checksum"] = np.float64(mad.detach().double().sum().item())
    for k, v in ref_gp.items():
        if store_full or v.numel() <= 65536:
            rec["g." + k] = v.numpy()
        else:
            rec["gsum." + k] = np.float64(v.double().sum().item())
            rec["gabs." + k] = np.float64(v.double().abs().sum().item())
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(f"[din] {name}: oracle-vs-reference rel err max {max(errs.values()):.2e}; "
          f"clamped-row fraction {frac_clamped:.3f}")


def din_mode_case(name, refmod, out_dir, *, mode, b, t, n, c, kernel, ratios, beta, scale=True, x_seed=71, w_seed=81, din_std=0.05,
                  offset_boost=1.0):
    """dynamic_sampling=False ('plain') and parallel_inference=True ('parallel'): the reference's forward computes the per-ratio
    features and then raises on the unbound ft_infer_MAD (dynamic_infer_module.py:151).  No-source-patch recipe: call the reference's
    OWN per-ratio methods (plain_infer_ratio :154-181 / parallel_infer :285-341) and combine them as :137-147 do."""
    DPI = refmod.Dynamic_Person_Inference
    torch.manual_seed(0)
    m = DPI(in_dim=c, person_mat_shape=(10, 12), stride=1, kernel_size=kernel, dynamic_sampling=(mode == "parallel"),
            sampling_ratio=ratios, group=1, scale_factor=scale, beta_factor=beta, parallel_inference=(mode == "parallel"), cfg=None)
    shapes = O.din_param_shapes("", c, tuple(kernel), ratios, scale, beta)
    if mode == "plain":
        shapes = {k: v for k, v in shapes.items() if "p_conv" not in k}
    p = O.synth_params(shapes, seed=w_seed, din_std=din_std)
    for k in p:
        if "p_conv" in k:
            p[k] = p[k] * offset_boost
        if k.endswith("beta"):
            p[k] = 1.0 + 0.25 * seeded(p[k].shape, w_seed + 7, 1.0)
    missing, unexpected = m.load_state_dict(p, strict=False)
    assert not unexpected and not [k for k in missing if "zero_padding" not in k], (missing, unexpected)
    x = seeded((b, t, n, c), x_seed).requires_grad_(True)
    cot = seeded((b, t, n, c), x_seed + 1)
    xc = x.permute(0, 3, 1, 2)
    feats = [(m.parallel_infer(xc, r) if mode == "parallel" else m.plain_infer_ratio(xc, r)) for r in ratios]
    st = torch.stack(feats, dim=4)                                    # :137-147
    dyn = torch.sum(m.beta * st, dim=-1) if beta else torch.mean(st, dim=4)
    out = m.hidden_weight(dyn)
    (out * cot).sum().backward()
    po = {("DPI." + k): v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.detach().clone().requires_grad_(True)
    oo, _ = O.din_person_inference(xo, po, "DPI.", tuple(kernel), ratios, scale, beta, dynamic_sampling=(mode == "parallel"),
                                   parallel_inference=(mode == "parallel"))
    (oo * cot).sum().backward()
    errs = dict(out=close(oo, out, 1e-5, name + ".out"), gx=close(xo.grad, x.grad, 1e-4, name + ".gx"))
    rec = dict(meta=np.array([b, t, n, c, int(beta), int(scale)], dtype=np.int64), kernel=np.array(kernel, dtype=np.int64),
               ratios=np.array(ratios, dtype=np.int64), mode=np.array(mode), x=x.detach().numpy(), cot=cot.numpy(),
               out=out.detach().numpy(), gx=x.grad.numpy())
    for k, v in m.named_parameters():
        errs["g_" + k] = close(po["DPI." + k].grad, v.grad, 1e-4, name + ".g_" + k)
        rec["g." + k] = v.grad.numpy()
        rec["p." + k] = p[k].numpy()
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(f"[mode] {name}: oracle-vs-reference rel err max {max(errs.values()):.2e}")


def prep_case(refutils, out_dir):
    x = torch.arange(0, 256, dtype=torch.float32)
    y = refutils.prep_images(x)
    assert torch.equal(O.prep_images(x), y), "prep_images oracle mismatch (must be bit-exact)"
    np.savez_compressed(os.path.join(out_dir, "prep_images.npz"), x=x.numpy(), y=y.numpy())
    print("[prep] 256 uint8 levels: oracle bit-exact with reference")


def model_case(name, refim, refcfg, out_dir, *, backbone, H, W, OH, OW, D, B, T, N, NFB, kernels, ratios,
               num_dim=1, beta=False, lite=None, seed=0, dtype=torch.float32, hier=False, full_grads_upto=0, refdin=None):
    """Whole Dynamic_volleyball forward (+ backward of CE loss) from the reference.  full_grads_upto: parameter gradients with at most
    that many elements are stored whole (`g.<name>`) next to the per-tensor sums.  hier=True applies the no-source-patch recipe of
    hier_case (DPI_1 wrapped to return ft, the always-on functional dropout neutralised) to the whole model."""
    cfg = refcfg.Config("volleyball")
    cfg.log_path = None
    cfg.backbone = backbone
    cfg.image_size, cfg.out_size, cfg.emb_features = (H, W), (OH, OW), D
    cfg.num_boxes, cfg.num_frames, cfg.batch_size = N, T, B
    cfg.num_features_boxes = cfg.num_features_gcn = NFB
    cfg.ST_kernel_size, cfg.sampling_ratio, cfg.num_DIM = kernels, ratios, num_dim
    cfg.dynamic_sampling, cfg.scale_factor, cfg.beta_factor = True, True, beta
    cfg.lite_dim, cfg.hierarchical_inference = lite, hier
    cfg.train_backbone = True
    cfg.train_dropout_prob = 0.3
    torch.manual_seed(0)
    model = refim.Dynamic_volleyball(cfg).to(dtype)
    model.eval()                                   # dropout off, BN running stats
    if backbone == "inv3":
        # reference has no 'inv3' head branch (infer_model.py:203-216; SURVEY section 0 bug 1):
        # oracle recipe = run the vgg16 branch (residual -> LN -> ReLU -> dropout)
        model.cfg.backbone = "vgg16"
    ocfg = O.OracleCfg(backbone=backbone, image_size=(H, W), out_size=(OH, OW), emb_features=D,
                       num_boxes=N, num_frames=T, num_features_boxes=NFB, ST_kernel_size=kernels,
                       sampling_ratio=ratios, num_DIM=num_dim, beta_factor=beta, lite_dim=lite,
                       hierarchical_inference=hier)
    shapes = O.model_param_shapes(ocfg)
    p = O.synth_params(shapes, seed=seed + 3, din_std=0.02, dtype=dtype)
    if hier:
        g_ = torch.Generator().manual_seed(seed + 11)
        p["DPI.hier_LN.weight"] = 0.75 + 0.5 * torch.rand(p["DPI.hier_LN.weight"].shape, generator=g_)
        p["DPI.hier_LN.bias"] = 0.1 * torch.randn(p["DPI.hier_LN.bias"].shape, generator=g_)
    missing, unexpected = model.load_state_dict(p, strict=False)
    bad = [k for k in missing if "num_batches_tracked" not in k and "zero_padding" not in k]
    assert not unexpected and not bad, (bad, unexpected)
    images, boxes, labels = O.synth_inputs(B, T, N, H, W, OH, OW, 8, seed=seed)
    images = images.to(dtype)
    boxes = boxes.to(dtype)
    realF = None
    if hier:
        model.DPI.DPI_1 = _First(model.DPI.DPI_1)
        realF = refdin.F
        shim = types.SimpleNamespace(**{k: getattr(realF, k) for k in dir(realF) if not k.startswith("__")})
        shim.dropout = lambda x, *a, **k: x
        refdin.F = shim
    try:
        ret = model((images, boxes))
        loss = F.cross_entropy(ret["activities"], labels)
        loss.backward()
    finally:
        if realF is not None:
            refdin.F = realF
    ref_grads = {k.replace("DPI_1.m.", "DPI_1."): v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}

    po = {k: v.clone().requires_grad_("running_" not in k) for k, v in p.items()}
    oret, inter = O.dynamic_volleyball_forward(ocfg, po, images, boxes, return_intermediates=True)
    oloss = F.cross_entropy(oret["activities"], labels)
    oloss.backward()
    tol = 2e-4 if dtype == torch.float32 else 1e-10
    e = close(oret["activities"], ret["activities"], tol, name + ".logits")
    eg = 0.0
    for k, v in ref_grads.items():
        if po[k].grad is None:
            raise AssertionError("oracle produced no grad for " + k)
        eg = max(eg, close(po[k].grad, v, 50 * tol, name + ".grad." + k))
    rec = dict(meta=np.array([B, T, N, H, W, OH, OW, D, NFB, num_dim, int(beta), lite or 0, int(hier)], dtype=np.int64),
               backbone=np.array(backbone), kernels=np.array(kernels, dtype=np.int64),
               ratios=np.array(ratios, dtype=np.int64), seed=np.int64(seed),
               dtype=np.array(str(dtype).replace("torch.", "")),
               logits=ret["activities"].detach().numpy(), loss=np.float64(loss.item()),
               labels=labels.numpy())
    for k, v in ref_grads.items():
        rec["gsum." + k] = np.float64(v.double().sum().item())
        rec["gabs." + k] = np.float64(v.double().abs().sum().item())
    for k in ("fc_activities.weight", "fc_activities.bias", "nl_emb_1.weight"):
        rec["g." + k] = ref_grads[k].numpy()
    for k, v in ref_grads.items():
        if v.numel() <= full_grads_upto and "g." + k not in rec:
            rec["g." + k] = v.numpy()
    if full_grads_upto and dtype == torch.float32 and not hier:
        # yardstick for the stored gradients (as in full_case): the same model in float64, and how far the reference's fp32 run is from it
        p64 = {k: v.double().requires_grad_("running_" not in k) for k, v in p.items()}
        o64 = O.dynamic_volleyball_forward(ocfg, p64, images.double(), boxes.double())
        F.cross_entropy(o64["activities"], labels).backward()
        for k in list(rec):
            if k.startswith("g.") and k[2:] in p64 and p64[k[2:]].grad is not None:
                g64 = p64[k[2:]].grad
                rec["g64." + k[2:]] = g64.float().numpy().copy()
                rec["yard." + k[2:]] = np.float64(float((ref_grads[k[2:]].double() - g64).abs().max() / (g64.abs().max() + 1e-300)))
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(f"[model] {name}: logits rel err {e:.2e}, worst grad rel err {eg:.2e}, loss {loss.item():.6f}")


def _probe_idx(numel, count=8192):
    """deterministic sample positions of a flat tensor (same recipe in tests/test_gpu_din_model.py)"""
    count = min(count, numel)
    return (torch.arange(count, dtype=torch.int64) * (numel // count))


def full_case(name, refim, refcfg, out_dir, *, backbone, OH, OW, D, B, seed, H=720, W=1280, T=3, N=12, NFB=1024, search=True, smooth=False):
    """ONE full-size 720x1280 run of the reference's Dynamic_volleyball (infer_model.py:141-234), fwd + backward of the
    CE loss, eval mode.  Inputs and the 29 M weights are NOT stored: both sides regenerate them from the seed recipe of
    oracle.din_oracle.synth_inputs / synth_params.  Stored: logits, loss, per-stage feature probes taken with forward hooks on the
    reference's own modules (backbone outputs, RoIAlign crops, embedding after LN+ReLU, DIN output: sums + 8192 strided samples each),
    every parameter gradient's sum / abs-sum, small gradients whole and 8192 strided samples of the big ones."""
    cfg = refcfg.Config("volleyball")
    cfg.log_path = None
    cfg.backbone = backbone
    cfg.image_size, cfg.out_size, cfg.emb_features = (H, W), (OH, OW), D
    cfg.num_boxes, cfg.num_frames, cfg.batch_size = N, T, B
    cfg.num_features_boxes = cfg.num_features_gcn = NFB
    cfg.ST_kernel_size, cfg.sampling_ratio, cfg.num_DIM = [(3, 3)], [1], 1
    cfg.dynamic_sampling, cfg.scale_factor, cfg.beta_factor = True, True, False
    cfg.lite_dim, cfg.hierarchical_inference = None, False
    cfg.train_backbone = True
    cfg.train_dropout_prob = 0.3
    torch.manual_seed(0)
    model = refim.Dynamic_volleyball(cfg)
    model.eval()
    if backbone == "inv3":
        model.cfg.backbone = "vgg16"               # same recipe as model_case: the reference has no inv3 head branch
    ocfg = O.OracleCfg(backbone=backbone, image_size=(H, W), out_size=(OH, OW), emb_features=D, num_boxes=N, num_frames=T,
                       num_features_boxes=NFB, ST_kernel_size=[(3, 3)], sampling_ratio=[1], num_DIM=1)
    # Seed choice: the head takes the maximum over the 12 actors per (clip, frame, channel) (infer_model.py:224).  With thousands of such
    # windows a random draw usually holds a few whose two largest entries agree to ~1e-6 of the activation scale; a 1e-6 forward
    # difference (any other fp32 summation order) then hands the whole gradient of that window to the other actor and every upstream
    # gradient moves by ~1e-2 -- a property of the draw, not of the implementation under test.  Take the first seed (seed, seed + 1000, ...)
    # whose smallest relative top-2 gap is >= 5e-6; the fixture records it.
    min_gap = 0.0
    for _try in range(64):
        p = O.synth_params(O.model_param_shapes(ocfg), seed=seed + 3, din_std=0.02)
        images, boxes, labels = O.synth_inputs(B, T, N, H, W, OH, OW, 8, seed=seed)
        if smooth:                             # scene-like frames instead of white noise; boxes / labels as before
            images = O.synth_scene_images(boxes, H, W, OH, OW, seed=seed + 7)
        with torch.no_grad():
            _o, inter0 = O.dynamic_volleyball_forward(ocfg, p, images.float(), boxes, return_intermediates=True)
            lw, lb = p["dpi_nl.weight"], p["dpi_nl.bias"]
            s_ = F.relu(F.layer_norm(inter0["graph"] + inter0["x"], lw.shape, lw, lb, 1e-5))
            top2 = s_.topk(2, dim=2).values
            live = top2[:, :, 0] > 0
            min_gap = float(((top2[:, :, 0] - top2[:, :, 1])[live] / s_.abs().max()).min())
            near_ties = int((((top2[:, :, 0] - top2[:, :, 1]) / s_.abs().max() < 5e-6) & live).sum())
        print(f"[full] {name}: seed {seed}: smallest relative top-2 gap of the actor max {min_gap:.2e} ({near_ties} windows under 5e-6)")
        if min_gap >= 5e-6 or not search:      # search=False: the FIRST draw, near-ties and all (one fixture that was not chosen to be easy)
            break
        seed += 1000
    assert min_gap >= 5e-6 or not search
    del inter0, s_, top2
    missing, unexpected = model.load_state_dict(p, strict=False)
    bad = [k for k in missing if "num_batches_tracked" not in k and "zero_padding" not in k]
    assert not unexpected and not bad, (bad, unexpected)
    probes = {}

    def probe(key, t):
        t = t.detach()
        flat = t.reshape(-1)
        probes["feat." + key + ".shape"] = np.array(t.shape, dtype=np.int64)
        probes["feat." + key + ".sum"] = np.float64(flat.double().sum().item())
        probes["feat." + key + ".abs"] = np.float64(flat.double().abs().sum().item())
        probes["feat." + key + ".max"] = np.float64(flat.abs().max().item())
        probes["feat." + key + ".sample"] = flat[_probe_idx(flat.numel())].clone().numpy()

    def hook(fn):
        def run(_m, _i, o):
            fn(o)                                   # (a forward hook's return value would replace the module output: return None)
        return run

    hooks = [model.backbone.register_forward_hook(hook(lambda o: [probe(f"fm{j}", t) for j, t in enumerate(o)])),
             model.roi_align.register_forward_hook(hook(lambda o: probe("crops", o))),
             model.nl_emb_1.register_forward_hook(hook(lambda o: probe("x_emb", torch.relu(o)))),     # (:185-186; the ReLU is in place)
             model.DPI.register_forward_hook(hook(lambda o: probe("graph", o[0])))]
    t0 = time.time()
    ret = model((images.float(), boxes))
    loss = F.cross_entropy(ret["activities"], labels)
    loss.backward()
    for h_ in hooks:
        h_.remove()
    t_ref = time.time() - t0
    ref_grads = {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}
    ref_logits = ret["activities"].detach().clone()
    del model, ret
    # the oracle restatement on the same seeds must agree before the fixture is written
    po = {k: v.clone().requires_grad_("running_" not in k) for k, v in p.items()}
    oret, inter = O.dynamic_volleyball_forward(ocfg, po, images.float(), boxes, return_intermediates=True)
    oloss = F.cross_entropy(oret["activities"], labels)
    oloss.backward()
    e = close(oret["activities"], ref_logits, 2e-4, name + ".logits")
    for key, t in (("crops", inter["crops"]), ("x_emb", inter["x"]), ("graph", inter["graph"])):
        got = t.detach().reshape(-1)[_probe_idx(t.numel())]
        close(got, torch.from_numpy(probes["feat." + key + ".sample"]), 2e-4, name + ".feat." + key)
    eg = 0.0
    for k, v in ref_grads.items():
        if search:
            eg = max(eg, close(po[k].grad, v, 1e-2, name + ".grad." + k))
        else:                                  # an un-searched draw may route a near-tied actor-max window differently in the oracle's own fp32 run
            eg = max(eg, float((po[k].grad - v).abs().max() / (v.abs().max() + 1e-30)))
    # Yardstick for the gradient comparisons: how far the reference's OWN fp32 run is from exact arithmetic.  Below a max-pool / ReLU a
    # 1e-7 perturbation of an activation can re-route a gradient element (near-tied pool windows, pre-activations at zero), so the fp32
    # reference differs from its fp64 self by up to a few 1e-2 in the first layers; any other fp32 summation order differs from the
    # reference by about as much.  The same model in float64 (oracle = the reference's arithmetic, checked above) gives that distance
    # per tensor; the GPU test allows a small multiple of it.
    del po, oret, inter, oloss
    p64 = {k: v.double().requires_grad_("running_" not in k) for k, v in p.items()}
    o64 = O.dynamic_volleyball_forward(ocfg, p64, images.double(), boxes.double())
    F.cross_entropy(o64["activities"], labels).backward()
    yard = {k: float((v.double() - p64[k].grad).abs().max() / (p64[k].grad.abs().max() + 1e-300)) for k, v in ref_grads.items()}
    yard_logits = float((ref_logits.double() - o64["activities"].detach()).abs().max() / o64["activities"].detach().abs().max())
    # the float64 gradients themselves (rounded to fp32 for storage), at the same positions as the reference's records below: lets the GPU
    # test say which side of a disagreement is the one that left exact arithmetic
    g64 = {}
    for k, v in ref_grads.items():
        t = p64[k].grad
        g64[("g64." if v.numel() <= 9216 else "gs64.") + k] = (t if v.numel() <= 9216 else t.reshape(-1)[_probe_idx(t.numel())]).float().numpy().copy()
    del p64, o64
    rec = dict(meta=np.array([B, T, N, H, W, OH, OW, D, NFB, 1, 0, 0, 0], dtype=np.int64), backbone=np.array(backbone),
               kernels=np.array([(3, 3)], dtype=np.int64), ratios=np.array([1], dtype=np.int64), seed=np.int64(seed),
               dtype=np.array("float32"), logits=ref_logits.numpy(), loss=np.float64(loss.item()), labels=labels.numpy(),
               ref_seconds_fwd_bwd=np.float64(t_ref), ref_threads=np.int64(torch.get_num_threads()),
               yard_logits=np.float64(yard_logits), min_actor_gap=np.float64(min_gap), near_ties=np.int64(near_ties), searched=np.int64(1 if search else 0),
               oracle_vs_ref_worst_grad=np.float64(eg), smooth=np.int64(1 if smooth else 0))
    for k, v in yard.items():
        rec["yard." + k] = np.float64(v)
    rec.update(g64)
    rec.update(probes)
    for k, v in ref_grads.items():
        rec["gsum." + k] = np.float64(v.double().sum().item())
        rec["gabs." + k] = np.float64(v.double().abs().sum().item())
        if v.numel() <= 9216:
            rec["g." + k] = v.numpy()
        else:
            rec["gs." + k] = v.reshape(-1)[_probe_idx(v.numel())].clone().numpy()
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    worst = sorted(yard.items(), key=lambda kv: -kv[1])[:3]
    print(f"[full] {name}: reference fwd+bwd {t_ref:.1f} s on {torch.get_num_threads()} threads; oracle logits rel err {e:.2e}, "
          f"worst grad rel err {eg:.2e}, loss {loss.item():.6f}; fp32 reference vs fp64: logits {yard_logits:.1e}, worst gradients {worst}")


def tce_case(name, refim, refcfg, out_dir, *, H, W, OH, OW, B, T, NFB, kernels, ratios, num_dim=1, seed=0, full_grads_upto=4096, hier=False,
             refdin=None):
    """Whole Dynamic_TCE_volleyball forward (+ backward of the CE loss) from the reference (infer_model.py:237-468), vgg16 trunk, eval mode
    (dropout off).  N = 12 is asserted by the reference's transformer (TCE_STBiP_module.py:263)."""
    N, D = 12, 512
    cfg = refcfg.Config("volleyball")
    cfg.log_path = None
    cfg.backbone = "vgg16"
    cfg.image_size, cfg.out_size, cfg.emb_features = (H, W), (OH, OW), D
    cfg.num_boxes, cfg.num_frames, cfg.batch_size = N, T, B
    cfg.num_features_boxes = cfg.num_features_gcn = NFB
    cfg.ST_kernel_size, cfg.sampling_ratio, cfg.num_DIM = kernels, ratios, num_dim
    cfg.dynamic_sampling, cfg.scale_factor, cfg.beta_factor = True, True, False
    cfg.lite_dim, cfg.hierarchical_inference = None, hier
    cfg.train_backbone = True
    cfg.train_dropout_prob = 0.3
    torch.manual_seed(0)
    model = refim.Dynamic_TCE_volleyball(cfg)
    model.eval()
    ocfg = O.OracleCfg(backbone="vgg16", image_size=(H, W), out_size=(OH, OW), emb_features=D, num_boxes=N, num_frames=T,
                       num_features_boxes=NFB, ST_kernel_size=kernels, sampling_ratio=ratios, num_DIM=num_dim, hierarchical_inference=hier)
    p = O.tce_synth_params(ocfg, seed)
    if hier:
        g_ = torch.Generator().manual_seed(seed + 11)
        p["DPI.hier_LN.weight"] = 0.75 + 0.5 * torch.rand(p["DPI.hier_LN.weight"].shape, generator=g_)
        p["DPI.hier_LN.bias"] = 0.1 * torch.randn(p["DPI.hier_LN.bias"].shape, generator=g_)
    missing, unexpected = model.load_state_dict(p, strict=False)
    bad = [k for k in missing if "num_batches_tracked" not in k and "zero_padding" not in k]
    assert not unexpected and not bad, (bad, unexpected)
    images, boxes, labels = O.synth_inputs(B, T, N, H, W, OH, OW, 8, seed=seed)
    realF = None
    if hier:                                             # the no-source-patch recipe of hier_case / model_case (reference bug 2, always-on dropout)
        model.DPI.DPI_1 = _First(model.DPI.DPI_1)
        realF = refdin.F
        shim = types.SimpleNamespace(**{k: getattr(realF, k) for k in dir(realF) if not k.startswith("__")})
        shim.dropout = lambda x, *a, **k: x
        refdin.F = shim
    try:
        ret = model((images.float(), boxes.float()))
        loss = F.cross_entropy(ret["activities"], labels)
        loss.backward()
    finally:
        if realF is not None:
            refdin.F = realF
    ref_grads = {k.replace("DPI_1.m.", "DPI_1."): v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}
    po = {k: v.clone().requires_grad_("running_" not in k) for k, v in p.items()}
    oret, inter = O.dynamic_tce_volleyball_forward(ocfg, po, images.float(), boxes.float(), return_intermediates=True)
    oloss = F.cross_entropy(oret["activities"], labels)
    oloss.backward()
    e = close(oret["activities"], ret["activities"], 2e-4, name + ".logits")
    eg = 0.0
    for k, v in ref_grads.items():
        if po[k].grad is None:
            raise AssertionError("oracle produced no grad for " + k)
        eg = max(eg, close(po[k].grad, v, 1e-2, name + ".grad." + k))
    att = model.multilayer_head_embfeature_context_encoding.CET[TCE_PROBE_HEAD].att_map.detach()     # [BT,N,P] of one head
    _, oatt = O.tce_context_encoding(inter["x"].reshape(B * T * N, -1), inter["context"], po, return_attention=True)
    close(oatt[TCE_PROBE_HEAD], att, 1e-4, name + ".att_map")
    rec = dict(meta=np.array([B, T, N, H, W, OH, OW, D, NFB, num_dim, int(hier)], dtype=np.int64), kernels=np.array(kernels, dtype=np.int64),
               ratios=np.array(ratios, dtype=np.int64), seed=np.int64(seed), logits=ret["activities"].detach().numpy(),
               loss=np.float64(loss.item()), labels=labels.numpy(), att_map=att.numpy(), att_head=np.int64(TCE_PROBE_HEAD),
               enc=inter["enc"].detach().numpy())
    for k, v in ref_grads.items():
        rec["gsum." + k] = np.float64(v.double().sum().item())
        rec["gabs." + k] = np.float64(v.double().abs().sum().item())
        if v.numel() <= full_grads_upto or ("context_encoding" in k and v.numel() <= 20000):
            rec["g." + k] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(f"[tce] {name}: logits rel err {e:.2e}, worst grad rel err {eg:.2e}, loss {loss.item():.6f}")


TCE_PROBE_HEAD = 2


class _First(nn.Module):
    """SURVEY 8c oracle recipe: the reference feeds DPI's (ft, MAD) tuple where a tensor is expected
    (dynamic_infer_module.py:492-493, infer_model.py:1294); wrap the sub-module so only `ft` flows on."""
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, x):
        return self.m(x)[0]


def hier_case(name, refmod, out_dir, x_seed=51, w_seed=61):
    """Hierarchical_Dynamic_Inference (dynamic_infer_module.py:446-498) with the no-source-patch recipe:
    DPI_1 wrapped to return ft; the always-on functional dropout (:495) neutralised by rebinding the module-global `F`."""
    T, N, C = 10, 12, 1024                      # forced by hier_LN = LayerNorm(person_mat_shape + (1024,))
    kernels, ratios = [(1, 3), (3, 1)], [1]
    torch.manual_seed(0)
    H = refmod.Hierarchical_Dynamic_Inference(in_dim=C, person_mat_shape=(T, N), kernel_size=kernels, dynamic_sampling=True,
                                              sampling_ratio=ratios, scale_factor=True, beta_factor=False, cfg=None)
    shapes = {}
    for i, sub in enumerate(("DPI_1.", "DPI_2.")):
        shapes.update(O.din_param_shapes(sub, C, kernels[i], ratios, True, False))
    shapes["hier_LN.weight"] = (T, N, C)
    shapes["hier_LN.bias"] = (T, N, C)
    p = O.synth_params(shapes, seed=w_seed, din_std=0.02)
    g = torch.Generator().manual_seed(w_seed + 1)
    p["hier_LN.weight"] = 0.75 + 0.5 * torch.rand((T, N, C), generator=g)
    p["hier_LN.bias"] = 0.1 * torch.randn((T, N, C), generator=g)
    missing, unexpected = H.load_state_dict(p, strict=False)
    assert not unexpected and not [k for k in missing if "zero_padding" not in k], (missing, unexpected)
    H.DPI_1 = _First(H.DPI_1)
    realF = refmod.F
    shim = types.SimpleNamespace(**{k: getattr(realF, k) for k in dir(realF) if not k.startswith("__")})
    shim.dropout = lambda x, *a, **k: x
    refmod.F = shim
    try:
        x = seeded((1, T, N, C), x_seed).requires_grad_(True)
        cot = seeded((1, T, N, C), x_seed + 1)
        out, _mad = H(x)
        (out * cot).sum().backward()
    finally:
        refmod.F = realF
    po = {("DPI." + k): v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.detach().clone().requires_grad_(True)
    oo, _ = O.din_hierarchical_inference(xo, po, "DPI.", kernels, ratios, True, False)
    (oo * cot).sum().backward()
    e = close(oo, out, 1e-5, name + ".out")
    eg = close(xo.grad, x.grad, 1e-4, name + ".gx")
    rec = dict(out=out.detach().numpy(), gx=x.grad.numpy(), x_seed=np.int64(x_seed), w_seed=np.int64(w_seed))
    for k, v in H.named_parameters():
        kk = k.replace("DPI_1.m.", "DPI_1.")
        eg = max(eg, close(po["DPI." + kk].grad, v.grad, 1e-4, name + ".g_" + kk))
        rec["gsum." + kk] = np.float64(v.grad.double().sum().item())
        rec["gabs." + kk] = np.float64(v.grad.double().abs().sum().item())
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(f"[hier] {name}: oracle-vs-reference out {e:.2e}, worst grad {eg:.2e}")


def collective_case(name, refim, refcfg, out_dir, seed=200):
    """Dynamic_collective.forward (infer_model.py:1226-1319), variable actors per clip, DPI wrapped per the recipe."""
    H_, W_, OH, OW, B, T, MAXN, NFB, A = 96, 160, 3, 5, 3, 3, 6, 64, 4
    cfg = refcfg.Config("collective")
    cfg.log_path = None
    cfg.backbone, cfg.image_size, cfg.out_size, cfg.emb_features = "vgg16", (H_, W_), (OH, OW), 512
    cfg.num_boxes, cfg.num_frames, cfg.batch_size, cfg.num_activities = MAXN, T, B, A
    cfg.num_features_boxes = cfg.num_features_gcn = NFB
    cfg.ST_kernel_size, cfg.sampling_ratio = (3, 3), [1]
    cfg.dynamic_sampling, cfg.scale_factor, cfg.beta_factor = True, True, False
    cfg.lite_dim, cfg.hierarchical_inference, cfg.train_backbone = None, False, True
    torch.manual_seed(0)
    model = refim.Dynamic_collective(cfg)
    model.eval()
    ocfg = O.OracleCfg(image_size=(H_, W_), out_size=(OH, OW), num_boxes=MAXN, num_frames=T, num_features_boxes=NFB,
                       ST_kernel_size=(3, 3), sampling_ratio=[1], num_activities=A, collective=True)
    p = O.synth_params(O.model_param_shapes(ocfg), seed=seed + 3, din_std=0.05)
    g = torch.Generator().manual_seed(seed + 5)
    p["dpi_nl.weight"] = 0.75 + 0.5 * torch.rand((T, NFB), generator=g)
    p["dpi_nl.bias"] = 0.1 * torch.randn((T, NFB), generator=g)
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected and not [k for k in missing if "zero_padding" not in k], (missing, unexpected)
    model.DPI = _First(model.DPI)
    images, boxes, labels = O.synth_inputs(B, T, MAXN, H_, W_, OH, OW, A, seed=seed)
    counts = torch.tensor([[6] * T, [1] * T, [4] * T], dtype=torch.int32)      # variable N incl. a single-actor clip
    for b in range(B):
        boxes[b, :, int(counts[b, 0]):] = 0.0                                  # zero padding boxes (collective.py:201-203)
    ret = model((images.float(), boxes, counts))
    loss = F.cross_entropy(ret["activities"], labels)
    loss.backward()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    oret = O.dynamic_collective_forward(ocfg, po, images.float(), boxes, counts)
    F.cross_entropy(oret["activities"], labels).backward()
    e = close(oret["activities"], ret["activities"], 2e-4, name + ".logits")
    rec = dict(logits=ret["activities"].detach().numpy(), loss=np.float64(loss.item()), labels=labels.numpy(),
               counts=counts.numpy(), seed=np.int64(seed))
    eg = 0.0
    for k, v in model.named_parameters():
        kk = k.replace("DPI.m.", "DPI.")
        eg = max(eg, close(po[kk].grad, v.grad, 2e-3, name + ".g_" + kk))
        rec["gsum." + kk] = np.float64(v.grad.double().sum().item())
        rec["gabs." + kk] = np.float64(v.grad.double().abs().sum().item())
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print(f"[collective] {name}: logits {e:.2e}, worst grad {eg:.2e}, loss {loss.item():.6f}")


def _synthetic_jpeg(path, h, w, rng):
    """a smooth random field + noise, saved as JPEG (fixture input: committed under tests/golden/dataset_tree/)"""
    from PIL import Image
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 110 * np.sin(yy / rng.uniform(5, 17) + rng.uniform(0, 6)) * np.cos(xx / rng.uniform(5, 23) + rng.uniform(0, 6))
                    for _ in range(3)], -1) + rng.normal(0, 9, (h, w, 3))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(path, quality=88)


def dataset_cases(out_dir):
    """the dataset -> tensor contract, pinned by the reference's OWN VolleyballDataset / CollectiveDataset
    (volleyball.py:146-275, collective.py:96-225) run over a small synthetic annotation tree.  The tree (annotation files, the tracks
    pickle, tiny JPEGs) is written under <out>/dataset_tree and committed as fixture INPUT; the tensors the reference produces from it go
    to dataset_volleyball.npz / dataset_collective.npz.  np.float (removed in numpy 1.24; collective.py:212 uses it) is aliased to the
    builtin float it always was, in this harness only."""
    import importlib
    import pickle
    if not hasattr(np, "float"):
        np.float = float
    refv = importlib.import_module("volleyball")
    refc = importlib.import_module("collective")
    tree = os.path.join(out_dir, "dataset_tree")
    rng = np.random.default_rng(2024)
    # ---------------- volleyball: <root>/<sid>/annotations.txt, <root>/<sid>/<src>/<fid>.jpg, <root>/tracks_normalized.pkl
    vroot = os.path.join(tree, "volleyball")
    people = {(1, 10): 12, (1, 20): 10, (4, 30): 7}                 # 12: no padding; 10 and 7 (>= 6): padded by repetition
    before, after = 1, 1
    tracks = {}
    lines = {}
    for (sid, src), n in people.items():
        acts = [refv.ACTIONS[int(a)] for a in rng.integers(0, 9, n)]
        xywh = np.stack([rng.integers(0, 1100, n), rng.integers(100, 500, n), rng.integers(30, 120, n), rng.integers(80, 220, n)], 1)
        row = "%d.jpg %s " % (src, refv.ACTIVITIES[int(rng.integers(0, 8))]) + " ".join(
            "%d %d %d %d %s" % (*xywh[i], acts[i]) for i in range(n))
        lines.setdefault(sid, []).append(row)
        tracks[(sid, src)] = {}
        base = np.stack([rng.uniform(0.25, 0.6, n), rng.uniform(0.02, 0.9, n)], 1)
        for fid in range(src - before, src + after + 1):
            yx = base + rng.normal(0, 0.004, (n, 2))
            hw = np.stack([rng.uniform(0.15, 0.33, n), rng.uniform(0.03, 0.08, n)], 1)
            tracks[(sid, src)][fid] = np.concatenate([yx, np.minimum(yx + hw, 1.0)], 1)          # (y1, x1, y2, x2), float64
            _synthetic_jpeg(os.path.join(vroot, str(sid), str(src), "%d.jpg" % fid), 72, 128, rng)
    for sid, rows in lines.items():
        with open(os.path.join(vroot, str(sid), "annotations.txt"), "w") as fh:
            fh.write("".join(r + "\n" for r in rows))
    with open(os.path.join(vroot, "tracks_normalized.pkl"), "wb") as fh:
        pickle.dump(tracks, fh, protocol=2)
    anns = refv.volley_read_dataset(vroot, [1, 4])
    frames = refv.volley_all_frames(anns)
    rec = {"frames": np.array(frames, dtype=np.int64)}
    for tag, fsize in (("vgg", (2, 3)), ("inv3", (87, 157))):
        ds = refv.VolleyballDataset(anns, tracks, frames, vroot, (64, 96), fsize, "dynamic_volleyball", num_boxes=12, num_before=before,
                                    num_after=after, is_training=True, is_finetune=False)
        for i in range(len(ds)):
            images, bboxes, actions, activities = ds[i]
            assert images.dtype == torch.float32 and torch.equal(images, images.round()) and images.min() >= 0 and images.max() <= 255
            if tag == "vgg":
                rec[f"images.{i}"] = images.to(torch.uint8).numpy()
                rec[f"actions.{i}"], rec[f"activities.{i}"] = actions.numpy(), activities.numpy()
            rec[f"boxes.{tag}.{i}"] = bboxes.numpy()
    for sid in anns:
        for fid, a in anns[sid].items():
            rec[f"ann.{sid}.{fid}.bboxes"] = np.asarray(a["bboxes"])
            rec[f"ann.{sid}.{fid}.actions"] = np.asarray(a["actions"])
            rec[f"ann.{sid}.{fid}.group_activity"] = np.int64(a["group_activity"])
    np.savez_compressed(os.path.join(out_dir, "dataset_volleyball.npz"), **rec)
    print(f"[dataset] volleyball: {len(frames)} clips of {before + after + 1} frames, people {sorted(people.values())}")
    # ---------------- collective: <root>/seq%02d/annotations.txt (tab separated), <root>/seq%02d/frame%04d.jpg
    croot = os.path.join(tree, "collective")
    nf = 3
    plan = {1: {1: 3, 11: 13}, 15: {1: 6, 11: 5}}                   # sid -> anchor frame -> people (13 = num_boxes: no padding); seq15 is 450x800
    for sid, anchors in plan.items():
        H_, W_ = refc.FRAMES_SIZE[sid]
        rows = []
        for fid in range(1, 14):
            n = anchors.get(fid, int(rng.integers(1, 6)))
            if (sid, fid) == (15, 11):
                acts = [1, 1, 1, 5, 5]                                # 'NA' is the most common: the activity is the runner-up (Walking)
            else:
                acts = [int(a) for a in rng.integers(1, 7, n)]
                if all(a == 1 for a in acts):
                    acts[0] = 3
            for j in range(n):
                w_, h_ = int(rng.integers(20, 90)), int(rng.integers(60, 200))
                x_, y_ = int(rng.integers(0, W_ - w_)), int(rng.integers(0, H_ - h_))
                rows.append("%d\t%d\t%d\t%d\t%d\t%d\t%d" % (fid, x_, y_, w_, h_, acts[j], 1))
            if fid in (1, 2, 3, 11, 12, 13):
                _synthetic_jpeg(os.path.join(croot, "seq%02d" % sid, "frame%04d.jpg" % fid), 60, 90, rng)
        with open(os.path.join(croot, "seq%02d" % sid, "annotations.txt"), "w") as fh:
            fh.write("".join(r + "\n" for r in rows))
    canns = refc.collective_read_dataset(croot, [1, 15])
    cframes = refc.collective_all_frames(canns)
    rec = {"frames": np.array(cframes, dtype=np.int64)}
    ds = refc.CollectiveDataset(canns, cframes, croot, (64, 96), (2, 3), num_boxes=13, num_frames=nf, is_training=True, is_finetune=False)
    for i in range(len(ds)):
        images, bboxes, actions, activities, bboxes_num = ds[i]
        assert torch.equal(images, images.round())
        rec[f"images.{i}"], rec[f"boxes.{i}"] = images.to(torch.uint8).numpy(), bboxes.numpy()
        rec[f"actions.{i}"], rec[f"activities.{i}"], rec[f"bboxes_num.{i}"] = actions.numpy(), activities.numpy(), bboxes_num.numpy()
    for sid in canns:
        for fid, a in canns[sid].items():
            rec[f"ann.{sid}.{fid}.bboxes"] = np.asarray(a["bboxes"], dtype=np.float64)
            rec[f"ann.{sid}.{fid}.actions"] = np.asarray(a["actions"])
            rec[f"ann.{sid}.{fid}.group_activity"] = np.int64(a["group_activity"])
    np.savez_compressed(os.path.join(out_dir, "dataset_collective.npz"), **rec)
    print(f"[dataset] collective: {len(cframes)} clips of {nf} frames, anchors {cframes}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--skip-big", action="store_true")
    ap.add_argument("--only", default="", help="'inv3': only the two reduced-size Inception fixtures; 'tce': only (re)generate the Dynamic_TCE_volleyball fixtures; 'full': only the two full-size 720x1280 fixtures; 'full_unsearched': the un-searched full-size Inception draw; 'full_scene': the full-size Inception fixture on scene-like frames; 'dataset': only the dataset -> tensor contract fixtures (SURVEY 8f-1)")
    a = ap.parse_args()
    sys.dont_write_bytecode = True
    install_stubs()
    sys.path.insert(0, a.ref)
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    import importlib
    refutils = importlib.import_module("utils")
    refdin = importlib.import_module("infer_module.dynamic_infer_module")
    refim = importlib.import_module("infer_model")
    refcfg = importlib.import_module("config")

    def tce_cases():
        # Dynamic_TCE_volleyball: OH x OW follows the VGG16 arithmetic (H/32, W/32), 15 and 24 context pixels per frame
        tce_case("tce_vgg16_96x160_nfb64", refim, refcfg, a.out, H=96, W=160, OH=3, OW=5, B=2, T=3, NFB=64, kernels=[(3, 3)], ratios=[1], seed=300)
        tce_case("tce_vgg16_128x192_nfb128_2dim", refim, refcfg, a.out, H=128, W=192, OH=4, OW=6, B=1, T=4, NFB=128, kernels=[(1, 3), (3, 1)],
                 ratios=[1], num_dim=2, seed=301)
        # (the reference hard-wires hier_LN to 1024 channels, dynamic_infer_module.py:483: NFB + 4 * 128 = 1024 -> NFB = 512 is its only shape)
        tce_case("tce_vgg16_64x96_hier_t10", refim, refcfg, a.out, H=64, W=96, OH=2, OW=3, B=1, T=10, NFB=512, kernels=[(1, 3), (3, 1)],
                 ratios=[1], seed=302, hier=True, refdin=refdin)
    def full_cases():
        # SURVEY 8(c)-(v) / BASELINE configs[1] and configs[0] at their real frame size (720x1280): the planner picks kernels here that
        # no reduced-size fixture sees together (stem halo tiles, mid-network halo, pipelined wgrad, sibling pacing)
        full_case("full_inv3_720x1280_b1", refim, refcfg, a.out, backbone="inv3", OH=87, OW=157, D=1056, B=1, seed=400)
        full_case("full_vgg16_720x1280_cfg1_b2", refim, refcfg, a.out, backbone="vgg16", OH=22, OW=40, D=512, B=2, seed=401)
    if a.only == "inv3":
        f32 = torch.float32
        model_case("model_inv3_139x203_nfb64", refim, refcfg, a.out, backbone="inv3", H=139, W=203, OH=15, OW=23,
                   D=1056, B=1, T=3, N=6, NFB=64, kernels=[(3, 3)], ratios=[1], seed=104, full_grads_upto=9216)
        model_case("model_inv3_139x203_lite128", refim, refcfg, a.out, backbone="inv3", H=139, W=203, OH=15, OW=23,
                   D=1056, B=1, T=3, N=6, NFB=256, kernels=[(3, 3)], ratios=[1], lite=128, seed=105, full_grads_upto=9216)
        return
    if a.only == "tce":
        tce_cases()
        return
    if a.only == "full":
        full_cases()
        return
    if a.only == "full_unsearched":
        # the first draw of another seed, NOT searched for a tie-free actor max (the searched fixtures above skip such draws)
        full_case("full_inv3_720x1280_b1_seed401_unsearched", refim, refcfg, a.out, backbone="inv3", OH=87, OW=157, D=1056, B=1, seed=401, search=False)
        return
    if a.only == "full_scene":
        # the same model on scene-like frames (smooth drifting background, a distinct textured actor in every box, sensor noise): the input
        # distribution the bf16 mode's gradient-direction floors are held on (white-noise frames are the worst case for the image layer's
        # weight gradient; a smooth field WITHOUT actors is the worst case for the head: every crop looks alike and the actor max ties)
        full_case("full_inv3_720x1280_b1_scene", refim, refcfg, a.out, backbone="inv3", OH=87, OW=157, D=1056, B=1, seed=402, smooth=True)
        return
    if a.only == "dataset":
        dataset_cases(a.out)
        return
    prep_case(refutils, a.out)
    f32, f64 = torch.float32, torch.float64
    # small, fully stored DIN cases (C=32) -- every structural variant that runs as shipped
    din_case("din_k33_c32_f64", refdin, 2, 3, 12, 32, [(3, 3)], [1], False, 1, f64, a.out, 11, 21, offset_boost=8.0)
    din_case("din_k33_c32", refdin, 2, 3, 12, 32, [(3, 3)], [1], False, 1, f32, a.out, 11, 21, offset_boost=8.0)
    din_case("din_k13_c32", refdin, 2, 3, 12, 32, [(1, 3)], [1], False, 1, f32, a.out, 12, 22, offset_boost=8.0)
    din_case("din_k31_c32", refdin, 2, 3, 12, 32, [(3, 1)], [1], False, 1, f32, a.out, 13, 23, offset_boost=8.0)
    din_case("din_k33_r13_beta_c32", refdin, 2, 4, 12, 32, [(3, 3)], [1, 3], True, 1, f32, a.out, 14, 24, offset_boost=8.0)
    din_case("din_2dim_k13_k31_c32", refdin, 2, 10, 12, 32, [(1, 3), (3, 1)], [1], False, 2, f32, a.out, 15, 25, offset_boost=8.0)
    din_case("din_k33_zero_init_c32", refdin, 1, 3, 5, 32, [(3, 3)], [1], False, 1, f32, a.out, 16, 26, din_std=0.0)
    din_case("din_k33_lite128", refdin, 2, 3, 12, 128, [(3, 3)], [1], False, 1, f32, a.out, 17, 27, offset_boost=4.0)
    din_case("din_k33_n1_c32", refdin, 1, 3, 1, 32, [(3, 3)], [1], False, 1, f32, a.out, 18, 28, offset_boost=8.0)
    din_case("din_k55_c32", refdin, 1, 5, 7, 32, [(5, 5)], [1], False, 1, f32, a.out, 19, 29, offset_boost=8.0)
    if not a.skip_big:
        # config-1 shape and the authors' smoke shape (inputs/weights from seeds, outputs stored)
        din_case("din_k33_c1024_cfg1", refdin, 2, 3, 12, 1024, [(3, 3)], [1], False, 1, f32, a.out, 31, 41,
                 store_full=False, din_std=0.02, offset_boost=1.0)
        din_case("din_k33_c1024_smoke_t10", refdin, 1, 10, 12, 1024, [(3, 3)], [1], False, 1, f32, a.out, 32, 42,
                 store_full=False, din_std=0.02, offset_boost=1.0)

    # dynamic_sampling=False (plain_infer_ratio) and parallel_inference=True (parallel_infer): SURVEY 8(f)-4
    din_mode_case("mode_plain_k33_r12_beta_c32", refdin, a.out, mode="plain", b=2, t=4, n=12, c=32, kernel=(3, 3), ratios=[1, 2], beta=True)
    din_mode_case("mode_plain_k13_noscale_c32", refdin, a.out, mode="plain", b=1, t=3, n=7, c=32, kernel=(1, 3), ratios=[1], beta=False, scale=False)
    din_mode_case("mode_parallel_k33_t10_c32", refdin, a.out, mode="parallel", b=2, t=10, n=12, c=32, kernel=(3, 3), ratios=[1, 3], beta=True,
                  offset_boost=8.0)
    # (parallel_infer clamps with person_mat_shape = (10, 12): on any other grid the reference gathers outside its padded map --
    #  "index 73 is out of bounds for dimension 1 with size 70" at T = 3 -- so T = 10, N = 12 is its only valid shape)
    # whole-network fixtures at reduced image sizes (reference wiring: trunk + head)
    model_case("model_vgg16_96x160_nfb64", refim, refcfg, a.out, backbone="vgg16", H=96, W=160, OH=3, OW=5, D=512,
               B=2, T=3, N=12, NFB=64, kernels=[(3, 3)], ratios=[1], seed=100)
    model_case("model_vgg16_96x160_lite", refim, refcfg, a.out, backbone="vgg16", H=96, W=160, OH=3, OW=5, D=512,
               B=2, T=3, N=12, NFB=64, kernels=[(3, 3)], ratios=[1], lite=32, seed=101)
    model_case("model_vgg16_96x160_2dim", refim, refcfg, a.out, backbone="vgg16", H=96, W=160, OH=3, OW=5, D=512,
               B=1, T=4, N=6, NFB=64, kernels=[(1, 3), (3, 1)], ratios=[1], num_dim=2, seed=102)
    if not a.skip_big:
        model_case("model_vgg16_192x320_nfb1024", refim, refcfg, a.out, backbone="vgg16", H=192, W=320, OH=6, OW=10,
                   D=512, B=2, T=3, N=12, NFB=1024, kernels=[(3, 3)], ratios=[1], seed=103)
        # Inception-v3: 299x299-ish small frame; OHxOW follows the layer arithmetic
        model_case("model_inv3_139x203_nfb64", refim, refcfg, a.out, backbone="inv3", H=139, W=203, OH=15, OW=23,
                   D=1056, B=1, T=3, N=6, NFB=64, kernels=[(3, 3)], ratios=[1], seed=104, full_grads_upto=9216)
        # BASELINE configs[2]: lite-DIN on Inception-v3 (lite_dim=128)
        model_case("model_inv3_139x203_lite128", refim, refcfg, a.out, backbone="inv3", H=139, W=203, OH=15, OW=23,
                   D=1056, B=1, T=3, N=6, NFB=256, kernels=[(3, 3)], ratios=[1], lite=128, seed=105, full_grads_upto=9216)
        # BASELINE configs[3]: ST-factorised hierarchical DIN, whole model; T=10, N=12, NFB=1024 are forced by the reference's
        # hier_LN = LayerNorm((10, 12, 1024)) (dynamic_infer_module.py:476, infer_model.py:88-101)
        model_case("model_vgg16_64x96_hier_t10", refim, refcfg, a.out, backbone="vgg16", H=64, W=96, OH=2, OW=3, D=512,
                   B=1, T=10, N=12, NFB=1024, kernels=[(1, 3), (3, 1)], ratios=[1], hier=True, seed=106, refdin=refdin)
    if not a.skip_big:
        hier_case("hier_k13_k31_t10_c1024", refdin, a.out)
    collective_case("collective_vgg16_96x160", refim, refcfg, a.out)
    tce_cases()
    dataset_cases(a.out)
    if not a.skip_big:
        full_cases()
    print("golden vectors written to", a.out)


if __name__ == "__main__":
    main()
