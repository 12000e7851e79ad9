"""CPU: the oracle (oracle/din_oracle.py) reproduces the golden vectors captured from the imported reference
(tools/gen_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP path to the oracle."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import din_oracle as O

TORCH_DT = {"float32": torch.float32, "float64": torch.float64}


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def seeded(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale).to(dtype)


def load_din_case(path):
    z = np.load(path)
    b, t, n, c, num_dim, beta = [int(v) for v in z["meta"]]
    kernels = [tuple(int(x) for x in k) for k in z["kernels"]]
    ratios = [int(r) for r in z["ratios"]]
    dt = TORCH_DT[str(z["dtype"])]
    if "x" in z.files:
        x = torch.from_numpy(z["x"])
        cot = torch.from_numpy(z["cot"])
        p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    else:   # big cases: regenerate inputs / weights from the recorded seeds (same recipe as tools/gen_golden.py)
        x_seed, w_seed = int(z["x_seed"]), int(z["w_seed"])
        x = seeded((b, t, n, c), x_seed, 1.0, dt)
        cot = seeded((b, t, n, c), x_seed + 1, 1.0, dt)
        shapes = {}
        for i in range(num_dim):
            shapes.update(O.din_param_shapes(f"DIMlist.{i}.", c, kernels[i], ratios, True, bool(beta)))
        p = O.synth_params(shapes, seed=w_seed, din_std=float(z["din_std"]), dtype=dt)
        for k in p:
            if "p_conv" in k:
                p[k] = p[k] * float(z["offset_boost"])
    return z, dict(b=b, t=t, n=n, c=c, num_dim=num_dim, beta=bool(beta), kernels=kernels, ratios=ratios, dt=dt), x, cot, p


DIN_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "din_*.npz")))


def test_golden_present():
    assert len(DIN_CASES) >= 10


def test_prep_images_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "prep_images.npz"))
    y = O.prep_images(torch.from_numpy(z["x"]))
    assert np.array_equal(y.numpy(), z["y"])


@pytest.mark.parametrize("path", DIN_CASES, ids=[os.path.basename(p)[:-4] for p in DIN_CASES])
def test_din_oracle_matches_reference(path):
    z, m, x, cot, p = load_din_case(path)
    po = {("DPI." + k): v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.clone().requires_grad_(True)
    out, mad = O.din_multi_inference(xo, po, "DPI.", m["kernels"], m["ratios"], True, m["beta"])
    (out * cot).sum().backward()
    tol = 1e-5 if m["dt"] == torch.float32 else 1e-11
    assert _rel(out.detach(), z["out"]) <= tol
    assert _rel(xo.grad, z["gx"]) <= 10 * tol
    if "mad" in z.files:
        assert _rel(mad.detach(), z["mad"]) <= tol
    for k in z.files:
        if k.startswith("g."):
            assert _rel(po["DPI." + k[2:]].grad, z[k]) <= 10 * tol, k
        if k.startswith("gsum."):
            g = po["DPI." + k[5:]].grad.double()
            assert abs(g.sum().item() - float(z[k])) <= 1e-3 * float(z["gabs." + k[5:]]) + 1e-6, k
    # integer corner decisions of the last module / ratio: bit-exact
    kh, kw = m["kernels"][-1]
    r = m["ratios"][-1]
    pre = f"DPI.DIMlist.{m['num_dim'] - 1}."
    _, _, aux = O.din_ratio_forward(x, p[pre[4:] + f"p_conv.{r}.weight"], p[pre[4:] + f"p_conv.{r}.bias"],
                                    p[pre[4:] + f"scale_conv.{r}.weight"], p[pre[4:] + f"scale_conv.{r}.bias"], (kh, kw), r,
                                    want_aux=True)
    for name in ("ly", "ry", "lx", "rx"):
        assert np.array_equal(aux[name].long().numpy().astype(np.int32), z[name]), name


MODEL_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "model_*.npz")))


def load_model_case(path):
    z = np.load(path)
    B, T, N, H, W, OH, OW, D, NFB, num_dim, beta, lite, hier = [int(v) for v in z["meta"]]
    kernels = [tuple(int(x) for x in k) for k in z["kernels"]]
    cfg = O.OracleCfg(backbone=str(z["backbone"]), image_size=(H, W), out_size=(OH, OW), emb_features=D, num_boxes=N,
                      num_frames=T, num_features_boxes=NFB, ST_kernel_size=kernels, sampling_ratio=[int(r) for r in z["ratios"]],
                      num_DIM=num_dim, beta_factor=bool(beta), lite_dim=lite or None, hierarchical_inference=bool(hier))
    seed = int(z["seed"])
    p = O.synth_params(O.model_param_shapes(cfg), seed=seed + 3, din_std=0.02)
    if hier:                                     # same recipe as tools/gen_golden.py::model_case
        g_ = torch.Generator().manual_seed(seed + 11)
        p["DPI.hier_LN.weight"] = 0.75 + 0.5 * torch.rand(p["DPI.hier_LN.weight"].shape, generator=g_)
        p["DPI.hier_LN.bias"] = 0.1 * torch.randn(p["DPI.hier_LN.bias"].shape, generator=g_)
    images, boxes, labels = O.synth_inputs(B, T, N, H, W, OH, OW, 8, seed=seed)
    if "smooth" in z.files and int(z["smooth"]):     # scene-like frames (tools/gen_golden.py --only full_scene)
        images = O.synth_scene_images(boxes, H, W, OH, OW, seed=seed + 7)
    return z, cfg, p, images, boxes, labels


@pytest.mark.parametrize("path", MODEL_CASES, ids=[os.path.basename(p)[:-4] for p in MODEL_CASES])
def test_model_oracle_matches_reference(path):
    z, cfg, p, images, boxes, labels = load_model_case(path)
    assert np.array_equal(labels.numpy(), z["labels"])
    po = {k: v.clone().requires_grad_("running_" not in k) for k, v in p.items()}
    out = O.dynamic_volleyball_forward(cfg, po, images.float(), boxes)
    loss = F.cross_entropy(out["activities"], labels)
    loss.backward()
    assert _rel(out["activities"].detach(), z["logits"]) <= 2e-4
    assert abs(loss.item() - float(z["loss"])) <= 2e-4 * max(1.0, abs(float(z["loss"])))
    for k in z.files:
        if k.startswith("g."):
            assert _rel(po[k[2:]].grad, z[k]) <= 1e-2, k


FULL_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "full_*.npz")))


def test_full_size_fixtures_are_present_and_seed_reproducible():
    """the 720x1280 fixtures (tools/gen_golden.py::full_case: two tie-free draws + the un-searched one; the oracle was held against the reference when they were written --
    re-running it here would take minutes): both backbones present, the seed recipe still regenerates the stored labels, every stage
    probe and gradient record is there"""
    assert {os.path.basename(p) for p in FULL_CASES} == {"full_inv3_720x1280_b1.npz", "full_vgg16_720x1280_cfg1_b2.npz",
                                                         "full_inv3_720x1280_b1_seed401_unsearched.npz", "full_inv3_720x1280_b1_scene.npz"}
    sm = np.load([p for p in FULL_CASES if "scene" in p][0])
    assert int(sm["smooth"]) == 1 and int(sm["searched"]) == 1
    bx = torch.tensor([[[[1.0, 1.0, 3.0, 4.0]], [[1.5, 1.0, 3.5, 4.0]]]])             # one actor, two frames, 6 x 8 feature grid
    fr = O.synth_scene_images(bx, 48, 64, 6, 8, seed=int(sm["seed"]) + 7)          # scene-like: neighbouring pixels and consecutive frames correlate
    a = fr[0, 0, 0].double()
    assert float(torch.corrcoef(torch.stack([a[:, :-1].flatten(), a[:, 1:].flatten()]))[0, 1]) > 0.9
    assert float(torch.corrcoef(torch.stack([fr[0, 0].double().flatten(), fr[0, 1].double().flatten()]))[0, 1]) > 0.5      # (the actor moved by half a feature cell; white noise: 0)
    un = np.load([p for p in FULL_CASES if "unsearched" in p][0])
    assert int(un["searched"]) == 0 and int(un["seed"]) == 401 and int(un["near_ties"]) >= 0 and float(un["min_actor_gap"]) >= 0.0
    for path in FULL_CASES:
        z = np.load(path)
        B, T, N, H, W, OH, OW = [int(v) for v in z["meta"][:7]]
        assert (H, W, T, N) == (720, 1280, 3, 12)
        r2 = np.random.default_rng(int(z["seed"]) + 2)
        assert np.array_equal(r2.integers(0, 8, size=(B,)).astype(np.int64), z["labels"])
        for key in ("fm0", "crops", "x_emb", "graph"):
            assert z[f"feat.{key}.sample"].shape == (8192,) and np.isfinite(z[f"feat.{key}.sample"]).all()
        assert z["logits"].shape == (B, 8) and "gs.fc_emb_1.weight" in z.files and "g.fc_activities.weight" in z.files


def test_roi_align_known_answers():
    """RoIAlign oracle (third-party algorithm, parity unpinned by the reference): hand-checked identities."""
    fm = torch.arange(2 * 3 * 6 * 8, dtype=torch.float32).reshape(2, 3, 6, 8)
    # a box whose K sample points land exactly on integer cells: x1=1,x2=6,K=5 -> spacing 1, first sample at 1+0.5-0.5=1
    boxes = torch.tensor([[1.0, 0.0, 6.0, 5.0], [0.0, 0.0, 0.0, 0.0]])
    ind = torch.tensor([1, 0], dtype=torch.int32)
    out, idx = O.roi_align(fm, boxes, ind, 5, return_index=True)
    assert torch.equal(out[0], fm[1][:, 0:5, 1:6])
    # zero box: samples at -0.5 -> out of range -> extrapolation value 0 (collective.py:201-203 padding boxes)
    assert torch.count_nonzero(out[1]) == 0 and bool(idx["oob_y"][1].all())
    # K == 1 special case: centre sample
    out1 = O.roi_align(fm, torch.tensor([[2.0, 2.0, 4.0, 4.0]]), torch.tensor([0], dtype=torch.int32), 1)
    assert torch.allclose(out1[0, :, 0, 0], fm[0][:, 2:4, 2:4].mean((1, 2)))


def test_din_zero_init_is_neighbourhood_mean():
    """Q5: zero-initialised p_conv/scale_conv => DIN = mean of the zero-padded 3x3 neighbourhood, then projection."""
    b, t, n, c = 1, 3, 5, 8
    x = seeded((b, t, n, c), 5)
    shapes = O.din_param_shapes("m.", c, (3, 3), [1], True, False)
    p = O.synth_params(shapes, seed=1, din_std=0.0)
    p["m.hidden_weight.weight"] = torch.eye(c)
    out, mad = O.din_person_inference(x, p, "m.", (3, 3), [1], True, False)
    ref = F.avg_pool2d(x.permute(0, 3, 1, 2), 3, 1, 1, count_include_pad=True).permute(0, 2, 3, 1)
    assert torch.allclose(out, ref, atol=1e-6)


def test_din_clamp_double_count_quirk():
    """Q3: with a size-1 kernel axis (no padding) the last row is counted twice at zero offset."""
    x = torch.ones(1, 4, 6, 2)
    shapes = O.din_param_shapes("m.", 2, (1, 3), [1], True, False)
    p = O.synth_params(shapes, seed=1, din_std=0.0)
    z, _ = O.din_ratio_forward(x, p["m.p_conv.1.weight"], p["m.p_conv.1.bias"], p["m.scale_conv.1.weight"],
                               p["m.scale_conv.1.bias"], (1, 3), 1)
    # interior columns: mean of three ones = 1; last time row doubled -> 2; edge columns lose one neighbour
    assert torch.allclose(z[0, 3, 2], torch.full((2,), 2.0))
    assert torch.allclose(z[0, 1, 2], torch.full((2,), 1.0))


def load_hier_case(golden_dir):
    z = np.load(os.path.join(golden_dir, "hier_k13_k31_t10_c1024.npz"))
    T, N, C = 10, 12, 1024
    kernels, ratios = [(1, 3), (3, 1)], [1]
    w_seed, x_seed = int(z["w_seed"]), int(z["x_seed"])
    shapes = {}
    for i, sub in enumerate(("DPI_1.", "DPI_2.")):
        shapes.update(O.din_param_shapes(sub, C, kernels[i], ratios, True, False))
    shapes["hier_LN.weight"] = (T, N, C)
    shapes["hier_LN.bias"] = (T, N, C)
    p = O.synth_params(shapes, seed=w_seed, din_std=0.02)
    g = torch.Generator().manual_seed(w_seed + 1)
    p["hier_LN.weight"] = 0.75 + 0.5 * torch.rand((T, N, C), generator=g)
    p["hier_LN.bias"] = 0.1 * torch.randn((T, N, C), generator=g)
    x = seeded((1, T, N, C), x_seed)
    cot = seeded((1, T, N, C), x_seed + 1)
    return z, p, x, cot, kernels, ratios


def test_hierarchical_oracle_matches_reference(golden_dir):
    """row D7 (reference crashes as shipped; golden captured with the SURVEY 8c no-source-patch recipe, dropout neutralised)"""
    z, p, x, cot, kernels, ratios = load_hier_case(golden_dir)
    po = {("DPI." + k): v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.clone().requires_grad_(True)
    out, _ = O.din_hierarchical_inference(xo, po, "DPI.", kernels, ratios, True, False)
    (out * cot).sum().backward()
    assert _rel(out.detach(), z["out"]) <= 1e-5 and _rel(xo.grad, z["gx"]) <= 1e-4
    for k in z.files:
        if k.startswith("gsum."):
            gq = po["DPI." + k[5:]].grad.double()
            assert abs(gq.sum().item() - float(z[k])) <= 1e-3 * float(z["gabs." + k[5:]]) + 1e-6, k


def load_collective_case(golden_dir):
    z = np.load(os.path.join(golden_dir, "collective_vgg16_96x160.npz"))
    H_, W_, OH, OW, B, T, MAXN, NFB, A = 96, 160, 3, 5, 3, 3, 6, 64, 4
    seed = int(z["seed"])
    ocfg = O.OracleCfg(image_size=(H_, W_), out_size=(OH, OW), num_boxes=MAXN, num_frames=T, num_features_boxes=NFB,
                       ST_kernel_size=(3, 3), sampling_ratio=[1], num_activities=A, collective=True)
    p = O.synth_params(O.model_param_shapes(ocfg), seed=seed + 3, din_std=0.05)
    g = torch.Generator().manual_seed(seed + 5)
    p["dpi_nl.weight"] = 0.75 + 0.5 * torch.rand((T, NFB), generator=g)
    p["dpi_nl.bias"] = 0.1 * torch.randn((T, NFB), generator=g)
    images, boxes, labels = O.synth_inputs(B, T, MAXN, H_, W_, OH, OW, A, seed=seed)
    counts = torch.from_numpy(z["counts"])
    for b in range(B):
        boxes[b, :, int(counts[b, 0]):] = 0.0
    return z, ocfg, p, images, boxes, labels, counts


def test_collective_oracle_matches_reference(golden_dir):
    """row C: variable actors per clip (incl. N=1), zero padding boxes"""
    z, ocfg, p, images, boxes, labels, counts = load_collective_case(golden_dir)
    assert np.array_equal(labels.numpy(), z["labels"])
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    out = O.dynamic_collective_forward(ocfg, po, images.float(), boxes, counts)
    loss = F.cross_entropy(out["activities"], labels)
    loss.backward()
    assert _rel(out["activities"].detach(), z["logits"]) <= 2e-4
    for k in z.files:
        if k.startswith("gsum."):
            gq = po[k[5:]].grad.double()
            assert abs(gq.sum().item() - float(z[k])) <= 2e-3 * float(z["gabs." + k[5:]]) + 1e-6, k


MODE_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mode_*.npz")))


def load_mode_case(path):
    z = np.load(path)
    b, t, n, c, beta, scale = [int(v) for v in z["meta"]]
    meta = dict(mode=str(z["mode"]), kernel=tuple(int(v) for v in z["kernel"]), ratios=[int(r) for r in z["ratios"]], beta=bool(beta),
                scale=bool(scale), c=c)
    p = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    return z, meta, torch.from_numpy(z["x"]), torch.from_numpy(z["cot"]), p


@pytest.mark.parametrize("path", MODE_CASES, ids=[os.path.basename(p)[:-4] for p in MODE_CASES])
def test_plain_and_parallel_oracle_matches_reference(path):
    """dynamic_sampling=False (plain_infer_ratio) and parallel_inference=True (parallel_infer) -- fixtures from the reference's
    own per-ratio methods combined as dynamic_infer_module.py:137-147 does."""
    z, m, x, cot, p = load_mode_case(path)
    assert len(MODE_CASES) >= 3
    po = {("DPI." + k): v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.clone().requires_grad_(True)
    out, _ = O.din_person_inference(xo, po, "DPI.", m["kernel"], m["ratios"], m["scale"], m["beta"], dynamic_sampling=m["mode"] == "parallel",
                                    parallel_inference=m["mode"] == "parallel")
    (out * cot).sum().backward()
    assert _rel(out.detach(), z["out"]) <= 1e-5 and _rel(xo.grad, z["gx"]) <= 1e-4
    for k in z.files:
        if k.startswith("g."):
            assert _rel(po["DPI." + k[2:]].grad, z[k]) <= 1e-4, k


TCE_CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tce_*.npz")))


def load_tce_case(path):
    z = np.load(path)
    B, T, N, H, W, OH, OW, D, NFB, num_dim, hier = [int(v) for v in z["meta"]]
    kernels = [tuple(int(x) for x in k) for k in z["kernels"]]
    cfg = O.OracleCfg(backbone="vgg16", image_size=(H, W), out_size=(OH, OW), emb_features=D, num_boxes=N, num_frames=T,
                      num_features_boxes=NFB, ST_kernel_size=kernels, sampling_ratio=[int(r) for r in z["ratios"]], num_DIM=num_dim,
                      hierarchical_inference=bool(hier))
    seed = int(z["seed"])
    p = O.tce_synth_params(cfg, seed)
    if hier:                                     # same recipe as tools/gen_golden.py::tce_case
        g_ = torch.Generator().manual_seed(seed + 11)
        p["DPI.hier_LN.weight"] = 0.75 + 0.5 * torch.rand(p["DPI.hier_LN.weight"].shape, generator=g_)
        p["DPI.hier_LN.bias"] = 0.1 * torch.randn(p["DPI.hier_LN.bias"].shape, generator=g_)
    images, boxes, labels = O.synth_inputs(B, T, N, H, W, OH, OW, 8, seed=seed)
    if "smooth" in z.files and int(z["smooth"]):     # scene-like frames (tools/gen_golden.py --only full_scene)
        images = O.synth_scene_images(boxes, H, W, OH, OW, seed=seed + 7)
    return z, cfg, p, images, boxes, labels


def test_tce_golden_present():
    assert len(TCE_CASES) >= 3


@pytest.mark.parametrize("path", TCE_CASES, ids=[os.path.basename(p)[:-4] for p in TCE_CASES])
def test_tce_oracle_matches_reference(path):
    """Dynamic_TCE_volleyball (reference infer_model.py:237-468) imported and run by tools/gen_golden.py::tce_case: logits, loss, one head's
    attention map, the context encoding and every stored parameter gradient"""
    z, cfg, p, images, boxes, labels = load_tce_case(path)
    assert np.array_equal(labels.numpy(), z["labels"])
    po = {k: v.clone().requires_grad_("running_" not in k) for k, v in p.items()}
    out, inter = O.dynamic_tce_volleyball_forward(cfg, po, images.float(), boxes, return_intermediates=True)
    loss = F.cross_entropy(out["activities"], labels)
    loss.backward()
    assert _rel(out["activities"].detach(), z["logits"]) <= 2e-4
    assert abs(loss.item() - float(z["loss"])) <= 2e-4 * max(1.0, abs(float(z["loss"])))
    assert _rel(inter["enc"].detach(), z["enc"]) <= 1e-4
    b, t, n = images.shape[0], images.shape[1], cfg.num_boxes
    _, atts = O.tce_context_encoding(inter["x"].reshape(b * t * n, -1), inter["context"], po, return_attention=True)
    assert _rel(atts[int(z["att_head"])].detach(), z["att_map"]) <= 1e-4
    for k in z.files:
        if k.startswith("g."):
            assert _rel(po[k[2:]].grad, z[k]) <= 1e-2, k
        if k.startswith("gsum."):
            name = k[5:]
            assert abs(po[name].grad.double().sum().item() - float(z[k])) <= 2e-3 * float(z["gabs." + name]) + 1e-6, name


def test_context_position_embedding_known_answers():
    """positional_encoding.py:67-92 by hand: 1-based coordinates x 16, channel pairs (sin, cos) at T^(2i/256), y half then x half"""
    pos = O.context_position_embedding(3, 5)
    assert tuple(pos.shape) == (512, 3, 5)
    import math
    assert abs(float(pos[0, 0, 0]) - math.sin(16.0)) < 1e-6 and abs(float(pos[1, 0, 3]) - math.cos(16.0)) < 1e-6      # y = 1, any x
    assert abs(float(pos[0, 2, 1]) - math.sin(48.0)) < 1e-5                                                            # y = 3
    assert abs(float(pos[256, 1, 4]) - math.sin(80.0)) < 1e-5 and abs(float(pos[257, 1, 4]) - math.cos(80.0)) < 1e-5    # x = 5
    t2 = 10000.0 ** (2.0 / 256.0)
    assert abs(float(pos[2, 1, 0]) - math.sin(32.0 / t2)) < 1e-5 and abs(float(pos[259, 0, 1]) - math.cos(32.0 / t2)) < 1e-5
