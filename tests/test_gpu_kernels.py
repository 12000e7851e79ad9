"""GPU (-m gpu): per-kernel parity of the HIP path, called through the C ABI (ctypes), against CPU fp32 references.
Integer/index outputs are compared bit-exactly; fp32 within the tolerance written next to each check."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import din_oracle as O

from tests.conftest import Measured

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from din_amd import _lib, nhwc, ops
    return _lib.load(), _lib, nhwc, ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return Measured(((a - b).abs().max() / (b.abs().max() + 1e-30)).item())


def to_nhwc(x, dtype, ld=None, coff=0):
    """CPU NCHW fp32 -> device NHWC buffer (test plumbing)."""
    n, c, h, w = x.shape
    ld = ld or c
    buf = torch.zeros(n, h, w, ld, dtype=dtype)
    buf[..., coff:coff + c] = x.permute(0, 2, 3, 1).to(dtype)
    return buf.cuda()


def from_nhwc(buf, c, coff=0):
    return buf[..., coff:coff + c].float().cpu().permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # name, nb, cin, h, w, cout, k, s, p, d
    ("3x3_s1_p1", 2, 16, 13, 17, 24, (3, 3), (1, 1), (1, 1), 1),
    ("conv1_cin3", 2, 3, 20, 24, 64, (3, 3), (1, 1), (1, 1), 1),
    ("1x1", 1, 64, 9, 11, 80, (1, 1), (1, 1), (0, 0), 1),
    ("3x3_s2_p0", 2, 32, 15, 19, 48, (3, 3), (2, 2), (0, 0), 1),
    ("3x3_s2_p1", 2, 32, 14, 18, 48, (3, 3), (2, 2), (1, 1), 1),
    ("1x1_s2", 1, 32, 9, 11, 32, (1, 1), (2, 2), (0, 0), 1),
    ("3x3_s2_cin288", 1, 288, 21, 25, 64, (3, 3), (2, 2), (0, 0), 1),           # Mixed_6a.branch3x3 dgrad: parity classes on 3 x 96-wide tiles
    ("5x5_p2", 1, 48, 10, 12, 64, (5, 5), (1, 1), (2, 2), 1),
    ("1x7", 1, 32, 9, 14, 32, (1, 7), (1, 1), (0, 3), 1),
    ("7x1", 1, 32, 14, 9, 192, (7, 1), (1, 1), (3, 0), 1),
    ("dil3_grid", 2, 32, 10, 12, 27, (3, 3), (1, 1), (3, 3), 3),
    ("big_tile", 1, 128, 40, 48, 256, (3, 3), (1, 1), (1, 1), 1),
    ("tile160_1x7", 1, 64, 24, 40, 160, (1, 7), (1, 1), (0, 3), 1),      # 160-filter tile: padded loader passes on the 8-wave variant
    ("tile96_3x3", 1, 64, 24, 40, 96, (3, 3), (1, 1), (1, 1), 1),         # 96-filter tile, likewise
    ("tile192_1x1", 1, 96, 24, 40, 192, (1, 1), (1, 1), (0, 0), 1),
    ("linear_splitk", 1, 1600, 1, 72, 128, (1, 1), (1, 1), (0, 0), 1),
    ("splitk_3x3_multi_image", 6, 256, 8, 12, 512, (3, 3), (1, 1), (1, 1), 1),   # VGG conv4_1 on a small frame: split-K, tiles span images
    ("korder_rows_w96", 6, 64, 64, 96, 64, (3, 3), (1, 1), (1, 1), 1),            # VGG conv1_2 on the smoke frame: tap-inner k order, tiles span rows
    # stem shapes (>= 256K pixels): bf16 runs the stationary-filter halo kernel (fwd cpt4, dgrad cpt4 / cpt8), ragged tile edges
    ("stem_32_32_p0", 1, 32, 515, 517, 32, (3, 3), (1, 1), (0, 0), 1),
    ("stem_32_64_p1", 2, 32, 363, 365, 64, (3, 3), (1, 1), (1, 1), 1),
    ("image_3_32_s2", 1, 3, 1031, 1029, 32, (3, 3), (2, 2), (0, 0), 1),     # Inception Conv2d_1a: image layer of the halo kernel
    # mid-network multi-tap shapes (>= 64K pixels): bf16 runs conv_halo_kernel for fwd and dgrad (partial 64-channel blocks,
    # ragged tiles, every tap geometry it is instantiated for)
    ("halo_3x3_64_96", 6, 64, 87, 157, 96, (3, 3), (1, 1), (1, 1), 1),
    ("halo_3x3_80_192_p0", 2, 80, 181, 321, 192, (3, 3), (1, 1), (0, 0), 1),
    ("halo_5x5_48_64", 6, 48, 87, 157, 64, (5, 5), (1, 1), (2, 2), 1),
    ("halo_3x3_96_96", 6, 96, 87, 157, 96, (3, 3), (1, 1), (1, 1), 1),        # (+ the three shapes of conv_wgrad_halo_kernel: dW stationary)
    ("halo_3x3_96_80", 6, 96, 87, 157, 80, (3, 3), (1, 1), (1, 1), 1),        # 80 filters: conv_halo_kernel<80>, 48 + 32 rows (also the dgrad of halo_3x3_80_192_p0)
    ("halo_1x7_128_160", 20, 128, 43, 78, 160, (1, 7), (1, 1), (0, 3), 1),
    ("halo_7x1_160_192", 20, 160, 43, 78, 192, (7, 1), (1, 1), (3, 0), 1),
    # 256-pixel software-pipelined tiles (conv_gather_pipe_kernel<128 | 192 | 256>, bf16; forced on these small shapes with DIN_GATHER_PIPE=2):
    # ragged pixel tiles, row / column padding taps, two filter tiles, filter tile wider than the bank, dgrad through the same kernel
    ("gp192_1x1", 2, 96, 24, 41, 192, (1, 1), (1, 1), (0, 0), 1),
    ("gp192_7x1", 8, 160, 43, 78, 192, (7, 1), (1, 1), (3, 0), 1),
    ("gp192_1x7_384", 4, 192, 43, 78, 384, (1, 7), (1, 1), (0, 3), 1),
    ("gp128_3x3", 3, 128, 61, 70, 256, (3, 3), (1, 1), (1, 1), 1),
    ("gp256_3x3_p0", 2, 64, 83, 79, 512, (3, 3), (1, 1), (0, 0), 1),          # DIN_CONV_BN=256: the 256-filter tile (VGG conv3+)
    ("gp192_dgrad_3x3", 4, 192, 80, 78, 192, (3, 3), (1, 1), (1, 1), 1),
    ("gp128_5x5_176", 4, 64, 80, 78, 176, (5, 5), (1, 1), (2, 2), 1),
    # persistent streaming 1x1 kernel (conv1x1_stream_kernel<64 | 96 | 192>, bf16; forced on these small maps with DIN_CONV_STREAM=2): ragged last pixel
    # tile, one / several / partial 64-channel blocks, one to three filter tiles, a filter tile wider than the bank, fwd and dgrad (+ mask, accumulate)
    ("st_64_80", 3, 64, 37, 41, 80, (1, 1), (1, 1), (0, 0), 1),
    ("st_288_176", 2, 288, 35, 45, 176, (1, 1), (1, 1), (0, 0), 1),
    ("st_48_8", 1, 48, 19, 23, 8, (1, 1), (1, 1), (0, 0), 1),
    ("st_192_64_long", 8, 192, 87, 157, 64, (1, 1), (1, 1), (0, 0), 1),      # > 256 items: every workgroup walks several
    ("st_96_176", 2, 96, 31, 45, 176, (1, 1), (1, 1), (0, 0), 1),            # the 192-filter tile on its 3-slot ring (forward), 96-filter tile (dgrad)
    # filters resident in registers (conv1x1_regw_kernel<20 | 24, masked>, bf16; forced on these small maps with DIN_CONV_REGW=2): one to
    # four classes of 192 filters, a last class with idle waves / partial 16-byte pieces, ragged last pixel tile, fewer tiles than teams, forward
    # (reduction 640 / 768) and dgrad (reduction 640 / 736 / 768: plain and with the ReLU mask through the ring)
    ("rw_768_192", 8, 768, 57, 55, 192, (1, 1), (1, 1), (0, 0), 1),          # (>= 192 tiles each: below that the planner splits the reduction)
    ("rw_768_448", 5, 768, 43, 41, 448, (1, 1), (1, 1), (0, 0), 1),          # three classes, the third with 64 filters (two idle waves); dgrad elsewhere
    ("rw_640_200", 12, 640, 33, 37, 200, (1, 1), (1, 1), (0, 0), 1),         # reduction 640 (20 k-steps); two classes, the second with 8 filters
    ("rw_160_736", 19, 160, 35, 39, 736, (1, 1), (1, 1), (0, 0), 1),         # dgrad: reduction 736 (the last stage half zero padding) -> 160 channels (a wave with 16 filters)
    ("rw_768_768", 4, 768, 43, 78, 768, (1, 1), (1, 1), (0, 0), 1),          # fwd and dgrad, four classes, every team walks several tiles
    ("rw_96_640", 20, 96, 35, 39, 640, (1, 1), (1, 1), (0, 0), 1),           # dgrad: reduction 640 -> 96 channels (two waves idle)
    # ... and its short-reduction form (<6 | 8 | 10, ..., two workgroups per CU, classes of 128 filters>: the Mixed_5 block entries)
    ("rw_192_176", 2, 192, 21, 25, 176, (1, 1), (1, 1), (0, 0), 1),          # Mixed_5b sibling group: 3 stages, classes 128 + 48
    ("rw_256_264", 6, 256, 45, 47, 264, (1, 1), (1, 1), (0, 0), 1),          # 4 stages, classes 128 + 128 + 8; dgrad: reduction 264 (4.1 stages) -> 256 channels
    ("rw_288_64", 12, 288, 45, 47, 64, (1, 1), (1, 1), (0, 0), 1),           # Mixed_5d branch_pool / Mixed_6a entry: 4.5 stages (half a stage zero padding), two waves idle
    ("rw_64_208", 3, 64, 33, 37, 208, (1, 1), (1, 1), (0, 0), 1),            # dgrad: reduction 208 -> 64 channels
    # per-lane k-walk (conv_gather_fast_kernel<..., FASTK, LANEK>, bf16): reduction channels that are not whole 64-channel k-steps per tap -- a
    # k-step straddles two taps -- on every 8-wave tile width; ragged pixel tiles, padding taps, a last k-step that runs past the last tap
    ("lanek_3x3_80_192_p0", 1, 80, 45, 70, 192, (3, 3), (1, 1), (0, 0), 1),     # Conv2d_4a (backbone.py:52): 11.25 k-steps; dgrad: whole k-steps (FASTK)
    ("lanek_1x7_160_160", 2, 160, 19, 37, 160, (1, 7), (1, 1), (0, 3), 1),       # Mixed_6c / 6d 7-tap layers: forward AND dgrad straddle (17.5 k-steps)
    ("lanek_7x1_160_192", 2, 160, 37, 19, 192, (7, 1), (1, 1), (3, 0), 1),
    ("lanek_3x3_96_128", 2, 96, 21, 23, 128, (3, 3), (1, 1), (1, 1), 1),         # 12 chunks per tap, 128-filter tile
    ("lanek_3x3_72_64", 1, 72, 33, 31, 64, (3, 3), (1, 1), (1, 1), 1),           # 9 chunks per tap, 64-filter tile
]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(env, case, dtype, monkeypatch):
    lib, L, nhwc, ops = env
    name, nb, cin, h, w, cout, k, s, p, dil = case
    if name.startswith("halo_"):
        monkeypatch.setenv("DIN_CONV_HALO", "2")        # the planner only picks the halo kernel where it wins; cover every instantiation
        monkeypatch.setenv("DIN_WGRAD_HALO", "2")       # ... and the halo weight-gradient kernel only on launches of >= 256K pixels
    if name.startswith("st_"):
        monkeypatch.setenv("DIN_CONV_STREAM", "2")      # ... and the streaming 1x1 kernel only on maps of >= 256K pixels
    if name.startswith("rw_"):
        monkeypatch.setenv("DIN_CONV_REGW", "2")        # ... and the register-resident-filter kernel only on maps of >= 64K / 128K pixels
    if name.startswith("gp"):
        monkeypatch.setenv("DIN_GATHER_PIPE", "2")      # ... and the 256-pixel pipelined tiles only on launches that fill the chip
        monkeypatch.setenv("DIN_CONV_HALO", "0")
        monkeypatch.setenv("DIN_CONV_TILE", "0")
        if name.startswith("gp256") and dtype == "bf16":
            monkeypatch.setenv("DIN_CONV_BN", "256")
    dt = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    epc = 4 if dtype == "fp32" else 8
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(nb, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, *k, generator=g) * (2.0 / (cin * k[0] * k[1])) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    if dtype == "bf16":                       # compare like with like: reference sees the bf16-rounded operands
        x, wt = x.bfloat16().float(), wt.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    y_ref = F.relu(F.conv2d(xr, wr, br, stride=s, padding=p, dilation=dil))
    oh, ow = y_ref.shape[2:]
    cot = torch.randn(y_ref.shape, generator=g)
    if dtype == "bf16":
        cot = cot.bfloat16().float()
    gz_ref = cot * (y_ref > 0).float()                      # dZ: gradient at the pre-activation
    y_ref.backward(cot)

    ldi = (cin + epc - 1) // epc * epc
    ldo = (cout + 7) // 8 * 8 + 16                          # extra room: exercise pixel stride != channels
    coff = 8
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, oh, ow, cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = k[0], k[1], s[0], s[1], p[0], p[1], dil, dil
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = ldi, 0, ldo, coff, dt
    st = None
    if name.startswith("gp") and dtype == "bf16":
        bm, bn = C.c_int32(0), C.c_int32(0)
        lib.din_conv_kernel_tile(C.byref(d), 0, C.byref(bm), C.byref(bn))
        assert bm.value == 2, f"{name}: forward not on the pipelined gather kernel (tile {bm.value} x {bn.value})"
    if name.startswith("st_") and dtype == "bf16":
        for which in (0, 1):
            bm, bn = C.c_int32(0), C.c_int32(0)
            lib.din_conv_kernel_tile(C.byref(d), which, C.byref(bm), C.byref(bn))
            assert bm.value == 4, f"{name}: {('forward', 'dgrad')[which]} not on the streaming 1x1 kernel (tile {bm.value} x {bn.value})"
    if name.startswith("lanek_") and dtype == "bf16":
        monkeypatch.setenv("DIN_CONV_LANEK", "2")       # forward and data gradient (the planner keeps data gradients on the general loop: slower there)
        fl = C.c_int32(0)
        lib.din_conv_kernel_variant(C.byref(d), 0, C.byref(fl))
        assert fl.value & 4, f"{name}: forward not on the per-lane k-walk instantiation (flags {fl.value})"
    if name.startswith("rw_") and dtype == "bf16":
        for which, cred, cprod in ((0, cin, cout), (1, cout, cin)):
            bm, bn = C.c_int32(0), C.c_int32(0)
            lib.din_conv_kernel_tile(C.byref(d), which, C.byref(bm), C.byref(bn))
            nks = (cred + 63) // 64 * 2                      # 64-channel stages x 2; short reductions: classes of 128 filters, at most four
            want = (nks in (20, 24) and cprod <= 768) or (nks in (6, 8, 10) and cprod <= 512)
            assert (bm.value == 5) == want, f"{name}: {('forward', 'dgrad')[which]} kernel code {bm.value}"
    xin = to_nhwc(x, tdt, ldi)
    wdev, bdev = wt.cuda(), bias.cuda()
    wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), dtype=tdt, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), wdev.data_ptr(), None, wpk.data_ptr(), 0, st))
    out = torch.full((nb, oh, ow, ldo), 7.0, dtype=tdt, device="cuda")
    wsb = lib.din_conv_workspace_bytes(C.byref(d), 0)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
    L.check(lib.din_conv_fwd(C.byref(d), xin.data_ptr(), wpk.data_ptr(), bdev.data_ptr(), out.data_ptr(),
                             L.CONV_BIAS | L.CONV_RELU, ws.data_ptr(), wsb, st))
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == "fp32" else 1.5e-2               # bf16: output rounding 2^-8 + fp32-accumulated products
    assert rel(from_nhwc(out, cout, coff), y_ref) <= tol
    assert float(out[..., :coff].float().min()) == 7.0 and float(out[..., coff + cout:].float().min()) == 7.0, "wrote outside its channel range"

    # ---- wgrad + bias grad
    if name in ("halo_3x3_64_96", "halo_5x5_48_64", "halo_3x3_96_96") and dtype == "bf16":
        bm, bn = C.c_int32(0), C.c_int32(0)
        lib.din_conv_kernel_tile(C.byref(d), 2, C.byref(bm), C.byref(bn))
        assert bm.value == 3, f"{name}: weight gradient not on conv_wgrad_halo_kernel (code {bm.value}, {bn.value})"
    gz = to_nhwc(gz_ref, tdt, ldo, coff)
    dw = torch.empty_like(wdev)
    db = torch.empty(cout, device="cuda")
    wsb = lib.din_conv_workspace_bytes(C.byref(d), 2)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
    L.check(lib.din_conv_wgrad(C.byref(d), xin.data_ptr(), gz.data_ptr(), dw.data_ptr(), db.data_ptr(), None, None, None, 0,
                               ws.data_ptr(), wsb, st))
    torch.cuda.synchronize()
    tolg = 5e-5 if dtype == "fp32" else 2e-2
    assert rel(dw, wr.grad) <= tolg
    assert rel(db, br.grad) <= tolg

    # ---- dgrad (+ fused ReLU mask of the producer of x, + accumulate)
    if cin % epc == 0:
        wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=tdt, device="cuda")
        L.check(lib.din_conv_pack_weights(C.byref(d), wdev.data_ptr(), None, wpt.data_ptr(), 1, st))
        dx = torch.zeros((nb, h, w, ldi), dtype=tdt, device="cuda")
        wsb = lib.din_conv_workspace_bytes(C.byref(d), 1)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
        L.check(lib.din_conv_dgrad(C.byref(d), gz.data_ptr(), wpt.data_ptr(), dx.data_ptr(), None, 0, 0, 0, ws.data_ptr(), wsb, st))
        torch.cuda.synchronize()
        assert rel(from_nhwc(dx, cin), xr.grad) <= tolg
        L.check(lib.din_conv_dgrad(C.byref(d), gz.data_ptr(), wpt.data_ptr(), dx.data_ptr(), xin.data_ptr(), ldi, 0,
                                   L.CONV_MASK | L.CONV_ACCUM, ws.data_ptr(), wsb, st))
        torch.cuda.synchronize()
        want = xr.grad + xr.grad * (x > 0).float()
        assert rel(from_nhwc(dx, cin), want) <= 2 * tolg
        if name.startswith("rw_"):                       # mask without accumulate: the launch the block-entry dgrads make (mask stages through the ring)
            dx.fill_(3.0)
            L.check(lib.din_conv_dgrad(C.byref(d), gz.data_ptr(), wpt.data_ptr(), dx.data_ptr(), xin.data_ptr(), ldi, 0, L.CONV_MASK,
                                       ws.data_ptr(), wsb, st))
            torch.cuda.synchronize()
            assert rel(from_nhwc(dx, cin), xr.grad * (x > 0).float()) <= tolg


@pytest.mark.parametrize("shape", [(80, 192, (3, 3), (0, 0), 2, 61, 77), (160, 160, (1, 7), (0, 3), 3, 43, 78), (160, 192, (7, 1), (3, 0), 3, 43, 78)],
                         ids=["conv2d_4a", "mixed_6c_1x7", "mixed_6c_7x1"])
def test_lane_k_walk_is_bit_identical_to_the_general_loop(env, monkeypatch, shape):
    """conv_gather_fast_kernel<..., LANEK> against the general loop (DIN_CONV_LANEK=0) on the shapes it was built for (reference
    backbone/backbone.py:52 Conv2d_4a_3x3, :67-74 the 160-channel 7-tap layers of InceptionC): same 16-byte chunks per k-step, same
    accumulation order -> the same bits, forward (bias + ReLU) and data gradient (mask + accumulate)."""
    lib, L, nhwc, ops = env
    cin, cout, k, p, nb, h, w = shape
    g = torch.Generator().manual_seed(cin + cout)
    bf = torch.bfloat16
    oh, ow = h + 2 * p[0] - k[0] + 1, w + 2 * p[1] - k[1] + 1
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, oh, ow, cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = k[0], k[1], 1, 1, p[0], p[1], 1, 1
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin + 16, 8, cout + 8, 0, L.DIN_BF16
    xin = torch.randn(nb, h, w, cin + 16, generator=g).to(bf).cuda()
    gy = torch.randn(nb, oh, ow, cout + 8, generator=g).to(bf).cuda()
    wt = (torch.randn(cout, cin, *k, generator=g) * (2.0 / (cin * k[0] * k[1])) ** 0.5).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), dtype=bf, device="cuda")
    wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=bf, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpk.data_ptr(), 0, None))
    L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpt.data_ptr(), 1, None))
    base = torch.randn(nb, h, w, cin + 16, generator=g).to(bf).cuda()
    outs = []
    for mode in ("2", "0"):                                        # (2: data gradients too; the shipped 1 keeps them on the general loop)
        monkeypatch.setenv("DIN_CONV_LANEK", mode)
        fl = C.c_int32(0)
        lib.din_conv_kernel_variant(C.byref(d), 0, C.byref(fl))
        assert bool(fl.value & 4) == (mode == "2"), (mode, fl.value)
        y = torch.full((nb, oh, ow, cout + 8), 5.0, dtype=bf, device="cuda")
        wsb = lib.din_conv_workspace_bytes(C.byref(d), 0)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
        L.check(lib.din_conv_fwd(C.byref(d), xin.data_ptr(), wpk.data_ptr(), bias.data_ptr(), y.data_ptr(), L.CONV_BIAS | L.CONV_RELU,
                                 ws.data_ptr(), wsb, None))
        dx = base.clone()
        wsb = lib.din_conv_workspace_bytes(C.byref(d), 1)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
        L.check(lib.din_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx.data_ptr(), xin.data_ptr(), cin + 16, 8,
                                   L.CONV_MASK | L.CONV_ACCUM, ws.data_ptr(), wsb, None))
        torch.cuda.synchronize()
        outs.append((y, dx))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0][..., :cout].float().abs().max()) > 0.1 and not bool((outs[0][1] == base).all())


@pytest.mark.parametrize("cout", [32, 64], ids=["dY32_conv_small_4", "dY64_conv_small_8"])
def test_stem_dgrad_early_operand_request_is_bit_identical(env, monkeypatch, cout):
    """conv_small_kernel<..., EPI = true> (dgrad launches: ReLU mask / accumulate operands requested at the top of the tile) against the
    in-loop fetch (DIN_CONV_SMALL_EPI=0) on the two stem dgrad geometries, mask + accumulate, ragged right / bottom tile edges."""
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(41)
    nb, h, w, bf = 2, 365, 645, torch.bfloat16                   # 363 x 643 outputs: not multiples of the 8 x 32 tile
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, 32, h - 2, w - 2, cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = 3, 3, 1, 1, 0, 0, 1, 1
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = 32, 0, cout, 0, L.DIN_BF16
    xin = torch.randn(nb, h, w, 32, generator=g).to(bf).cuda()
    gy = torch.randn(nb, h - 2, w - 2, cout, generator=g).to(bf).cuda()
    wt = (torch.randn(cout, 32, 3, 3, generator=g) * 0.1).cuda()
    wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=bf, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpt.data_ptr(), 1, None))
    base = torch.randn(nb, h, w, 32, generator=g).to(bf).cuda()
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("DIN_CONV_SMALL_EPI", mode)
        dx = base.clone()
        L.check(lib.din_conv_dgrad(C.byref(d), gy.data_ptr(), wpt.data_ptr(), dx.data_ptr(), xin.data_ptr(), 32, 0,
                                   L.CONV_MASK | L.CONV_ACCUM, None, 0, None))
        torch.cuda.synchronize()
        outs.append(dx)
    assert torch.equal(outs[0], outs[1])
    changed = outs[0] != base
    assert 0.3 < float(changed.float().mean()) < 0.7                      # about half of the positions pass the ReLU mask
    assert not bool((changed & ~(xin > 0)).any())                         # nothing was added where the mask is off


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_pipe", "bf16_stream", "bf16_regw", "bf16_regw_short"])
def test_conv_fwd_two_destinations(env, dtype, monkeypatch):
    """din_conv_fwd2: sibling 1x1 convs of one input as ONE launch -- channels [0, csplit) into the first tensor's view, the rest into a second
    tensor; equals the separate convs, and nothing outside the two channel ranges is touched.  bf16_pipe: through the 256-pixel tiles;
    bf16_stream: through the persistent streaming kernel (conv1x1_stream_kernel)."""
    lib, L, nhwc, ops = env
    monkeypatch.setenv("DIN_CONV_STREAM", "2" if dtype == "bf16_stream" else "0")
    regw = dtype == "bf16_regw"
    monkeypatch.setenv("DIN_CONV_REGW", "2" if dtype.startswith("bf16_regw") else "0")     # (bf16_regw_short: the Mixed_5 group 192 -> 64 | 48 + 64 below, raw-stored
    if dtype in ("bf16_stream", "bf16_regw", "bf16_regw_short"):                           #  sibling included, on the short-reduction instantiation)
        dtype = "bf16"
    if dtype == "bf16_pipe":
        monkeypatch.setenv("DIN_GATHER_PIPE", "2")
        dtype = "bf16"
    dt = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(21)
    nb, h, w, cin = 2, 40, 48, 192
    couts = (64, 48, 64)
    if regw:                                                 # the Mixed_6c entry group: 768 -> 192 | 160 + 160 (conv1x1_regw_kernel, three classes)
        nb, h, w, cin, couts = 3, 43, 78, 768, (192, 160, 160)
    ctot = sum(couts)
    x = torch.randn(nb, cin, h, w, generator=g)
    ws_ = [torch.randn(c, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5 for c in couts]
    bias = torch.randn(ctot, generator=g) * 0.1
    if dtype == "bf16":
        x, ws_ = x.bfloat16().float(), [t.bfloat16().float() for t in ws_]
    ref = F.relu(F.conv2d(x, torch.cat(ws_), bias))
    ldi, ld1, off1, ld2, off2 = cin, 256, 8, 128, 8
    if regw:
        ld1, ld2 = 768, 336
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, h, w, ctot
    d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1
    d.ph = d.pw = 0
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = ldi, 0, ld1, off1, dt
    xin = to_nhwc(x, tdt, ldi)
    # the fused bank: every sibling packed at its row offset (what nhwc._pack_cache does with din_conv_pack_desc / din_conv_pack_multi)
    bank = torch.zeros(lib.din_conv_packed_elems(C.byref(d), 0), dtype=tdt, device="cuda")
    esz = 4 if dtype == "fp32" else 2
    keep, descs, row0 = [], [], 0
    for wt in ws_:
        dm = L.ConvDesc()
        for f, _ in L.ConvDesc._fields_:
            setattr(dm, f, getattr(d, f))
        dm.cout = wt.shape[0]
        wd = wt.cuda()
        keep.append(wd)
        pd = L.PackDesc()
        L.check(lib.din_conv_pack_desc(C.byref(dm), wd.data_ptr(), None, bank.data_ptr(), 0, C.byref(pd)))
        pd.out = bank.data_ptr() + row0 * pd.kelems * esz
        pd.rows_pad = pd.rows
        descs.append(pd)
        row0 += wt.shape[0]
    chunk = 4096
    layer_of, chunk_index = [], []
    for li, pd in enumerate(descs):
        n = (pd.rows * pd.kelems + chunk - 1) // chunk
        layer_of += [li] * n
        chunk_index += list(range(n))
    raw = (L.PackDesc * len(descs))(*descs)
    table = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone().cuda()
    lo, ci = torch.tensor(layer_of, dtype=torch.int32).cuda(), torch.tensor(chunk_index, dtype=torch.int32).cuda()
    L.check(lib.din_conv_pack_multi(table.data_ptr(), lo.data_ptr(), ci.data_ptr(), len(layer_of), chunk, None))
    out1 = torch.full((nb, h, w, ld1), 7.0, dtype=tdt, device="cuda")
    out2 = torch.full((nb, h, w, ld2), 7.0, dtype=tdt, device="cuda")
    bdev = bias.cuda()
    craw = couts[0] + couts[1]                               # the last sibling stored raw: no bias, no ReLU (branch_pool conv)
    L.check(lib.din_conv_fwd2(C.byref(d), xin.data_ptr(), bank.data_ptr(), bdev.data_ptr(), out1.data_ptr(), out2.data_ptr(), ld2, off2,
                              couts[0], craw, L.CONV_BIAS | L.CONV_RELU, None, 0, None))
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == "fp32" else 1.5e-2
    if craw:
        ref = torch.cat([ref[:, :craw], F.conv2d(x, ws_[2])], dim=1)
    assert rel(from_nhwc(out1, couts[0], off1), ref[:, :couts[0]]) <= tol
    assert rel(from_nhwc(out2, ctot - couts[0], off2), ref[:, couts[0]:]) <= tol
    assert float(out1[..., :off1].float().min()) == 7.0 and float(out1[..., off1 + couts[0]:].float().min()) == 7.0
    assert float(out2[..., :off2].float().min()) == 7.0 and float(out2[..., off2 + ctot - couts[0]:].float().min()) == 7.0


def test_conv_kernel_variant_names_the_instantiation(env):
    """din_conv_kernel_variant: bit 0 = FASTK, bit 1 = 8 waves, bit 2 = LANEK -- the flags bench.py spells the rocprofv3 kernel name from."""
    lib, L, nhwc, ops = env
    def flags(cin, cout, k, which=0, dtype=None, s=1):
        d = L.ConvDesc()
        d.nb, d.h, d.w, d.cin, d.cout = 96, 43, 78, cin, cout
        d.kh, d.kw = k
        d.sh = d.sw = s
        d.ph, d.pw, d.dh, d.dw = k[0] // 2, k[1] // 2, 1, 1
        d.oh, d.ow = (43 + 2 * d.ph - k[0]) // s + 1, (78 + 2 * d.pw - k[1]) // s + 1
        d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin, 0, cout, 0, (L.DIN_BF16 if dtype is None else dtype)
        f = C.c_int32(-1)
        L.check(lib.din_conv_kernel_variant(C.byref(d), which, C.byref(f)))
        return f.value
    assert flags(192, 192, (7, 1)) == 3                 # whole k-steps per tap, 8-wave 128x192 tile
    assert flags(512, 192, (1, 1)) == 3                 # single tap
    assert flags(768, 192, (1, 1)) == 0                 # (the 768-channel entries left the gather tiles in round 5: conv1x1_regw_kernel, din_conv_kernel_tile code 5)
    assert flags(80, 192, (3, 3)) == 7                  # 80 channels: k-steps straddle taps -> the per-lane k-walk (bit 2, with FASTK), 8 waves
    assert flags(160, 160, (1, 7), which=1) == 2        # ... forward launches only: the data gradient keeps the general loop
    L.set_option("DIN_CONV_LANEK", "0")
    try:
        assert flags(80, 192, (3, 3)) == 2
    finally:
        L.set_option("DIN_CONV_LANEK", None)
    assert flags(192, 192, (7, 1), dtype=L.DIN_F32) == 0


def test_bn_fold_and_wdot(env):
    lib, L, nhwc, ops = env
    c = 24
    g = torch.Generator().manual_seed(3)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    mean, var = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    x = torch.randn(2, 16, 9, 9, generator=g)
    wt = torch.randn(c, 16, 3, 3, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.relu(F.batch_norm(F.conv2d(xr, wr, None, padding=1), mean, var, gr, br, False, 0.0, 1e-3))
    cot = torch.randn(y.shape, generator=g)
    y.backward(cot)
    ga = cot * (y > 0).float()
    dev = lambda t: t.cuda().contiguous()
    gd_, bd_, md_, vd_ = dev(gamma), dev(beta), dev(mean), dev(var)        # keep alive: raw pointers below
    scale, shift = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    L.check(lib.din_bn_fold(gd_.data_ptr(), bd_.data_ptr(), md_.data_ptr(), vd_.data_ptr(), 1e-3,
                            scale.data_ptr(), shift.data_ptr(), c, None))
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = 2, 9, 9, 16, 9, 9, c
    d.kh = d.kw = 3
    d.sh = d.sw = d.ph = d.pw = d.dh = d.dw = 1
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = 16, 0, c, 0, L.DIN_F32
    xin, wdev = to_nhwc(x, torch.float32), dev(wt)
    wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), wdev.data_ptr(), scale.data_ptr(), wpk.data_ptr(), 0, None))
    out = torch.empty(2, 9, 9, c, device="cuda")
    L.check(lib.din_conv_fwd(C.byref(d), xin.data_ptr(), wpk.data_ptr(), shift.data_ptr(), out.data_ptr(), L.CONV_BIAS | L.CONV_RELU, None, 0, None))
    assert rel(from_nhwc(out, c), y) <= 2e-5
    gdev = to_nhwc(ga, torch.float32)
    dw, dshift, wdot = torch.empty_like(wdev), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    wsb = lib.din_conv_workspace_bytes(C.byref(d), 2)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    L.check(lib.din_conv_wgrad(C.byref(d), xin.data_ptr(), gdev.data_ptr(), dw.data_ptr(), dshift.data_ptr(), scale.data_ptr(),
                               wdev.data_ptr(), wdot.data_ptr(), 0, ws.data_ptr(), wsb, None))
    dgamma, dbeta = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    L.check(lib.din_bn_fold_bwd(wdot.data_ptr(), dshift.data_ptr(), md_.data_ptr(), vd_.data_ptr(), 1e-3,
                                dgamma.data_ptr(), dbeta.data_ptr(), c, None))
    torch.cuda.synchronize()
    assert rel(dw, wr.grad) <= 5e-5 and rel(dgamma, gr.grad) <= 5e-5 and rel(dbeta, br.grad) <= 5e-5


@pytest.mark.parametrize("ldi,cioff", [(24, 4), (32, 8)])      # (32, 8) takes the 16-byte-vector bf16 kernels
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("kind,k,s,p", [("maxpool", 2, 2, 0), ("maxpool", 3, 2, 0), ("avgpool", 3, 1, 1)])
def test_pools(env, kind, k, s, p, dtype, ldi, cioff):
    lib, L, nhwc, ops = env
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(k * 10 + s)
    x = F.relu(torch.randn(2, 16, 13, 15, generator=g))            # post-ReLU input with ties at zero
    if dtype == "bf16":
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, k, s, p) if kind == "maxpool" else F.avg_pool2d(xr, k, s, p)
    cot = torch.randn(y.shape, generator=g)
    if dtype == "bf16":
        cot = cot.bfloat16().float()
    y.backward(cot)
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = 2, 13, 15, 16, y.shape[2], y.shape[3]
    d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff = k, s, p, ldi, cioff, 16, 0
    d.dtype = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    xin = to_nhwc(x, tdt, ldi, cioff)
    out = torch.empty(2, y.shape[2], y.shape[3], 16, dtype=tdt, device="cuda")
    amax = torch.empty(2, y.shape[2], y.shape[3], 16, dtype=torch.uint8, device="cuda")
    if kind == "maxpool":
        L.check(lib.din_maxpool_fwd(C.byref(d), xin.data_ptr(), out.data_ptr(), amax.data_ptr(), None))
    else:
        L.check(lib.din_avgpool_fwd(C.byref(d), xin.data_ptr(), out.data_ptr(), None, 0, None))
    tol = 1e-6 if dtype == "fp32" else 8e-3
    assert rel(from_nhwc(out, 16), y) <= tol
    if kind == "avgpool":                                                 # fused bias + ReLU epilogue (commuted 1x1 conv) and colsum
        bias = torch.randn(16, generator=g).cuda()
        L.check(lib.din_avgpool_fwd(C.byref(d), xin.data_ptr(), out.data_ptr(), bias.data_ptr(), L.CONV_BIAS | L.CONV_RELU, None))
        assert rel(from_nhwc(out, 16), F.relu(y + bias.cpu().view(1, -1, 1, 1))) <= tol
        cs = torch.empty(16, device="cuda")
        L.check(lib.din_colsum(xin.data_ptr(), d.dtype, 2 * 13 * 15, 16, ldi, cioff, cs.data_ptr(), None))
        assert rel(cs, x.sum(dim=(0, 2, 3))) <= (1e-5 if dtype == "fp32" else 1e-5)
    gout = to_nhwc(cot, tdt)
    dx = torch.zeros_like(xin)
    if kind == "maxpool":
        L.check(lib.din_maxpool_bwd(C.byref(d), xin.data_ptr(), None, gout.data_ptr(), dx.data_ptr(), 1, 0, None))
        torch.cuda.synchronize()
        want = xr.grad * (x > 0).float()
        assert rel(from_nhwc(dx, 16, cioff), want) <= (1e-6 if dtype == "fp32" else 1.5e-2)
        dx.zero_()                                                        # same answer from the saved arg-max map
        L.check(lib.din_maxpool_bwd(C.byref(d), None, amax.data_ptr(), gout.data_ptr(), dx.data_ptr(), 1, 0, None))
    else:
        L.check(lib.din_avgpool_bwd(C.byref(d), gout.data_ptr(), dx.data_ptr(), xin.data_ptr(), 0, None))
    torch.cuda.synchronize()
    want = xr.grad * (x > 0).float()
    assert rel(from_nhwc(dx, 16, cioff), want) <= (1e-6 if dtype == "fp32" else 1.5e-2)


def test_bilinear_align_corners(env):
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 7, 11, generator=g)
    xr = x.clone().requires_grad_(True)
    y = F.interpolate(xr, size=(15, 23), mode="bilinear", align_corners=True)
    assert rel(O.bilinear_resize_align_corners(x, 15, 23), y) <= 1e-6        # oracle restatement vs ATen
    cot = torch.randn(y.shape, generator=g)
    y.backward(cot)
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = 2, 7, 11, 8, 15, 23
    d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = 1, 1, 0, 8, 0, 16, 8, L.DIN_F32
    xin = to_nhwc(x, torch.float32)
    out = torch.zeros(2, 15, 23, 16, device="cuda")
    L.check(lib.din_bilinear_fwd(C.byref(d), xin.data_ptr(), out.data_ptr(), None))
    assert rel(from_nhwc(out, 8, 8), y) <= 2e-6
    gout = to_nhwc(cot, torch.float32, 16, 8)
    dx = torch.empty_like(xin)
    L.check(lib.din_bilinear_bwd(C.byref(d), gout.data_ptr(), dx.data_ptr(), None, 0, None))
    torch.cuda.synchronize()
    assert rel(from_nhwc(dx, 8), xr.grad) <= 5e-6


@pytest.mark.parametrize("shape", [(2, 43, 78, 87, 157, 768, 1056, 288, "bf16"), (3, 5, 9, 5, 31, 16, 16, 0, "fp32"), (1, 7, 4, 20, 4, 8, 24, 8, "fp32")],
                         ids=["fuse_43x78_to_87x157_bf16", "x_only", "y_only"])
def test_bilinear_cells_kernel_is_bit_identical(env, shape, monkeypatch):
    """the up-sampling forward (one thread per source cell, corners loaded once for all of the cell's outputs) writes every output pixel
    exactly once and reproduces the per-output kernel bit for bit"""
    lib, L, nhwc, ops = env
    nb, h, w, oh, ow, c, ldo, coff, dtype = shape
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(5)
    x = torch.randn(nb, h, w, c, generator=g).to(tdt).cuda()
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = nb, h, w, c, oh, ow
    d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff = 1, 1, 0, c, 0, ldo, coff
    d.dtype = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("DIN_BILINEAR_CELLS", mode)
        out = torch.full((nb, oh, ow, ldo), float("nan"), dtype=tdt, device="cuda")
        L.check(lib.din_bilinear_fwd(C.byref(d), x.data_ptr(), out.data_ptr(), None))
        torch.cuda.synchronize()
        outs.append(out)
    view = slice(coff, coff + c)
    assert not torch.isnan(outs[1][..., view].float()).any(), "an output pixel was not written"
    assert torch.equal(outs[0][..., view], outs[1][..., view])
    if ldo > c:
        rest = torch.cat([outs[1][..., :coff], outs[1][..., coff + c:]], -1)
        assert torch.isnan(rest.float()).all(), "wrote outside its channel range"
    # backward: the hoisted candidate scan against the nested form, with the ReLU mask and with accumulation
    gout = torch.randn(nb, oh, ow, ldo, generator=g).to(tdt).cuda()
    base = torch.randn(nb, h, w, c, generator=g).to(tdt).cuda()
    res = []
    for mode in ("0", "1"):
        monkeypatch.setenv("DIN_BILINEAR_HOIST", mode)
        dx = base.clone()
        L.check(lib.din_bilinear_bwd(C.byref(d), gout.data_ptr(), dx.data_ptr(), x.data_ptr(), 1, None))
        torch.cuda.synchronize()
        res.append(dx)
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("shape", [(2, 43, 78, 192, 768, 576, "bf16"), (3, 15, 23, 64, 288, 224, "bf16"), (2, 9, 5, 16, 16, 0, "fp32"),
                                   (1, 3, 1, 8, 24, 8, "fp32")], ids=["6e_pool_bf16", "5d_pool_bf16", "h_gt_strip_fp32", "one_column_fp32"])
def test_avgpool_strip_kernel_is_bit_identical(env, shape, monkeypatch):
    """the column-strip 3x3 box filter (taps of R + 2 rows kept in registers) sums every output in the order of the one-thread-per-output
    kernel (= ATen's avg_pool2d loop), forward (+ bias, ReLU, strided destination view) and backward (+ ReLU mask, accumulate)"""
    lib, L, nhwc, ops = env
    nb, h, w, c, ldo, coff, dtype = shape
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(31)
    x = torch.randn(nb, h, w, c, generator=g).to(tdt).cuda()
    bias = torch.randn(c, generator=g).cuda()
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = nb, h, w, c, h, w
    d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff = 3, 1, 1, c, 0, ldo, coff
    d.dtype = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    for flags, b in ((0, None), (L.CONV_BIAS | L.CONV_RELU, bias.data_ptr())):
        outs = []
        for mode in ("0", "1"):
            monkeypatch.setenv("DIN_AVGPOOL_STRIP", mode)
            out = torch.full((nb, h, w, ldo), float("nan"), dtype=tdt, device="cuda")
            L.check(lib.din_avgpool_fwd(C.byref(d), x.data_ptr(), out.data_ptr(), b, flags, None))
            torch.cuda.synchronize()
            outs.append(out)
        view = slice(coff, coff + c)
        assert not torch.isnan(outs[1][..., view].float()).any(), "an output pixel was not written"
        assert torch.equal(outs[0][..., view], outs[1][..., view])
        if ldo > c:
            rest = torch.cat([outs[1][..., :coff], outs[1][..., coff + c:]], -1)
            assert torch.isnan(rest.float()).all(), "wrote outside its channel range"
    gout = torch.randn(nb, h, w, ldo, generator=g).to(tdt).cuda()
    base = torch.randn(nb, h, w, c, generator=g).to(tdt).cuda()
    for acc, mask in ((0, None), (1, x.data_ptr())):
        res = []
        for mode in ("0", "1"):
            monkeypatch.setenv("DIN_AVGPOOL_STRIP", mode)
            dx = base.clone()
            L.check(lib.din_avgpool_bwd(C.byref(d), gout.data_ptr(), dx.data_ptr(), mask, acc, None))
            torch.cuda.synchronize()
            res.append(dx)
        assert torch.equal(res[0], res[1])


def test_image_layer_reads_uint8_frames_bit_identically(env):
    """din_conv_desc.in_u8: Inception's Conv2d_1a (3 -> 32, 3x3 stride 2) fed with the raw uint8 clip batch [nb,3,h,w] -- forward and weight /
    bias gradient -- equals din_prep_images_nhwc + the same launches on the prepared tensor BIT FOR BIT (the loader writes the same bf16
    halo image the LDS-DMA would have); ragged tile edges; and the planner refuses layers the image-layer kernels do not serve"""
    lib, L, nhwc, ops = env
    nb, h, w, cout = 2, 727, 1283, 32
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    g = torch.Generator().manual_seed(12)
    img = torch.randint(0, 256, (nb, 3, h, w), dtype=torch.uint8, generator=g).cuda()
    wt = (torch.randn(cout, 3, 3, 3, generator=g) * 0.3).cuda()
    bias = (torch.randn(cout, generator=g) * 0.1).cuda()
    gz = torch.randn(nb, oh, ow, cout, generator=g).bfloat16().cuda()
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, 3, oh, ow, cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = 3, 3, 2, 2, 0, 0, 1, 1
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = 8, 0, cout, 0, L.DIN_BF16
    assert lib.din_conv_accepts_u8(C.byref(d)) == 1
    wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), wt.data_ptr(), None, wpk.data_ptr(), 0, None))
    prepared = torch.empty(nb, h, w, 8, dtype=torch.bfloat16, device="cuda")
    L.check(lib.din_prep_images_nhwc(img.data_ptr(), 1, prepared.data_ptr(), L.DIN_BF16, nb, h, w, 8, None))
    res = []
    for u8 in (0, 1):
        d.in_u8 = u8
        src = img if u8 else prepared
        out = torch.full((nb, oh, ow, cout), 7.0, dtype=torch.bfloat16, device="cuda")
        wsb = lib.din_conv_workspace_bytes(C.byref(d), 0)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        L.check(lib.din_conv_fwd(C.byref(d), src.data_ptr(), wpk.data_ptr(), bias.data_ptr(), out.data_ptr(), L.CONV_BIAS | L.CONV_RELU,
                                 ws.data_ptr(), wsb, None))
        dw, db = torch.empty_like(wt), torch.empty(cout, device="cuda")
        wsb = lib.din_conv_workspace_bytes(C.byref(d), 2)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        L.check(lib.din_conv_wgrad(C.byref(d), src.data_ptr(), gz.data_ptr(), dw.data_ptr(), db.data_ptr(), None, None, None, 0,
                                   ws.data_ptr(), wsb, None))
        torch.cuda.synchronize()
        res.append((out, dw, db))
    assert torch.equal(res[0][0], res[1][0]), "forward differs"
    assert torch.equal(res[0][1], res[1][1]), "weight gradient differs"
    assert rel(res[0][2], res[1][2]) <= 1e-4, "bias gradient differs"          # (summed with fp32 atomics across workgroups: order-dependent last bits)
    # against torch on the normalised image (what the layer computes at all)
    x = O.prep_images(img.cpu().float()).bfloat16().float()
    y = F.relu(F.conv2d(x, wt.cpu().bfloat16().float(), bias.cpu(), stride=2))
    assert rel(res[1][0].float().cpu().permute(0, 3, 1, 2), y) <= 1.5e-2
    # layers the image-layer kernels do not serve: small frames, stride 1, fp32
    d.in_u8 = 0
    d.nb = 1; d.h, d.w, d.oh, d.ow = 139, 203, 69, 101
    assert lib.din_conv_accepts_u8(C.byref(d)) == 0
    d.in_u8 = 1
    small = torch.zeros(1, 3, 139, 203, dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 69, 101, cout, dtype=torch.bfloat16, device="cuda")
    assert lib.din_conv_fwd(C.byref(d), small.data_ptr(), wpk.data_ptr(), None, out.data_ptr(), 0, None, 0, None) != 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_tiled_1x1_filter_pack_is_bit_identical(env, dtype, monkeypatch):
    """the 64 x 64-tile pack of large 1x1 filter banks (fc_emb_1: 26400 x 1024, both orientations, every step) writes exactly what the
    element-per-thread pack writes, padded rows / columns included, with and without the folded scale"""
    lib, L, nhwc, ops = env
    cout, cin = 300, 5003 if dtype == "fp32" else 5000
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(3)
    w = torch.randn(cout, cin, 1, 1, generator=g).cuda()
    scale = (torch.rand(cout, generator=g) + 0.5).cuda()
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = 1, 1, 64, cin, 1, 64, cout
    d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1
    epc = 4 if dtype == "fp32" else 8
    d.ldi, d.ldo, d.dtype = (cin + epc - 1) // epc * epc, (cout + 7) // 8 * 8, L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    for transposed in (0, 1):
        for sc in (None, scale):
            outs = []
            for mode in ("0", "1"):
                monkeypatch.setenv("DIN_PACK_TILES", mode)
                out = torch.full((lib.din_conv_packed_elems(C.byref(d), transposed),), 3.0, dtype=tdt, device="cuda")
                L.check(lib.din_conv_pack_weights(C.byref(d), w.data_ptr(), sc.data_ptr() if sc is not None else None, out.data_ptr(), transposed, None))
                torch.cuda.synchronize()
                outs.append(out)
            assert torch.equal(outs[0], outs[1]), (transposed, sc is not None)


@pytest.mark.parametrize("shape", [(2, 35, 47, 64), (3, 16, 9, 192), (1, 3, 3, 8)], ids=["ragged", "wide", "one_window"])
def test_maxpool_strip_kernel_is_bit_identical(env, shape, monkeypatch):
    """the column-strip 3x3 / stride-2 max pool (2R + 1 input rows in registers per R outputs) returns the values AND the arg-max bytes of the
    one-thread-per-output kernel (same scan order, first maximum wins, 255 for non-positive winners)"""
    lib, L, nhwc, ops = env
    nb, h, w, c = shape
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    g = torch.Generator().manual_seed(13)
    x = torch.randn(nb, h, w, c, generator=g).relu().bfloat16().cuda()          # post-ReLU: many ties at zero
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = nb, h, w, c, oh, ow
    d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = 3, 2, 0, c, 0, c, 0, L.DIN_BF16
    res = []
    monkeypatch.setenv("DIN_MAXPOOL_ROWS", "0")                                   # (the row kernel would take these launches first)
    for mode in ("0", "2"):                                                       # never / always (the default picks by channel count)
        monkeypatch.setenv("DIN_MAXPOOL_STRIP", mode)
        out = torch.full((nb, oh, ow, c), float("nan"), dtype=torch.bfloat16, device="cuda")
        am = torch.full((nb, oh, ow, c), 77, dtype=torch.uint8, device="cuda")
        L.check(lib.din_maxpool_fwd(C.byref(d), x.data_ptr(), out.data_ptr(), am.data_ptr(), None))
        torch.cuda.synchronize()
        res.append((out, am))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[1][0].float().cpu().permute(0, 3, 1, 2), F.max_pool2d(x.float().cpu().permute(0, 3, 1, 2), 3, 2))


@pytest.mark.parametrize("shape,ldi,cioff,ldo,cooff", [((2, 35, 47, 64), 64, 0, 64, 0), ((3, 16, 9, 192), 200, 8, 208, 16), ((1, 3, 3, 8), 8, 0, 8, 0),
                                                       ((2, 36, 48, 32), 40, 8, 32, 0), ((1, 7, 600, 16), 16, 0, 16, 0)],
                         ids=["ragged", "wide_views", "one_window", "even_uncovered_edge", "long_row"])
def test_maxpool_row_kernels_are_bit_identical(env, shape, ldi, cioff, ldo, cooff, monkeypatch):
    """the one-workgroup-per-row 3x3 / stride-2 max pool (forward: v_max3_f32 on (bf16 << 16) | (15 - tap) words; backward: scalar row bases)
    returns the values, the arg-max bytes and the input gradients of the element-per-thread kernels (bytes and gradients bit for bit), on strided channel views, with
    and without accumulation, and the forward matches F.max_pool2d"""
    lib, L, nhwc, ops = env
    nb, h, w, c = shape
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    g = torch.Generator().manual_seed(17)
    xf = torch.randn(nb, h, w, c, generator=g)
    xf[0, : h // 2] = xf[0, : h // 2].relu()                                     # ties at zero in one half, negative winners in the other
    xf[-1, :, : w // 2] = -xf[-1, :, : w // 2].abs()
    xbuf = torch.zeros(nb, h, w, ldi, dtype=torch.bfloat16, device="cuda")
    xbuf[..., cioff:cioff + c] = xf.bfloat16().cuda()
    gout = torch.zeros(nb, oh, ow, ldo, dtype=torch.bfloat16, device="cuda")
    gout[..., cooff:cooff + c] = torch.randn(nb, oh, ow, c, generator=g).bfloat16().cuda()
    base = torch.randn(nb, h, w, ldi, generator=g).bfloat16().cuda()
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = nb, h, w, c, oh, ow
    d.k, d.stride, d.pad, d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = 3, 2, 0, ldi, cioff, ldo, cooff, L.DIN_BF16
    res = []
    for mode in ("0", "1"):                                                       # element-per-thread kernels / row kernels (the default)
        monkeypatch.setenv("DIN_MAXPOOL_ROWS", mode)
        monkeypatch.setenv("DIN_MAXPOOL_STRIP", "0")
        out = torch.full((nb, oh, ow, ldo), 5.0, dtype=torch.bfloat16, device="cuda")
        am = torch.full((nb, oh, ow, c), 77, dtype=torch.uint8, device="cuda")
        L.check(lib.din_maxpool_fwd(C.byref(d), xbuf.data_ptr(), out.data_ptr(), am.data_ptr(), None))
        dx0 = torch.full_like(xbuf, 3.0)
        L.check(lib.din_maxpool_bwd(C.byref(d), None, am.data_ptr(), gout.data_ptr(), dx0.data_ptr(), 1, 0, None))
        dx1 = base.clone()
        L.check(lib.din_maxpool_bwd(C.byref(d), None, am.data_ptr(), gout.data_ptr(), dx1.data_ptr(), 1, 1, None))
        torch.cuda.synchronize()
        res.append((out, am, dx0, dx1))
    # values: equal as numbers -- a window holding both -0.0 and +0.0 and nothing positive pools to whichever zero comes first in the scan
    # kernels and to +0.0 in the max3 kernel (its arg-max byte is 255 either way); everything else: the same bits
    assert torch.equal(res[0][0].float(), res[1][0].float()), "values"
    for a, b, what in zip(res[0][1:], res[1][1:], ("arg-max bytes", "gradient", "accumulated gradient")):
        assert torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a, b.view(torch.uint8) if b.dtype != torch.uint8 else b), what
    want = F.max_pool2d(xf.bfloat16().float().permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    assert torch.equal(res[1][0][..., cooff:cooff + c].float().cpu(), want)
    assert float((res[1][0][..., :cooff].float() - 5.0).abs().sum()) == 0.0      # channels outside the view untouched
    if res[1][1].numel() > 64:
        assert int((res[1][1] != 255).sum()) > 0 and int((res[1][1] == 255).sum()) > 0


def test_prep_images_bit_exact(env):
    lib, L, nhwc, ops = env
    x = torch.arange(0, 256, dtype=torch.float32)
    y = ops.prep_images_f32(x.cuda()).cpu()
    assert torch.equal(y, O.prep_images(x))
    img = torch.randint(0, 256, (2, 3, 5, 7), dtype=torch.uint8)
    out = torch.empty(2, 5, 7, 4, device="cuda")
    L.check(lib.din_prep_images_nhwc(img.cuda().data_ptr(), 1, out.data_ptr(), L.DIN_F32, 2, 5, 7, 4, None))
    torch.cuda.synchronize()
    assert torch.equal(out[..., :3].cpu(), O.prep_images(img.float()).permute(0, 2, 3, 1))
    assert float(out[..., 3].abs().sum()) == 0.0


@pytest.mark.parametrize("k", [5, 1, 3])
def test_roi_align_index_bit_exact_and_values(env, k):
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(5)
    nb, c, hf, wf, n = 3, 96, 22, 40, 12
    fm = torch.randn(nb, c, hf, wf, generator=g)
    _, boxes, _ = O.synth_inputs(1, nb, n, 8, 8, hf, wf, seed=7)
    boxes = boxes.reshape(nb * n, 4).clone()
    boxes[5] = 0.0                                     # Collective padding box -> out of range -> zeros
    boxes[6] = torch.tensor([-3.0, 2.0, 50.0, 30.0])   # partially outside
    boxes[7] = torch.tensor([4.0, 4.0, 9.0, 9.0])      # integer-aligned samples (floor == ceil)
    ind = O.boxes_frame_index(nb, n)
    fmr = fm.clone().requires_grad_(True)
    ref, ridx = O.roi_align(fmr, boxes, ind, k, return_index=True)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    fmd = to_nhwc(fm, torch.float32).requires_grad_(True)
    out, idx = ops.RoIAlignFunction.apply(fmd, boxes.cuda(), ind.cuda(), k, c, False, True)
    idx = idx.cpu()
    # integer decisions: bit-exact
    assert torch.equal(idx[:, :, 0, 0], ridx["top"]) and torch.equal(idx[:, :, 0, 1], ridx["bot"])
    assert torch.equal(idx[:, 0, :, 2], ridx["left"]) and torch.equal(idx[:, 0, :, 3], ridx["right"])
    assert torch.equal(idx[:, :, 0, 4].bool(), ridx["oob_y"]) and torch.equal(idx[:, 0, :, 5].bool(), ridx["oob_x"])
    assert torch.equal(out.detach().cpu(), ref.detach()), "RoIAlign values must be bit-exact in fp32 (same op order)"
    out.backward(cot.cuda())
    assert rel(fmd.grad.cpu().permute(0, 3, 1, 2), fmr.grad) <= 1e-5
    assert torch.equal(ops.boxes_frame_index(nb, n, "cuda").cpu(), ind)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_roi_align_bwd_gather_matches_scatter(env, dtype):
    """gather-form backward (any box order, > 256 boxes in one frame -> two batches, ReLU mask) == atomic scatter + cast/mask"""
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(21)
    nb, c, hf, wf, k, m = 3, 64, 22, 40, 5, 330
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    dt = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    x1 = torch.rand(m, generator=g) * (wf - 8); y1 = torch.rand(m, generator=g) * (hf - 8)
    boxes = torch.stack([x1, y1, x1 + 1 + torch.rand(m, generator=g) * 9, y1 + 1 + torch.rand(m, generator=g) * 9], 1)
    boxes[3] = 0.0
    boxes[4] = torch.tensor([-3.0, 2.0, 50.0, 30.0])
    ind = torch.zeros(m, dtype=torch.int32)                       # 300 boxes in frame 0 (two batches), the rest interleaved
    ind[300:] = torch.randint(1, nb, (m - 300,), generator=g, dtype=torch.int32)
    perm = torch.randperm(m, generator=g)
    boxes, ind = boxes[perm].contiguous(), ind[perm].contiguous()
    fm = torch.randn(nb, hf, wf, c, generator=g).to(tdt).cuda()
    gout = torch.randn(m, c, k, k, generator=g).cuda()
    bd, idv = boxes.cuda(), ind.cuda()
    g32 = torch.zeros(nb, hf, wf, c, device="cuda")
    L.check(lib.din_roi_align_bwd(gout.data_ptr(), nb, hf, wf, c, bd.data_ptr(), idv.data_ptr(), m, k, g32.data_ptr(), None))
    want = g32 * (fm.float() > 0).float()
    outs = []
    for staged in (False, True):                               # reference-layout crop gradient / channel-contiguous staging copy
        got = torch.full((nb, hf, wf, c), 9.0, dtype=tdt, device="cuda")
        src = gout
        if staged:
            src = torch.empty_like(gout)
            L.check(lib.din_roi_crop_grad_transpose(gout.data_ptr(), m, c, k, src.data_ptr(), None))
            assert torch.equal(src.view(m, k * k, c), gout.view(m, c, k * k).transpose(1, 2))
        L.check(lib.din_roi_align_bwd_nhwc(src.data_ptr(), c, 0, int(staged), nb, hf, wf, c, hf, wf, bd.data_ptr(), idv.data_ptr(), m, k,
                                           fm.data_ptr(), dt, c, got.data_ptr(), c, None))
        torch.cuda.synchronize()
        assert rel(got.float(), want) <= (1e-5 if dtype == "fp32" else 1e-2)      # bf16: one rounding per batch
        assert bool(((want == 0) == (got.float() == 0)).all()), "footprint (zeros outside the boxes, mask) must match"
        outs.append(got)
    assert torch.equal(outs[0], outs[1]), "the staged copy must not change the summation order"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("k", [5, 2])
def test_roi_align_multiscale_equals_resize_cat_crop(env, dtype, k):
    """RoIAlign through the virtual resize (ops.RoIAlignMultiScaleFunction) == RoIAlign(torch.cat([a, F.interpolate(b, size, 'bilinear',
    align_corners=True)], 1)) -- the reference's multi-scale fuse, infer_model.py:165-180 -- forward and both maps' gradients (with the maps'
    ReLU masks), incl. boxes outside / on the border / integer-aligned, and a grid that is not a multiple of the stored map."""
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(77 + k)
    nb, ca, cb, gh, gw, hb, wb, n = 3, 32, 64, 23, 41, 11, 20, 12
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    fa = torch.randn(nb, ca, gh, gw, generator=g).to(tdt).float()
    fb = torch.randn(nb, cb, hb, wb, generator=g).to(tdt).float()
    _, boxes, _ = O.synth_inputs(1, nb, n, 8, 8, gh, gw, seed=9)
    boxes = boxes.reshape(nb * n, 4).clone()
    boxes[5] = 0.0
    boxes[6] = torch.tensor([-3.0, 2.0, 50.0, 30.0])
    boxes[7] = torch.tensor([4.0, 4.0, 9.0, 9.0])
    boxes[8] = torch.tensor([30.0, 15.0, 40.0, 22.0])              # touches the last row / column of the grid
    ind = O.boxes_frame_index(nb, n)
    far, fbr = fa.clone().requires_grad_(True), fb.clone().requires_grad_(True)
    fused = torch.cat([far, F.interpolate(fbr, size=(gh, gw), mode="bilinear", align_corners=True)], 1)
    ref = O.roi_align(fused, boxes, ind, k)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    fad = to_nhwc(fa, tdt).requires_grad_(True)
    fbd = to_nhwc(fb, tdt).requires_grad_(True)
    for masked in (False, True):
        fad.grad = fbd.grad = None
        out = ops.RoIAlignMultiScaleFunction.apply(boxes.cuda(), ind.cuda(), k, (gh, gw), (ca, cb), (masked, masked), fad, fbd)
        assert out.shape == ref.shape
        assert torch.equal(out[:, :ca].detach().cpu(), ref[:, :ca].detach()), "the unresized map keeps the bit-exact plain sampling"
        assert rel(out[:, ca:], ref[:, ca:]) <= 2e-6
        out.backward(cot.cuda())
        wa = far.grad * ((fa > 0).float() if masked else 1.0)
        wb = fbr.grad * ((fb > 0).float() if masked else 1.0)
        tol = 2e-5 if dtype == "fp32" else 1e-2
        assert rel(fad.grad.float().cpu().permute(0, 3, 1, 2), wa) <= tol
        assert rel(fbd.grad.float().cpu().permute(0, 3, 1, 2), wb) <= tol
        assert bool(((wb == 0) == (fbd.grad.float().cpu().permute(0, 3, 1, 2) == 0)).all()), "footprint of the composed backward"


@pytest.mark.parametrize("shape", [(3, 12, 15, 4, 128), (2, 12, 3600, 4, 128), (2, 5, 70, 2, 32), (1, 16, 257, 1, 256)],
                         ids=["small", "vgg_720p_map", "odd", "n16_c256"])
def test_context_attention_matches_torch(env, shape):
    """ops.ContextAttentionFunction (scores, row softmax, weighted sum) against torch on the CPU: attention map, output and
    both gradients"""
    lib, L, nhwc, ops = env
    bt, n, p, heads, c = shape
    g = torch.Generator().manual_seed(bt * 100 + p)
    q = torch.randn(bt, n, heads * c, generator=g) * 0.3
    kf = torch.randn(bt, p, heads * c, generator=g) * 0.3
    cot = torch.randn(bt, n, heads * c, generator=g)
    qr, kr = q.clone().requires_grad_(True), kf.clone().requires_grad_(True)
    qh = qr.reshape(bt, n, heads, c).permute(0, 2, 1, 3)                     # [bt,h,n,c]
    kh = kr.reshape(bt, p, heads, c).permute(0, 2, 1, 3)                     # [bt,h,p,c]
    att_ref = torch.softmax(qh @ kh.transpose(2, 3), dim=3)                  # [bt,h,n,p]
    out_ref = (att_ref @ kh).permute(0, 2, 1, 3).reshape(bt, n, heads * c)
    out_ref.backward(cot)
    qd, kd = q.cuda().requires_grad_(True), kf.cuda().requires_grad_(True)
    out, att = ops.ContextAttentionFunction.apply(qd, kd, heads)
    out.backward(cot.cuda())
    assert rel(att, att_ref) <= 2e-5 and rel(out, out_ref) <= 2e-5
    assert Measured(abs(float(att.sum(-1).mean()) - 1.0)) <= 1e-5
    assert rel(qd.grad, qr.grad) <= 5e-5 and rel(kd.grad, kr.grad) <= 5e-5


def test_add_position_and_act_dropout(env):
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(4)
    pos = torch.randn(3, 5, 8, generator=g)
    for tdt in (torch.float32, torch.bfloat16):
        x = torch.randn(4, 3, 5, 8, generator=g).to(tdt)
        xd = x.cuda().requires_grad_(True)
        y = ops.AddPositionFunction.apply(xd, pos.cuda(), True)
        assert y.dtype == torch.float32 and torch.equal(y.cpu(), x.float() + pos)
        cot = torch.randn(4, 3, 5, 8, generator=g)
        y.backward(cot.cuda())
        assert xd.grad.dtype == tdt and torch.equal(xd.grad.cpu(), (cot * (x.float() > 0)).to(tdt))
    x = torch.randn(64, 512, generator=g).cuda().requires_grad_(True)
    y0 = ops.ActDropoutFunction.apply(x, True, 0.0, 1)
    assert torch.equal(y0, torch.relu(x))
    y = ops.ActDropoutFunction.apply(x, True, 0.25, 77)
    kept = (y != 0) | (x <= 0)
    frac = float(((y != 0).sum()) / (x > 0).sum())
    assert 0.70 <= frac <= 0.80 and rel(y[y != 0], torch.relu(x)[y != 0] / 0.75) <= 1e-6
    y.sum().backward()
    assert torch.equal(x.grad != 0, y != 0)                       # the backward regenerates the same mask
    assert torch.equal(ops.ActDropoutFunction.apply(x, True, 0.25, 77), y) and not torch.equal(ops.ActDropoutFunction.apply(x, True, 0.25, 78), y)


def test_layernorm_variants(env):
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(11)
    for shape, norm in [((72, 1024), (1024,)), ((2, 3, 12, 128), (3, 12, 128)), ((5, 10, 64), (10, 64))]:
        x = torch.randn(shape, generator=g)
        r = torch.randn(shape, generator=g)
        gamma, beta = torch.rand(norm, generator=g) + 0.5, torch.randn(norm, generator=g)
        for use_res in (False, True):
            xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
            gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            y = F.relu(F.layer_norm(xr + rr if use_res else xr, norm, gr, br, 1e-5))
            cot = torch.randn(shape, generator=g)
            y.backward(cot)
            xd, rd = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
            gd, bd = gamma.cuda().requires_grad_(True), beta.cuda().requires_grad_(True)
            yd = ops.layer_norm(xd, gd, bd, res=rd if use_res else None, relu=True)
            yd.backward(cot.cuda())
            assert rel(yd, y) <= 1e-5
            assert rel(xd.grad, xr.grad) <= 1e-4 and rel(gd.grad, gr.grad) <= 1e-4 and rel(bd.grad, br.grad) <= 1e-4
            if use_res:
                assert rel(rd.grad, rr.grad) <= 1e-4
    # dropout: same mask forward/backward, keep-rate ~ 1-p, survivors scaled by 1/(1-p)
    x = torch.randn(4, 4096, generator=g).cuda().requires_grad_(True)
    ga, be = torch.ones(4096, device="cuda"), torch.zeros(4096, device="cuda")
    y0 = ops.layer_norm(x, ga, be, relu=False, drop_p=0.0)
    y1 = ops.layer_norm(x, ga, be, relu=False, drop_p=0.3, seed=1234)
    keep = (y1 != 0)
    assert Measured(abs(keep.float().mean().item() - 0.7)) < 0.02
    assert rel(y1[keep], y0[keep] / 0.7) <= 1e-5
    y1b = ops.layer_norm(x, ga, be, relu=False, drop_p=0.3, seed=1234)
    assert torch.equal(y1, y1b)


def test_head_and_adam(env):
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(13)
    b, t, n, c, a = 2, 3, 12, 256, 8
    s = torch.randn(b, t, n, c, generator=g)
    w, bias = torch.randn(a, c, generator=g) * 0.1, torch.randn(a, generator=g)
    sr, wr, br = s.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    pooled = sr.max(dim=2).values
    ref = F.linear(pooled.reshape(b * t, c), wr, br).reshape(b, t, a).mean(1)
    cot = torch.randn(b, a, generator=g)
    ref.backward(cot)
    sd, wd, bd = s.cuda().requires_grad_(True), w.cuda().requires_grad_(True), bias.cuda().requires_grad_(True)
    out = ops.HeadFunction.apply(sd, wd, bd, None)
    out.backward(cot.cuda())
    assert rel(out, ref) <= 1e-5 and rel(sd.grad, sr.grad) <= 1e-5 and rel(wd.grad, wr.grad) <= 1e-5 and rel(bd.grad, br.grad) <= 1e-5
    # Adam vs torch.optim.Adam
    p = torch.randn(1000, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-2, weight_decay=0.01)
    pd, m, v = p.cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        pr.grad = gr.clone()
        opt.step()
        ops.adam_step(pd, gr.cuda(), m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, step)
    assert rel(pd, pr) <= 1e-5
    # FusedAdam: the whole parameter list in one multi-tensor launch (ragged sizes around the 8192-element chunk)
    from din_amd.optim import FusedAdam
    shapes = [(3,), (8192,), (8193,), (70, 300), (1,)]
    ps = [torch.randn(sh, generator=g) for sh in shapes]
    refs = [t.clone().requires_grad_(True) for t in ps]
    devs = [t.cuda().requires_grad_(True) for t in ps]
    ropt = torch.optim.Adam(refs, lr=3e-3, weight_decay=1e-4)
    fopt = FusedAdam(devs, lr=3e-3, weight_decay=1e-4)
    for step in range(3):
        for r_, d_ in zip(refs, devs):
            gr = torch.randn(r_.shape, generator=g)
            r_.grad, d_.grad = gr.clone(), gr.cuda()
        ropt.step(); fopt.step()
    for r_, d_ in zip(refs, devs):
        assert rel(d_.detach(), r_.detach()) <= 1e-5


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "bf16_stream", "bf16_regw", "bf16_regw_768", "bf16_regw_short"])
def test_conv1x1_dgrad_multi_source(env, dtype, monkeypatch):
    """fused dgrad of three 1x1 convs reading the same tensor == sum of the three separate dgrads (+ mask, + accumulate); bf16_stream: through
    the persistent streaming kernel (each source = its own 64-channel blocks, the 48- and 104-channel sources end in partial blocks)"""
    lib, L, nhwc, ops = env
    monkeypatch.setenv("DIN_CONV_STREAM", "2" if dtype == "bf16_stream" else "0")
    regw = dtype.startswith("bf16_regw")
    monkeypatch.setenv("DIN_CONV_REGW", "2" if regw else "0")
    nb, h, w, cin = 2, 9, 13, 288
    couts = [64, 48, 104]
    if regw:
        # bf16_regw: the Mixed_6c / 6d block entry (sources 192 + 160 + 160 + 192: the 160-channel sources end in half-padded stages)
        # -> 768 channels in four classes, through conv1x1_regw_kernel; bf16_regw_768: Mixed_6e (24 k-steps), three tiles per team
        # bf16_regw_short: the Mixed_5c entry (64 + 64 + 48 + 64: every source its own, partly padded, stage) -> 256 channels in two classes of 128
        nb, h, w, cin = {"bf16_regw": (2, 37, 41, 768), "bf16_regw_768": (3, 87, 157, 768), "bf16_regw_short": (3, 87, 157, 256)}[dtype]
        couts = {"bf16_regw": [192, 160, 160, 192], "bf16_regw_768": [192, 192, 192, 192], "bf16_regw_short": [64, 64, 48, 64]}[dtype]
    if dtype in ("bf16_stream", "bf16_regw", "bf16_regw_768", "bf16_regw_short"):
        dtype = "bf16"
    dt = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(21)
    x = torch.randn(nb, cin, h, w, generator=g)
    if dtype == "bf16":
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    total = 0
    srcs = (L.ConvSrc * len(couts))()
    keep = []
    for j, co in enumerate(couts):
        wt = torch.randn(co, cin, 1, 1, generator=g) * 0.1
        gz = torch.randn(nb, co, h, w, generator=g)
        if dtype == "bf16":
            wt, gz = wt.bfloat16().float(), gz.bfloat16().float()
        total = total + (F.conv2d(xr, wt) * gz).sum()
        d = L.ConvDesc()
        d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, h, w, co
        d.kh = d.kw = d.sh = d.sw = d.dh = d.dw = 1
        d.ph = d.pw = 0
        ldo = (co + 7) // 8 * 8 + 8
        d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin, 0, ldo, 8, dt
        wdev = wt.cuda()
        wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=tdt, device="cuda")
        L.check(lib.din_conv_pack_weights(C.byref(d), wdev.data_ptr(), None, wpt.data_ptr(), 1, None))
        gdev = to_nhwc(gz, tdt, ldo, 8)
        keep += [wdev, wpt, gdev]
        srcs[j].dout, srcs[j].wpk_t, srcs[j].cout, srcs[j].ldo, srcs[j].cooff = gdev.data_ptr(), wpt.data_ptr(), co, ldo, 8
    total.backward()
    xin = to_nhwc(x, tdt)
    dx = torch.zeros(nb, h, w, cin, dtype=tdt, device="cuda")
    L.check(lib.din_conv1x1_dgrad_multi(len(couts), srcs, dt, nb, h, w, cin, cin, 0, dx.data_ptr(), None, 0, 0, 0, None))
    torch.cuda.synchronize()
    tol = 5e-5 if dtype == "fp32" else 2e-2
    assert rel(from_nhwc(dx, cin), xr.grad) <= tol
    L.check(lib.din_conv1x1_dgrad_multi(len(couts), srcs, dt, nb, h, w, cin, cin, 0, dx.data_ptr(), xin.data_ptr(), cin, 0,
                                        L.CONV_MASK | L.CONV_ACCUM, None))
    torch.cuda.synchronize()
    assert rel(from_nhwc(dx, cin), xr.grad + xr.grad * (x > 0).float()) <= 2 * tol
    if regw:
        # the launch the block entries make: mask, no accumulate -- and the register-resident kernel against the tile kernel on the same operands
        # (different summation order: not bit-identical; both within the bf16 output rounding of the fp32 reference)
        outs = []
        for mode in ("2", "0"):
            monkeypatch.setenv("DIN_CONV_REGW", mode)
            dx.fill_(5.0)
            L.check(lib.din_conv1x1_dgrad_multi(len(couts), srcs, dt, nb, h, w, cin, cin, 0, dx.data_ptr(), xin.data_ptr(), cin, 0, L.CONV_MASK, None))
            torch.cuda.synchronize()
            assert rel(from_nhwc(dx, cin), xr.grad * (x > 0).float()) <= tol
            outs.append(dx.clone())
        assert bool(((outs[0] == 0) == (outs[1] == 0)).all())               # the same positions are masked
        assert rel(outs[0].float(), outs[1].float()) <= 2.0 ** -7


@pytest.mark.parametrize("case", [("fused_bf16", "bf16", 288, 21, 25, 1, "1"), ("fused_bf16_p1_ragged", "bf16", 96, 20, 27, 1, "1"),
                                  ("fallback_bf16", "bf16", 288, 21, 25, 1, "0"), ("fallback_fp32", "fp32", 288, 13, 15, 0, "1"),
                                  ("fallback_bf16_tile160", "bf16", 160, 21, 25, 0, "1")], ids=lambda c: c[0])
def test_conv_dgrad_x_strided_plus_1x1(env, case, monkeypatch):
    """din_conv_dgrad_x: the dgrad of a 3x3 / stride-2 conv carrying the dgrad of a 1x1 / stride-1 conv that reads the same view (InceptionB:
    Mixed_6a.branch3x3 + branch3x3dbl_1) == autograd of the sum of both, plain / + mask + accumulate; the fused kernel (bf16, 96-wide parity
    tiles: extra k-steps at the output pixel, partial last 64-channel block of the 1x1) and the two-launch form for every other shape."""
    lib, L, nhwc, ops = env
    name, dtype, cin, h, w, pad, switch = case
    monkeypatch.setenv("DIN_DGRAD_X", switch)
    dt = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(33)
    nb, cout, cx = 2, 64, 72                      # 72 channels of the 1x1: one whole 64-channel k-step + a partial one
    x = torch.randn(nb, cin, h, w, generator=g)
    w3 = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    w1 = torch.randn(cx, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5
    if dtype == "bf16":
        x, w3, w1 = x.bfloat16().float(), w3.bfloat16().float(), w1.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    y3 = F.conv2d(xr, w3, stride=2, padding=pad)
    y1 = F.conv2d(xr, w1)
    g3, g1 = torch.randn(y3.shape, generator=g), torch.randn(y1.shape, generator=g)
    if dtype == "bf16":
        g3, g1 = g3.bfloat16().float(), g1.bfloat16().float()
    ((y3 * g3).sum() + (y1 * g1).sum()).backward()
    oh, ow = y3.shape[2:]
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, oh, ow, cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = 3, 3, 2, 2, pad, pad, 1, 1
    ldo3 = cout + 16
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = cin + 8, 8, ldo3, 8, dt
    d1 = L.ConvDesc()
    d1.nb, d1.h, d1.w, d1.cin, d1.oh, d1.ow, d1.cout = nb, h, w, cin, h, w, cx
    d1.kh = d1.kw = d1.sh = d1.sw = d1.dh = d1.dw = 1
    d1.ph = d1.pw = 0
    ldo1 = cx + 24
    d1.ldi, d1.cioff, d1.ldo, d1.cooff, d1.dtype = cin + 8, 8, ldo1, 16, dt
    assert lib.din_conv_dgrad_x_fused(C.byref(d)) == (1 if name.startswith("fused") else 0), "planner: fused kernel / two-launch form"
    wpt3 = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=tdt, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d), w3.cuda().data_ptr(), None, wpt3.data_ptr(), 1, None))
    wpt1 = torch.empty(lib.din_conv_packed_elems(C.byref(d1), 1), dtype=tdt, device="cuda")
    L.check(lib.din_conv_pack_weights(C.byref(d1), w1.cuda().data_ptr(), None, wpt1.data_ptr(), 1, None))
    g3d, g1d = to_nhwc(g3, tdt, ldo3, 8), to_nhwc(g1, tdt, ldo1, 16)
    xs = L.ConvSrc()
    xs.dout, xs.wpk_t, xs.cout, xs.ldo, xs.cooff = g1d.data_ptr(), wpt1.data_ptr(), cx, ldo1, 16
    xin = to_nhwc(x, tdt, cin + 8, 8)
    dx = torch.full((nb, h, w, cin + 8), 3.0, dtype=tdt, device="cuda")
    wsb = lib.din_conv_workspace_bytes(C.byref(d), 1)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
    L.check(lib.din_conv_dgrad_x(C.byref(d), g3d.data_ptr(), wpt3.data_ptr(), dx.data_ptr(), None, 0, 0, 0, C.byref(xs), ws.data_ptr(), wsb, None))
    torch.cuda.synchronize()
    tol = 5e-5 if dtype == "fp32" else 2e-2
    assert rel(from_nhwc(dx, cin, 8), xr.grad) <= tol
    assert float(dx[..., :8].float().min()) == 3.0 and float(dx[..., :8].float().max()) == 3.0, "wrote outside its channel range"
    L.check(lib.din_conv_dgrad_x(C.byref(d), g3d.data_ptr(), wpt3.data_ptr(), dx.data_ptr(), xin.data_ptr(), cin + 8, 8,
                                 L.CONV_MASK | L.CONV_ACCUM, C.byref(xs), ws.data_ptr(), wsb, None))
    torch.cuda.synchronize()
    assert rel(from_nhwc(dx, cin, 8), xr.grad + xr.grad * (x > 0).float()) <= 2 * tol
    if name.startswith("fused"):
        # same bits as the kernel's own reduction order allows: against the two-launch form, within bf16 rounding of the intermediate sum
        monkeypatch.setenv("DIN_DGRAD_X", "0")
        dx2 = torch.zeros((nb, h, w, cin + 8), dtype=tdt, device="cuda")
        L.check(lib.din_conv_dgrad_x(C.byref(d), g3d.data_ptr(), wpt3.data_ptr(), dx2.data_ptr(), None, 0, 0, 0, C.byref(xs), ws.data_ptr(), wsb, None))
        monkeypatch.setenv("DIN_DGRAD_X", "1")
        dx1 = torch.zeros((nb, h, w, cin + 8), dtype=tdt, device="cuda")
        L.check(lib.din_conv_dgrad_x(C.byref(d), g3d.data_ptr(), wpt3.data_ptr(), dx1.data_ptr(), None, 0, 0, 0, C.byref(xs), ws.data_ptr(), wsb, None))
        torch.cuda.synchronize()
        assert rel(from_nhwc(dx1, cin, 8), from_nhwc(dx2, cin, 8)) <= 1e-2


PIPE_CASES = [
    # name, nb, cin, h, w, cout, k, s, p          (bf16 wgrad shapes that the planner gives to conv_wgrad_pipe_kernel)
    ("pipe192_3x3_p0_tail", 2, 80, 61, 97, 192, (3, 3), (1, 1), (0, 0)),          # 192-filter tile, 720 k columns (ragged last k tile), M % 32 != 0
    ("pipe192_7x1", 3, 160, 43, 78, 192, (7, 1), (1, 1), (3, 0)),                 # Mixed_6 7x1: row padding taps
    ("pipe192_1x7_narrow", 4, 64, 20, 24, 192, (1, 7), (1, 1), (0, 3)),           # OW < 32: the general (looping) cursor update
    ("pipe384_3x3_s2", 2, 96, 47, 63, 384, (3, 3), (2, 2), (0, 0)),               # Mixed_6a.branch3x3: two filter tiles, stride 2
    ("pipe128_1x1", 2, 512, 30, 40, 256, (1, 1), (1, 1), (0, 0)),                 # 128-filter tiles
]


@pytest.mark.parametrize("waves", ["16", "8", "4"], ids=["w16", "w8", "w4"])
@pytest.mark.parametrize("atomic", ["0", "1"], ids=["slices", "atomic"])
@pytest.mark.parametrize("case", PIPE_CASES, ids=[c[0] for c in PIPE_CASES])
def test_wgrad_pipe_kernel(env, case, atomic, waves, monkeypatch):
    """The software-pipelined 32x32x16 weight-gradient kernel (conv_wgrad_pipe.hip), every wave grid it is instantiated for, both epilogues (slice partials + reduce, fp32 atomics
    into one tile buffer), against autograd of F.conv2d on bf16-rounded operands; bias gradient (packed dot-product path) included."""
    lib, L, nhwc, ops = env
    name, nb, cin, h, w, cout, k, s, p = case
    monkeypatch.setenv("DIN_WGRAD_PIPE", "1")
    monkeypatch.setenv("DIN_WGRAD_ATOMIC", atomic)
    monkeypatch.setenv("DIN_WGRAD_PIPE_WAVES", waves)        # wave grids 2 x 8 (default), 2 x 4, 2 x 2 of the same tile
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.randn(nb, cin, h, w, generator=g).relu().bfloat16().float()
    wt = torch.randn(cout, cin, *k, generator=g).requires_grad_(True)
    br = torch.zeros(cout, requires_grad=True)
    y = F.conv2d(x, wt, br, stride=s, padding=p)
    oh, ow = y.shape[2:]
    gz = torch.randn(y.shape, generator=g).bfloat16().float()
    y.backward(gz)
    ldi, ldo, coff = cin + 8, cout + 16, 8
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, oh, ow, cout
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = k[0], k[1], s[0], s[1], p[0], p[1], 1, 1
    d.ldi, d.cioff, d.ldo, d.cooff, d.dtype = ldi, 0, ldo, coff, L.DIN_BF16
    bm, bn = C.c_int32(0), C.c_int32(0)
    lib.din_conv_kernel_tile(C.byref(d), 2, C.byref(bm), C.byref(bn))
    assert bn.value >= 2000, f"planner did not pick the pipe kernel for {name}: tile {bm.value} x {bn.value}"
    xin = to_nhwc(x, torch.bfloat16, ldi)
    gzd = to_nhwc(gz, torch.bfloat16, ldo, coff)
    dw = torch.empty(cout, cin, *k, device="cuda")
    db = torch.empty(cout, device="cuda")
    wsb = lib.din_conv_workspace_bytes(C.byref(d), 2)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device="cuda")
    for _ in range(2):                                      # twice: the atomic epilogue must not depend on what the workspace held before
        ws.fill_(0x7f)
        L.check(lib.din_conv_wgrad(C.byref(d), xin.data_ptr(), gzd.data_ptr(), dw.data_ptr(), db.data_ptr(), None, None, None, 0,
                                   ws.data_ptr(), wsb, None))
        torch.cuda.synchronize()
        assert rel(dw, wt.grad) <= 2e-3                     # bf16 operands are exact in the reference too: only fp32 summation order differs
        assert rel(db, br.grad) <= 2e-3


@pytest.mark.parametrize("use_shift", [False, True], ids=["noshift", "shift"])
@pytest.mark.parametrize("offset", [0.4, 300.0], ids=["mean0.4", "mean300"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_batch_stat_bn_kernels(env, dtype, offset, use_shift):
    """din_bn_stats / din_bn_finalize / din_bn_apply / din_bn_bwd_stats / din_bn_bwd_apply on channel views against torch's
    F.batch_norm(training=True) + ReLU in float64 (on the storage-rounded input): output, batch mean / rstd, running statistics
    (momentum 0.1, unbiased variance), dy, dgamma, dbeta.  The statistics are fp64 sums of exact products in a fixed order: the
    |mean| >> std case needs no shift (noshift, and a shift UNRELATED to the batch mean -- the running mean of another domain -- is
    harmless: ADVICE r3), and a second run gives the same BITS."""
    lib, L, nhwc, ops = env
    dt = L.DIN_F32 if dtype == "fp32" else L.DIN_BF16
    tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(7)
    rows, c, ldx, cxoff, ldy, cyoff = 4 * 37 * 53 + 5, 96, 112, 8, 160, 32
    # offset 300: |mean| >> std -- E[x^2] - E[x]^2 cancels 4-5 digits; the fp64 accumulators (exact products) absorb that
    x = (torch.randn(rows, c, generator=g) * 1.7 + offset).to(tdt)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    # the running mean doubles as the shift: for offset 300 it is put 450 AWAY from the batch mean (|shift - mean| > |mean|)
    rmean, rvar = torch.randn(c, generator=g) * 0.1 + (-150.0 if offset > 1 else 0.0), torch.rand(c, generator=g) + 0.5
    cot = torch.randn(rows, c, generator=g).to(tdt)
    # reference in float64
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm64, rv64 = rmean.double().clone(), rvar.double().clone()
    xn = x64.t().reshape(1, c, rows, 1)
    y64 = F.relu(F.batch_norm(xn, rm64, rv64, g64, b64, training=True, momentum=0.1, eps=1e-3)).reshape(c, rows).t()
    gz64 = cot.double() * (y64 > 0)                                   # the gradient the producing kernels hand over is already masked
    (y64 * cot.double()).sum().backward()
    xb = torch.zeros(rows, ldx, dtype=tdt, device="cuda")
    xb[:, cxoff:cxoff + c] = x.cuda()
    yb = torch.full((rows, ldy), 7.0, dtype=tdt, device="cuda")
    nparts = lib.din_bn_parts(rows)
    assert lib.din_bn_workspace(rows, c) == (nparts + 1) * 2 * c * 8
    sums = torch.full(((nparts + 1) * 2 * c,), float("nan"), dtype=torch.float64, device="cuda")   # the workspace needs no zeroing
    a, b, mean, rstd = (torch.empty(c, device="cuda") for _ in range(4))
    gd, bd, rmd, rvd = gamma.cuda(), beta.cuda(), rmean.cuda(), rvar.cuda()
    shift = rmd.clone() if use_shift else None
    sp = shift.data_ptr() if use_shift else None
    L.check(lib.din_bn_stats(xb.data_ptr(), dt, rows, c, ldx, cxoff, sp, sums.data_ptr(), None))
    L.check(lib.din_bn_finalize(sums.data_ptr(), nparts, rows, c, gd.data_ptr(), bd.data_ptr(), 1e-3, 0.1, rmd.data_ptr(), rvd.data_ptr(),
                                a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sp, None))
    L.check(lib.din_bn_apply(xb.data_ptr(), dt, rows, c, ldx, cxoff, a.data_ptr(), b.data_ptr(), 1, yb.data_ptr(), ldy, cyoff, None))
    torch.cuda.synchronize()
    # fp32: 1e-5 when |mean| ~ std.  With |mean| = 176 std (offset 300) the INPUT's own fp32 spacing is 176 x 6e-8 = 1e-5 of a standard
    # deviation, and a*x + b cancels it into the output: 5.7e-6 .. 7.0e-6 measured (deterministic), bound 3e-5.  bf16: output rounding only
    f32tol = 1e-5 if offset < 1 else 3e-5
    tol = f32tol if dtype == "fp32" else 6e-3
    assert rel(yb[:, cyoff:cyoff + c].float().cpu(), y64.detach()) <= tol
    assert float(yb[:, :cyoff].float().min()) == 7.0 and float(yb[:, cyoff + c:].float().min()) == 7.0
    assert rel(mean.cpu(), x.double().mean(0)) <= 1e-5 and rel(rstd.cpu(), 1.0 / (x.double().var(0, unbiased=False) + 1e-3).sqrt()) <= 1e-5
    assert rel(rmd.cpu(), rm64) <= 1e-5 and rel(rvd.cpu(), rv64) <= 1e-5
    gzb = torch.zeros(rows, ldy, dtype=tdt, device="cuda")
    gzb[:, cyoff:cyoff + c] = gz64.to(tdt).cuda()
    gz_used = gzb[:, cyoff:cyoff + c].double().cpu()                   # what the kernel actually sees (bf16-rounded)
    fwd_bits = (sums[:2 * c].clone(), mean.clone(), rstd.clone())
    sums.fill_(float("nan"))
    dy = torch.empty(rows, c, dtype=tdt, device="cuda")
    dgamma, dbeta = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    L.check(lib.din_bn_bwd_stats(gzb.data_ptr(), ldy, cyoff, xb.data_ptr(), ldx, cxoff, dt, rows, c, mean.data_ptr(), rstd.data_ptr(),
                                 sums.data_ptr(), None))
    L.check(lib.din_bn_bwd_apply(gzb.data_ptr(), ldy, cyoff, xb.data_ptr(), ldx, cxoff, dt, rows, c, gd.data_ptr(), mean.data_ptr(),
                                 rstd.data_ptr(), sums.data_ptr(), dy.data_ptr(), c, 0, dgamma.data_ptr(), dbeta.data_ptr(), None))
    torch.cuda.synchronize()
    # closed form in float64 on the gradient the kernel saw
    mu, var = x.double().mean(0), x.double().var(0, unbiased=False)
    rs = 1.0 / (var + 1e-3).sqrt()
    xh = (x.double() - mu) * rs
    s1, s2 = gz_used.sum(0), (gz_used * xh).sum(0)
    want = gamma.double() * rs * (gz_used - s1 / rows - xh * s2 / rows)
    assert rel(dy.float().cpu(), want) <= tol
    assert rel(dgamma.cpu(), s2) <= f32tol and rel(dbeta.cpu(), s1) <= 1e-5
    if dtype == "fp32":                                                # and autograd agrees with the closed form
        assert rel(dy.cpu(), x64.grad) <= f32tol and rel(dgamma.cpu(), g64.grad) <= f32tol and rel(dbeta.cpu(), b64.grad) <= 1e-5
    # determinism: the same launches again (workspace refilled with NaN) reproduce every bit, forward and backward
    bwd_bits = (sums[:2 * c].clone(), dy.clone(), dgamma.clone(), dbeta.clone())
    for _ in range(3):
        sums.fill_(float("nan"))
        rm2, rv2 = rmean.cuda(), rvar.cuda()
        L.check(lib.din_bn_stats(xb.data_ptr(), dt, rows, c, ldx, cxoff, sp, sums.data_ptr(), None))
        L.check(lib.din_bn_finalize(sums.data_ptr(), nparts, rows, c, gd.data_ptr(), bd.data_ptr(), 1e-3, 0.1, rm2.data_ptr(), rv2.data_ptr(),
                                    a.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sp, None))
        torch.cuda.synchronize()
        assert torch.equal(sums[:2 * c], fwd_bits[0]) and torch.equal(mean, fwd_bits[1]) and torch.equal(rstd, fwd_bits[2])
        assert torch.equal(rm2, rmd) and torch.equal(rv2, rvd)
        sums.fill_(float("nan"))
        L.check(lib.din_bn_bwd_stats(gzb.data_ptr(), ldy, cyoff, xb.data_ptr(), ldx, cxoff, dt, rows, c, mean.data_ptr(), rstd.data_ptr(),
                                     sums.data_ptr(), None))
        L.check(lib.din_bn_bwd_apply(gzb.data_ptr(), ldy, cyoff, xb.data_ptr(), ldx, cxoff, dt, rows, c, gd.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), sums.data_ptr(), dy.data_ptr(), c, 0, dgamma.data_ptr(), dbeta.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(sums[:2 * c], bwd_bits[0]) and torch.equal(dy, bwd_bits[1])
        assert torch.equal(dgamma, bwd_bits[2]) and torch.equal(dbeta, bwd_bits[3])


@pytest.mark.parametrize("rows,c", [(1_300_003, 192), (40_001, 2048), (9, 8)], ids=["capped_parts", "wide", "tiny"])
def test_batch_stat_bn_statistics_exact_and_reproducible(env, rows, c):
    """din_bn_stats + din_bn_finalize at the extremes of the decomposition (more rows than 1024 parts of 256: the capped case; 2048
    channels: one row-lane per workgroup; fewer rows than lanes) in bf16: the sums of bf16 values and of their exact squares are
    fp64 sums, so mean / rstd agree with a float64 reference to 1e-6 and repeat bit for bit."""
    lib, L, nhwc, ops = env
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.randn(rows, c, generator=g, device="cuda") * 0.9 + 37.0).to(torch.bfloat16)
    nparts = lib.din_bn_parts(rows)
    assert 1 <= nparts <= 1024
    gam, bet = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    outs = []
    for _ in range(3):
        ws = torch.full((lib.din_bn_workspace(rows, c) // 8,), float("nan"), dtype=torch.float64, device="cuda")
        a, b, mean, rstd = (torch.empty(c, device="cuda") for _ in range(4))
        L.check(lib.din_bn_stats(x.data_ptr(), L.DIN_BF16, rows, c, c, 0, None, ws.data_ptr(), None))
        L.check(lib.din_bn_finalize(ws.data_ptr(), nparts, rows, c, gam.data_ptr(), bet.data_ptr(), 1e-3, 0.1, None, None, a.data_ptr(),
                                    b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, None))
        torch.cuda.synchronize()
        outs.append((ws[:2 * c].clone(), mean, rstd))
    for o in outs[1:]:
        assert all(torch.equal(p, q) for p, q in zip(o, outs[0]))
    xd = x.double()
    s1, s2 = xd.sum(0), (xd * xd).sum(0)
    assert rel(outs[0][0][:c], s1) <= 1e-12 and rel(outs[0][0][c:], s2) <= 1e-12
    mu = s1 / rows
    var = (s2 / rows - mu * mu).clamp_min(0)
    assert rel(outs[0][1].double(), mu) <= 1e-6 and rel(outs[0][2].double(), 1.0 / (var + 1e-3).sqrt()) <= 1e-6


def test_din_walk_variable_actors_matches_per_clip_runs(env):
    """din_walk_{fwd,bwd} with n_per_clip (Dynamic_collective's batched form) against running the module on each clip's own
    [1, T, n_b, C] slice, as the reference does (infer_model.py:1284-1293): outputs and gradients agree to fp32 rounding on the valid
    actors (the contractions around the walk tile differently for the two batch shapes), exactly 0 / no gradient on the padding actors."""
    lib, L, nhwc, ops = env
    from din_amd.infer_module.dynamic_infer_module import Dynamic_Person_Inference
    g = torch.Generator().manual_seed(3)
    B, T, N, Cc = 4, 3, 7, 96
    counts = torch.tensor([7, 1, 4, 2], dtype=torch.int32)
    mod = Dynamic_Person_Inference(in_dim=Cc, person_mat_shape=(10, 12), kernel_size=(3, 3), dynamic_sampling=True, sampling_ratio=[1, 2],
                                   scale_factor=True, beta_factor=True).cuda()
    with torch.no_grad():
        for name, p in mod.named_parameters():
            if "p_conv" in name or "scale_conv" in name:
                p.copy_((torch.randn(p.shape, generator=g) * 0.08).cuda())
    x = torch.randn(B, T, N, Cc, generator=g)
    for b in range(B):
        x[b, :, int(counts[b]):] = 0.0
    cot = torch.randn(B, T, N, Cc, generator=g)
    xd = x.cuda().requires_grad_(True)
    out, _ = mod(ops.MaskActorsFunction.apply(xd, counts.cuda()), counts.cuda())      # as Dynamic_collective.forward does
    (out * cot.cuda()).sum().backward()
    batched = {k: p.grad.clone() for k, p in mod.named_parameters()}
    gx = xd.grad.clone()
    for p in mod.parameters():
        p.grad = None
    outs, gxs = [], []
    for b in range(B):
        nb = int(counts[b])
        xb = x[b:b + 1, :, :nb].contiguous().cuda().requires_grad_(True)
        ob, _ = mod(xb)
        (ob * cot[b:b + 1, :, :nb].cuda()).sum().backward()
        outs.append(ob.detach())
        gxs.append(xb.grad)
    for b in range(B):
        nb = int(counts[b])
        assert rel(out[b:b + 1, :, :nb], outs[b]) <= 2e-6, f"clip {b}: batched output differs from the per-clip run"
        assert float(out[b, :, nb:].detach().abs().sum()) == 0.0
        assert rel(gx[b:b + 1, :, :nb], gxs[b]) <= 2e-6
        assert float(gx[b, :, nb:].abs().sum()) == 0.0
    for k, p in mod.named_parameters():
        assert rel(batched[k], p.grad) <= 2e-5, k               # parameter gradients: sums over clips in a different order


def test_din_walk_bwd_lds_tile_and_global_scatter_agree(env, monkeypatch):
    """The backward scatters the feature gradient into dx with global atomics by default (round 2: 320 -> 118 us on 32 clips); the LDS
    dP tile (DIN_WALK_BWD_GLOBAL=0) must give the same gradients up to the summation order."""
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(12)
    B, T, N, Cc, r, k2 = 4, 3, 12, 128, 1, 9
    x = torch.randn(B, T, N, Cc, generator=g)
    w = torch.randn(3 * k2, Cc, 3, 3, generator=g) * 0.03
    b = torch.randn(3 * k2, generator=g) * 0.4
    cot = torch.randn(B, T, N, Cc, generator=g).cuda()
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DIN_WALK_BWD_GLOBAL", mode)
        xd, wd, bd = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
        pred = ops.GridConvFunction.apply(xd, wd, bd, r)
        z = ops.DynamicWalkFunction.apply(xd, pred, 3, 3, r, True, False)[0]
        (z * cot).sum().backward()
        grads[mode] = (z.detach(), xd.grad, wd.grad, bd.grad)
    assert torch.equal(grads["1"][0], grads["0"][0])
    for a_, b_ in zip(grads["1"][1:], grads["0"][1:]):
        assert rel(a_, b_) <= 2e-6


def test_din_walk_bwd_large_kernel_uses_global_scatter(env):
    """at T=10, N=12 a 5x5 ST kernel with ratio 2 needs two 92-KiB padded tiles in the backward (> 160 KiB of LDS): the
    backward then scatters the feature gradient with global atomics instead of failing.  Checked against the oracle's autograd."""
    lib, L, nhwc, ops = env
    from oracle import din_oracle as O
    g = torch.Generator().manual_seed(11)
    B, T, N, Cc, k, r = 2, 10, 12, 64, (5, 5), 2
    k2 = 25
    x = torch.randn(B, T, N, Cc, generator=g)
    pw, pb = torch.randn(2 * k2, Cc, 5, 5, generator=g) * 0.02, torch.randn(2 * k2, generator=g) * 0.3
    sw, sb = torch.randn(k2, Cc, 5, 5, generator=g) * 0.02, torch.randn(k2, generator=g) * 0.1
    cot = torch.randn(B, T, N, Cc, generator=g)
    ins = [t.clone().requires_grad_(True) for t in (x, pw, pb, sw, sb)]
    z_ref, _ = O.din_ratio_forward(ins[0], ins[1], ins[2], ins[3], ins[4], k, r)[:2]
    (z_ref * cot).sum().backward()
    xd = x.cuda().requires_grad_(True)
    wd = torch.cat([pw, sw], 0).cuda().requires_grad_(True)
    bd = torch.cat([pb, sb], 0).cuda().requires_grad_(True)
    pred = ops.GridConvFunction.apply(xd, wd, bd, r)
    z, a, idx, mad = ops.DynamicWalkFunction.apply(xd, pred, 5, 5, r, True, False)
    (z * cot.cuda()).sum().backward()
    assert rel(z, z_ref) <= 1e-4
    assert rel(xd.grad, ins[0].grad) <= 2e-4
    assert rel(wd.grad, torch.cat([ins[1].grad, ins[3].grad], 0)) <= 2e-4
    assert rel(bd.grad, torch.cat([ins[2].grad, ins[4].grad], 0)) <= 2e-4


@pytest.mark.parametrize("cin,couts", [(192, (64, 112, 32)), (256, (64, 112, 64)), (288, (64, 112, 64)), (288, (112, 16, 96, 32))],
                         ids=["5b_192", "5c_256", "5d_288", "four_sources"])
def test_conv1x1_wgrad_multi_source(env, monkeypatch, cin, couts):
    """din_conv1x1_wgrad_multi (conv_wgrad_1x1_multi_kernel: the weight gradients of every 1x1 conv reading one tensor view in one launch,
    dW stationary in registers) against fp32 einsums on the bf16-rounded operands: dW with the BatchNorm scale folded, the <W, dW> dot, the
    bias gradient; gradient operands at channel offsets of wider tensors, a pixel count that is not a multiple of the 64-pixel stage."""
    lib, L, nhwc, ops = env
    monkeypatch.setenv("DIN_WGRAD_1X1_MULTI", "2")
    g = torch.Generator().manual_seed(cin + len(couts))
    nb, h, w = 3, 37, 41                                              # 4551 pixels = 71 stages + 7 pixels
    M, ldi, cioff = nb * h * w, cin + 16, 8
    xb = torch.randn(M, ldi, generator=g).bfloat16()
    x = xb[:, cioff:cioff + cin].float()
    srcs = (L.ConvWSrc * len(couts))()
    keep, refs = [], []
    for j, co in enumerate(couts):
        ld, coff = co + 24, 16 if j % 2 == 0 else 0
        gb = torch.randn(M, ld, generator=g).bfloat16()
        gsl = gb[:, coff:coff + co].float()
        wj = torch.randn(co, cin, 1, 1, generator=g) * 0.05
        scale = torch.rand(co, generator=g) + 0.5
        dw_raw = gsl.t() @ x                                          # [co, cin]
        refs.append((dw_raw * scale[:, None], (dw_raw * wj.reshape(co, cin)).sum(1), gsl.sum(0)))
        dev = dict(g=gb.cuda(), w=wj.cuda(), scale=scale.cuda(), dw=torch.full((co, cin, 1, 1), 7.0, device="cuda"),
                   db=torch.zeros(co, device="cuda"), wdot=torch.zeros(co, device="cuda"))
        keep.append(dev)
        srcs[j].dout, srcs[j].dw, srcs[j].dbias = dev["g"].data_ptr(), dev["dw"].data_ptr(), dev["db"].data_ptr() if j != 1 else None
        srcs[j].scale, srcs[j].w, srcs[j].wdot = dev["scale"].data_ptr(), dev["w"].data_ptr(), dev["wdot"].data_ptr()
        srcs[j].cout, srcs[j].ldo, srcs[j].cooff = co, ld, coff
    wsb = lib.din_conv1x1_wgrad_multi_workspace(len(couts), srcs, L.DIN_BF16, M, cin)
    assert wsb > 0
    assert lib.din_conv1x1_wgrad_multi_workspace(len(couts), srcs, L.DIN_F32, M, cin) == 0          # bf16 only
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    xd = xb.cuda()
    L.check(lib.din_conv1x1_wgrad_multi(len(couts), srcs, L.DIN_BF16, M, cin, ldi, cioff, xd.data_ptr(), 2, ws.data_ptr(), wsb, None))
    torch.cuda.synchronize()
    for j, (dev, (dw_ref, wdot_ref, db_ref)) in enumerate(zip(keep, refs)):
        assert rel(dev["dw"].reshape(dw_ref.shape), dw_ref) <= 2e-5, j
        assert rel(dev["wdot"], wdot_ref) <= 2e-5, j
        if j != 1:
            assert rel(dev["db"], db_ref) <= 2e-5, j
        else:
            assert float(dev["db"].abs().max()) == 0.0


@pytest.mark.parametrize("case", ["inception_c_block", "mixed_maps", "two_tiles", "twelve_layers"])
def test_conv_wgrad_group_matches_per_layer_launches(env, case):
    """din_conv_wgrad_group (several layers' weight gradients in ONE launch of the pipelined kernel, conv_wgrad.h: WgradGroupK) against
    din_conv_wgrad layer by layer and against fp32 F.conv2d autograd on the bf16-rounded operands: dW with the BatchNorm scale folded, the
    <W, dW> dot, the bias gradient; 7-tap and 1x1 layers of one map size (an InceptionC block's shapes, reference backbone/backbone.py:67-74
    through torchvision), layers of DIFFERENT map sizes in one group, a bank of two filter tiles, gradient / input views at channel offsets."""
    lib, L, nhwc, ops = env
    g = torch.Generator().manual_seed(len(case))
    if case == "inception_c_block":      # (cin, cout, kh, kw, ph, pw, nb, h, w)
        layers = [(160, 160, 1, 7, 0, 3, 2, 19, 37), (160, 192, 7, 1, 3, 0, 2, 19, 37), (160, 160, 7, 1, 3, 0, 2, 19, 37),
                  (768, 192, 1, 1, 0, 0, 2, 19, 37), (160, 192, 1, 7, 0, 3, 2, 19, 37), (192, 192, 7, 1, 3, 0, 2, 19, 37)]
    elif case == "twelve_layers":         # more items than one hand of the first[] scan: two blocks' worth in one launch
        layers = [(160, 160, 1, 7, 0, 3, 1, 19, 37), (160, 192, 7, 1, 3, 0, 1, 19, 37), (768, 192, 1, 1, 0, 0, 1, 19, 37)] * 4
    elif case == "mixed_maps":
        layers = [(192, 192, 1, 7, 0, 3, 1, 21, 40), (80, 192, 3, 3, 0, 0, 1, 45, 70), (192, 192, 7, 1, 3, 0, 3, 17, 33)]
    else:
        layers = [(288, 384, 3, 3, 1, 1, 1, 23, 35), (256, 384, 1, 1, 0, 0, 2, 23, 35)]
    items = (L.ConvWgradItem * len(layers))()
    keep, refs = [], []
    for j, (cin, cout, kh, kw, ph, pw, nb, h, w) in enumerate(layers):
        oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
        ldi, cioff, ldo, cooff = cin + 16, 8 * (j % 2), cout + 24, 16 if j % 2 else 0
        xb = torch.randn(nb, h, w, ldi, generator=g).bfloat16()
        gb = torch.randn(nb, oh, ow, ldo, generator=g).bfloat16()
        wj = torch.randn(cout, cin, kh, kw, generator=g) * 0.05
        scale = torch.rand(cout, generator=g) + 0.5
        xr = xb[..., cioff:cioff + cin].float().permute(0, 3, 1, 2)
        gr = gb[..., cooff:cooff + cout].float().permute(0, 3, 1, 2)
        dw_raw = torch.nn.grad.conv2d_weight(xr, wj.shape, gr, padding=(ph, pw))
        refs.append((dw_raw * scale[:, None, None, None], (dw_raw * wj).sum((1, 2, 3)), gr.sum((0, 2, 3))))
        dev = dict(x=xb.cuda(), g=gb.cuda(), w=wj.cuda(), scale=scale.cuda())
        for tag in ("grp", "one"):
            dev["dw_" + tag] = torch.full(wj.shape, 7.0, device="cuda")
            dev["db_" + tag], dev["wdot_" + tag] = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")
        keep.append(dev)
        d = items[j].desc
        d.nb, d.h, d.w, d.cin, d.oh, d.ow, d.cout = nb, h, w, cin, oh, ow, cout
        d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = kh, kw, 1, 1, ph, pw, 1, 1
        d.ldi, d.cioff, d.ldo, d.cooff, d.dtype, d.in_u8 = ldi, cioff, ldo, cooff, L.DIN_BF16, 0
        it = items[j]
        it.in_, it.dout, it.dw = dev["x"].data_ptr(), dev["g"].data_ptr(), dev["dw_grp"].data_ptr()
        it.dbias = dev["db_grp"].data_ptr() if j != 1 else None
        it.scale, it.w, it.wdot, it.accumulate = dev["scale"].data_ptr(), dev["w"].data_ptr(), dev["wdot_grp"].data_ptr(), 2
    keys = [lib.din_conv_wgrad_group_key(C.byref(items[j].desc)) for j in range(len(layers))]
    assert all(k > 0 for k in keys) and len(set(keys)) == 1, keys
    f32 = L.ConvDesc.from_buffer_copy(items[0].desc)
    f32.dtype = L.DIN_F32
    assert lib.din_conv_wgrad_group_key(C.byref(f32)) == 0                      # bf16 pipelined kernel only
    wsb = lib.din_conv_wgrad_group_workspace(len(layers), items)
    assert wsb > 0
    single = sum(lib.din_conv_workspace_bytes(C.byref(items[j].desc), 2) for j in range(len(layers)))
    print(f"{case}: group workspace {wsb / 1e6:.1f} MB for {len(layers)} layers, the per-layer launches' partials {single / 1e6:.1f} MB")
    assert wsb < single                                                          # the point of the exercise: fewer slice partials
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    L.check(lib.din_conv_wgrad_group(len(layers), items, ws.data_ptr(), wsb, None))
    for j, dev in enumerate(keep):
        d = items[j].desc
        nb1 = lib.din_conv_workspace_bytes(C.byref(d), 2)
        w1 = torch.empty(nb1, dtype=torch.uint8, device="cuda")
        L.check(lib.din_conv_wgrad(C.byref(d), dev["x"].data_ptr(), dev["g"].data_ptr(), dev["dw_one"].data_ptr(),
                                   dev["db_one"].data_ptr() if j != 1 else None, dev["scale"].data_ptr(), dev["w"].data_ptr(),
                                   dev["wdot_one"].data_ptr(), 2, w1.data_ptr(), nb1, None))
    torch.cuda.synchronize()
    for j, (dev, (dw_ref, wdot_ref, db_ref)) in enumerate(zip(keep, refs)):
        assert rel(dev["dw_grp"], dev["dw_one"]) <= 2e-6, j                      # same products, another slicing of the pixel sum
        assert rel(dev["wdot_grp"], dev["wdot_one"]) <= 2e-5, j
        assert rel(dev["dw_grp"], dw_ref) <= 2e-5 and rel(dev["wdot_grp"], wdot_ref) <= 2e-5, j
        if j != 1:
            assert rel(dev["db_grp"], db_ref) <= 2e-5 and rel(dev["db_grp"], dev["db_one"]) <= 2e-6, j
        else:
            assert float(dev["db_grp"].abs().max()) == 0.0
    # a list the library does not take as one launch (a lone item) still produces the gradient: the per-layer path
    keep[0]["dw_grp"].fill_(3.0)
    keep[0]["wdot_grp"].zero_()
    if items[0].dbias:
        keep[0]["db_grp"].zero_()
    assert lib.din_conv_wgrad_group_workspace(1, items) == 0
    nb1 = lib.din_conv_workspace_bytes(C.byref(items[0].desc), 2)
    w1 = torch.empty(nb1, dtype=torch.uint8, device="cuda")
    L.check(lib.din_conv_wgrad_group(1, items, w1.data_ptr(), nb1, None))
    torch.cuda.synchronize()
    assert rel(keep[0]["dw_grp"], keep[0]["dw_one"]) == 0.0


@pytest.mark.parametrize("shape", [(144, 26400, 1024), (37, 776, 200)], ids=["fc_emb_1_cfg1_4clips", "ragged"])
def test_lowp_linear_and_direct_wgrad_epilogue(env, shape, monkeypatch):
    """ops.linear(lowp=True): the embedding layer of the benchmarked mode (infer_model.py:183 fc_emb_1, K*K*D = 26400 -> 1024; bf16 operands, fp32
    accumulation) against torch on the bf16-rounded operands -- y, dX, dW, dbias -- and the single-slice weight gradient written straight from the
    pipelined kernel's accumulators (WgradK::direct) against the partial-buffer + reduce form of the same launch: bit-identical."""
    lib, L, nhwc, ops = env
    rows, cin, cout = shape
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, cin, generator=g)).bfloat16().float()
    w = (torch.randn(cout, cin, generator=g) * cin ** -0.5).bfloat16().float()
    b = torch.randn(cout, generator=g) * 0.1
    cot = torch.randn(rows, cout, generator=g).bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    yr.backward(cot)
    outs = []
    for direct in ("1", "0"):
        monkeypatch.setenv("DIN_WGRAD_DIRECT", direct)
        xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        y = ops.linear(xd, wd, bd, lowp=True)
        y.backward(cot.cuda())
        torch.cuda.synchronize()
        outs.append((y.detach(), xd.grad, wd.grad, bd.grad))
    y, dx, dw, db = outs[0]
    assert rel(y, yr) <= 1e-2                       # bf16 output rounding
    assert rel(dx, xr.grad) <= 1e-2
    assert rel(dw, wr.grad) <= 1e-4                 # fp32 result of bf16 products
    assert rel(db, br.grad) <= 1e-4
    assert torch.equal(outs[0][2], outs[1][2]), "direct epilogue and reduce launch disagree"
    assert torch.equal(outs[0][3], outs[1][3])
