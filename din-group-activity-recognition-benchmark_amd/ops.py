"""torch.autograd.Function wrappers over the C ABI for everything after the backbone (rows R, E, L, D, H).

Each Function's forward/backward is a handful of C calls; PyTorch provides device memory, the current HIP stream and
the autograd tape only.  All tensors here are fp32 (the tail of the path is always fp32, include/din_hip.h).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L
from .nhwc import _ptr, _stream, din_dtype, require_gpu, workspace


# Device-side per-step part of every dropout seed (int64 [1]) or None.  A training step captured in a HIP graph (din_amd.graph_step) bakes
# the host-computed seeds into the launches; the kernels add *SEED_OFFSET, which the captured step advances once per replay.
SEED_OFFSET: Optional[torch.Tensor] = None


# ------------------------------------------------------------------------------------------------
# Row P (API-parity form)
# ------------------------------------------------------------------------------------------------
def prep_images_f32(images: torch.Tensor) -> torch.Tensor:
    lib = L.load()
    x = images.float().contiguous()
    require_gpu(x)
    out = torch.empty_like(x)
    L.check(lib.din_prep_images_f32(_ptr(x), _ptr(out), x.numel(), _stream()), "prep_images")
    return out


def boxes_frame_index(bt: int, n: int, device) -> torch.Tensor:
    """infer_model.py:155-157 -- int32 [bt*n], box i*n+j crops from frame i."""
    lib = L.load()
    out = torch.empty(bt * n, dtype=torch.int32, device=device)
    require_gpu(out)
    L.check(lib.din_boxes_frame_index(_ptr(out), bt, n, _stream()), "boxes_frame_index")
    return out


# ------------------------------------------------------------------------------------------------
# Row R: RoIAlign
# ------------------------------------------------------------------------------------------------
def _roi_forward(lib, fms, chans, grid, boxes, box_ind, k, want_index):
    """crops fp32 [m, sum(chans), k, k]: one launch per stored map, each writes its channel range (the torch.cat of infer_model.py:172)"""
    m, ctot = boxes.shape[0], sum(chans)
    out = torch.empty((m, ctot, k, k), dtype=torch.float32, device=boxes.device)
    idx = torch.empty((m, k, k, 6), dtype=torch.int32, device=boxes.device) if want_index else None
    coff = 0
    for fm, c in zip(fms, chans):
        nb, hf, wf, ld = fm.shape
        gh, gw = grid if grid is not None else (hf, wf)
        L.check(lib.din_roi_align_fwd(_ptr(fm), din_dtype(fm), nb, hf, wf, c, ld, gh, gw, _ptr(boxes), _ptr(box_ind), m, k,
                                      _ptr(out), ctot, coff, _ptr(idx) if (hf, wf) == (gh, gw) else None, _stream()), "roi_align_fwd")
        coff += c
    return out, idx


def _roi_backward(lib, gout, fms, chans, masked, grid, boxes, box_ind, k):
    """gather form: every stored map's gradient tensor is written once, in the map's storage type, already multiplied by the ReLU mask
    of the map (no fp32 scatter buffer, no cast pass); with a larger box grid this is also the backward of the bilinear resize"""
    st = _stream()
    m, ctot = boxes.shape[0], sum(chans)
    gout = gout.contiguous().float()
    src, transposed = gout, 0
    if ctot % 4 == 0 and all(c % 4 == 0 for c in chans):
        src, transposed = torch.empty_like(gout), 1                  # channel-contiguous copy of the crop gradient, shared by the maps
        L.check(lib.din_roi_crop_grad_transpose(_ptr(gout), m, ctot, k, _ptr(src), st), "roi_crop_grad_transpose")
    grads, coff = [], 0
    for fm, c, mk in zip(fms, chans, masked):
        nb, hf, wf, ld = fm.shape
        gh, gw = grid if grid is not None else (hf, wf)
        gfm = torch.zeros_like(fm) if ld != c else torch.empty_like(fm)
        L.check(lib.din_roi_align_bwd_nhwc(_ptr(src), ctot, coff, transposed, nb, hf, wf, c, gh, gw, _ptr(boxes), _ptr(box_ind), m, k,
                                           _ptr(fm) if mk else None, din_dtype(fm), ld, _ptr(gfm), ld, st), "roi_align_bwd_nhwc")
        grads.append(gfm)
        coff += c
    return grads


class RoIAlignFunction(torch.autograd.Function):
    """fm: NHWC buffer [nb,hf,wf,ld] (fp32|bf16) -> crops fp32 [m, c, k, k] (reference flatten order)."""

    @staticmethod
    def forward(ctx, fm: torch.Tensor, boxes: torch.Tensor, box_ind: torch.Tensor, k: int, c: int, relu_masked: bool,
                want_index: bool = False):
        lib = L.load()
        boxes = boxes.detach().float().contiguous()
        box_ind = box_ind.detach().to(torch.int32).contiguous()
        require_gpu(fm, boxes, box_ind)
        out, idx = _roi_forward(lib, [fm], [c], None, boxes, box_ind, k, want_index)
        ctx.save_for_backward(fm, boxes, box_ind)
        ctx.k, ctx.c, ctx.relu_masked = k, c, relu_masked
        if want_index:
            ctx.mark_non_differentiable(idx)
            return out, idx
        return out

    @staticmethod
    def backward(ctx, gout, *_):
        fm, boxes, box_ind = ctx.saved_tensors
        (gfm,) = _roi_backward(L.load(), gout, [fm], [ctx.c], [ctx.relu_masked], None, boxes, box_ind, ctx.k)
        return gfm, None, None, None, None, None, None


class RoIAlignMultiScaleFunction(torch.autograd.Function):
    """RoIAlign over torch.cat([resize(fm_i, grid) for fm_i in fms], channel) without building it (infer_model.py:165-180): boxes are
    in `grid` = (OH, OW) pixels, every stored map fm_i [nb,h_i,w_i,ld_i] (h_i <= OH, w_i <= OW) is sampled through its virtual
    align_corners resize, and the backward writes each map's gradient directly.  chans / masked: per-map channel count and "apply the
    map's ReLU mask" flag.  Returns crops fp32 [m, sum(chans), k, k]."""

    @staticmethod
    def forward(ctx, boxes: torch.Tensor, box_ind: torch.Tensor, k: int, grid, chans, masked, *fms):
        lib = L.load()
        boxes = boxes.detach().float().contiguous()
        box_ind = box_ind.detach().to(torch.int32).contiguous()
        require_gpu(boxes, box_ind, *fms)
        out, _ = _roi_forward(lib, fms, list(chans), tuple(grid), boxes, box_ind, k, False)
        ctx.save_for_backward(boxes, box_ind, *fms)
        ctx.k, ctx.grid, ctx.chans, ctx.masked = k, tuple(grid), list(chans), list(masked)
        return out

    @staticmethod
    def backward(ctx, gout):
        boxes, box_ind, *fms = ctx.saved_tensors
        grads = _roi_backward(L.load(), gout, fms, ctx.chans, ctx.masked, ctx.grid, boxes, box_ind, ctx.k)
        return (None, None, None, None, None, None, *grads)


# ------------------------------------------------------------------------------------------------
# dense contractions on the MFMA conv kernel (fp32): Linear and the (T x N)-grid p_conv/scale_conv
# ------------------------------------------------------------------------------------------------
def _desc(nb, h, w, cin, cout, kh, kw, ph, pw, dil, ldi, ldo) -> L.ConvDesc:
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin = nb, h, w, cin
    d.oh, d.ow, d.cout = h, w, cout          # "same" geometry (stride 1, symmetric dilation-scaled padding)
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw = kh, kw, 1, 1, ph, pw, dil, dil
    d.ldi, d.cioff, d.ldo, d.cooff = ldi, 0, ldo, 0
    d.dtype = L.DIN_F32
    return d


class GridConvFunction(torch.autograd.Function):
    """x [nb,h,w,cin] fp32 NHWC, weight [cout,cin,kh,kw], bias [cout]|None -> y [nb,h,w,ldo] (ldo = cout padded to 4,
    padding channels are zero).  Stride 1, padding = (k-1)//2*dil: nn.Linear (1x1), point_conv (infer_model.py:190) and
    the fused p_conv/scale_conv (dynamic_infer_module.py:191,195)."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil: int, lowp: bool = False):
        """lowp: bf16 operands / fp32 MFMA accumulation (throughput mode, used for the 26400-wide embedding GEMM when the backbone
        already runs in bf16); x, y and all gradients stay fp32 tensors at the interface."""
        lib = L.load()
        x = x.contiguous()
        weight = weight.contiguous()
        require_gpu(x, weight, bias)
        nb, h, w, cin = x.shape
        cout, _, kh, kw = weight.shape
        lowp = bool(lowp) and cin % 8 == 0 and cout % 8 == 0
        tdt = torch.bfloat16 if lowp else torch.float32
        ldo = cout if lowp else (cout + 3) // 4 * 4
        d = _desc(nb, h, w, cin, cout, kh, kw, (kh - 1) // 2 * dil, (kw - 1) // 2 * dil, dil, cin, ldo)
        if lowp:
            d.dtype = L.DIN_BF16
            x = x.to(torch.bfloat16)
        st = _stream()
        y = (torch.zeros if ldo != cout else torch.empty)((nb, h, w, ldo), dtype=tdt, device=x.device)
        wpk = torch.empty(lib.din_conv_packed_elems(C.byref(d), 0), dtype=tdt, device=x.device)
        L.check(lib.din_conv_pack_weights(C.byref(d), _ptr(weight), None, _ptr(wpk), 0, st), "conv_pack")
        ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 0), x.device)
        L.check(lib.din_conv_fwd(C.byref(d), _ptr(x), _ptr(wpk), _ptr(bias), _ptr(y), L.CONV_BIAS if bias is not None else 0,
                                 _ptr(ws), wsb, st), "grid_conv_fwd")
        ctx.save_for_backward(x, weight)
        ctx.d, ctx.has_bias, ctx.lowp = d, bias is not None, lowp
        return y.float() if lowp else y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        x, weight = ctx.saved_tensors
        d = ctx.d
        tdt = torch.bfloat16 if ctx.lowp else torch.float32
        gy = gy.contiguous().to(tdt)
        st = _stream()
        dx = dw = db = None
        dw_installed = False
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            # data parallelism: the weight gradient goes straight into the parameter's slot of its all-reduce bucket (parallel.GradBuckets through
            # nhwc.GRAD_BUFFER, as for the backbone's conv weights) -- the embedding layer's 108 MB gradient is 93 % of the model's bytes, copying it
            # into the bucket cost a 108 MB device copy per step, and reporting it (GRAD_HOOK) starts its all-reduce under the whole backbone backward
            from . import nhwc as _nhwc
            buf = _nhwc.GRAD_BUFFER(weight) if _nhwc.GRAD_BUFFER is not None else None
            dw = buf if buf is not None else torch.empty_like(weight)
            db = torch.empty(weight.shape[0], dtype=torch.float32, device=x.device) if ctx.has_bias else None
            ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 2), x.device, "wgrad")   # not shared with split-K launches
            L.check(lib.din_conv_wgrad(C.byref(d), _ptr(x), _ptr(gy), _ptr(dw), _ptr(db), None, None, None, 0, _ptr(ws), wsb, st),
                    "grid_conv_wgrad")
            if _nhwc.GRAD_HOOK is not None:
                _nhwc.GRAD_HOOK(weight, dw)
            if buf is not None and _nhwc.GRAD_ASSIGN is not None and _nhwc.GRAD_ASSIGN(weight, dw):
                dw_installed = True                              # .grad IS the bucket slot now: autograd gets None (it would clone a shared tensor)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wpt = torch.empty(lib.din_conv_packed_elems(C.byref(d), 1), dtype=tdt, device=x.device)
            L.check(lib.din_conv_pack_weights(C.byref(d), _ptr(weight), None, _ptr(wpt), 1, st), "conv_pack_t")
            ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 1), x.device)
            L.check(lib.din_conv_dgrad(C.byref(d), _ptr(gy), _ptr(wpt), _ptr(dx), None, 0, 0, 0, _ptr(ws), wsb, st), "grid_conv_dgrad")
            if ctx.lowp:
                dx = dx.float()
        return dx, (None if dw_installed else dw), db, None, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], lowp: bool = False) -> torch.Tensor:
    """y = x @ weight.T + bias over the last dim, on the MFMA contraction kernel (cout must be a multiple of 4)."""
    shp = x.shape
    rows = x.numel() // shp[-1]
    y = GridConvFunction.apply(x.reshape(1, 1, rows, shp[-1]), weight.reshape(weight.shape[0], weight.shape[1], 1, 1), bias, 1, lowp)
    cout = weight.shape[0]
    if y.shape[-1] != cout:
        y = y[..., :cout]
    return y.reshape(*shp[:-1], cout)


# ------------------------------------------------------------------------------------------------
# LayerNorm (+residual, +ReLU, +dropout)
# ------------------------------------------------------------------------------------------------
class LayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, n_norm_dims: int, relu: bool, drop_p: float, seed: int):
        lib = L.load()
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        gamma, beta = gamma.contiguous(), beta.contiguous()
        require_gpu(x, res, gamma, beta)
        length = gamma.numel()
        assert tuple(x.shape[x.dim() - n_norm_dims:]) == tuple(gamma.shape), (x.shape, gamma.shape)
        rows = x.numel() // length
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        L.check(lib.din_layernorm_fwd(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), 1e-5, _ptr(y), _ptr(stats), rows, length,
                                      int(relu), float(drop_p), int(seed), _ptr(SEED_OFFSET), _stream()), "layernorm_fwd")
        ctx.seed_offset = SEED_OFFSET
        ctx.save_for_backward(x, res if res is not None else x.new_empty(0), gamma, y, stats)
        ctx.has_res, ctx.relu, ctx.drop_p, ctx.seed, ctx.rows, ctx.length = res is not None, relu, drop_p, seed, rows, length
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        x, res, gamma, y, stats = ctx.saved_tensors
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.zeros_like(gamma)
        dbeta = torch.zeros_like(gamma)
        L.check(lib.din_layernorm_bwd(_ptr(gy), _ptr(x), _ptr(res) if ctx.has_res else None, _ptr(gamma), _ptr(y), _ptr(stats),
                                      _ptr(dx), _ptr(dgamma), _ptr(dbeta), ctx.rows, ctx.length, int(ctx.relu), float(ctx.drop_p),
                                      int(ctx.seed), _ptr(ctx.seed_offset), _stream()), "layernorm_bwd")
        return dx, (dx if ctx.has_res else None), dgamma, dbeta, None, None, None, None


def mask_seed(base: int, step: int) -> int:
    """Seed of one dropout mask: (stream id `base`, this process' rank, call counter).  Every rank draws different masks (the reference's
    DataParallel replicas share one host RNG stream, so their masks differ too), and a resumed run continues the sequence when the
    caller restores its counter."""
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    return mask_seed_of(base, rank, step)


RANK_SEED_STRIDE = 0x9E3779B97F4A7C15


def mask_seed_of(base: int, rank: int, step: int) -> int:
    return (int(base) * 1000003 + int(rank) * RANK_SEED_STRIDE + int(step) * 7919) & 0x7FFFFFFFFFFFFFFF


def layer_norm(x, gamma, beta, res=None, relu=False, drop_p=0.0, seed=0):
    return LayerNormFunction.apply(x, res, gamma, beta, gamma.dim(), relu, drop_p, seed)


# ------------------------------------------------------------------------------------------------
# Rows D2-D4: Dynamic Relation + Dynamic Walk
# ------------------------------------------------------------------------------------------------
class DynamicWalkFunction(torch.autograd.Function):
    """(x [b,t,n,c], pred [b,t,n,cp]) -> z [b,t,n,c]   (+ non-differentiable a, idx, optional mad)."""

    @staticmethod
    def forward(ctx, x, pred, kh: int, kw: int, ratio: int, scale_factor: bool, want_mad: bool, n_per_clip=None, plain: bool = False,
                clamp=None):
        """plain: lattice gather without walk (plain_infer_ratio / relation half of parallel_infer).  clamp: (iy_max, ix_max, py_max,
        px_max) clamp maxima of parallel_infer's walk half (person_mat_shape based, dynamic_infer_module.py:307-317)."""
        lib = L.load()
        x, pred = x.contiguous(), pred.contiguous()
        require_gpu(x, pred, n_per_clip)
        b, t, n, c = x.shape
        cp = pred.shape[-1]
        k2 = kh * kw
        z = torch.empty_like(x)
        a = torch.empty((b, t, n, k2), dtype=torch.float32, device=x.device)
        idx = torch.empty((b, t, n, k2, 4), dtype=torch.int32, device=x.device)
        mad = torch.empty((b, t, n, k2, c), dtype=torch.float32, device=x.device) if want_mad else None
        cl = (C.c_int32 * 4)(*[int(v) for v in clamp]) if clamp is not None else None
        L.check(lib.din_walk_fwd(_ptr(x), _ptr(pred), cp, b, t, n, c, kh, kw, ratio, int(scale_factor), int(plain), cl, _ptr(n_per_clip),
                                 _ptr(z), _ptr(a), _ptr(idx), _ptr(mad), _stream()), "din_walk_fwd")
        ctx.save_for_backward(x, pred, a)
        ctx.n_per_clip, ctx.plain, ctx.clamp = n_per_clip, bool(plain), clamp
        ctx.geom = (kh, kw, ratio, scale_factor)
        if mad is None:
            mad = x.new_empty(0)
        ctx.mark_non_differentiable(a, idx, mad)
        return z, a, idx, mad

    @staticmethod
    def backward(ctx, gz, *_):
        lib = L.load()
        x, pred, a = ctx.saved_tensors
        kh, kw, ratio, scale_factor = ctx.geom
        b, t, n, c = x.shape
        cp = pred.shape[-1]
        gz = gz.contiguous()
        dx = torch.empty_like(x)
        dpred = torch.zeros_like(pred)
        scratch = torch.empty(((c + 63) // 64) * b * t * n * 3 * kh * kw, dtype=torch.float32, device=x.device)
        cl = (C.c_int32 * 4)(*[int(v) for v in ctx.clamp]) if ctx.clamp is not None else None
        L.check(lib.din_walk_bwd(_ptr(x), _ptr(pred), cp, _ptr(a), _ptr(gz), b, t, n, c, kh, kw, ratio, int(scale_factor), int(ctx.plain), cl,
                                 _ptr(ctx.n_per_clip), _ptr(dx), _ptr(dpred), _ptr(scratch), _stream()), "din_walk_bwd")
        return dx, dpred, None, None, None, None, None, None, None, None


class MaskActorsFunction(torch.autograd.Function):
    """x [b,t,n,c] with actors >= n_per_clip[b] zeroed (Dynamic_collective: the reference slices boxes_features_all[b, :, :N])."""

    @staticmethod
    def forward(ctx, x, n_per_clip):
        lib = L.load()
        x = x.contiguous()
        require_gpu(x, n_per_clip)
        b, t, n, c = x.shape
        out = torch.empty_like(x)
        L.check(lib.din_mask_actors(_ptr(x), _ptr(n_per_clip), b, t, n, c, _ptr(out), _stream()), "mask_actors")
        ctx.n_per_clip = n_per_clip
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        g = g.contiguous()
        b, t, n, c = g.shape
        out = torch.empty_like(g)
        L.check(lib.din_mask_actors(_ptr(g), _ptr(ctx.n_per_clip), b, t, n, c, _ptr(out), _stream()), "mask_actors")
        return out, None


class AxpbyFunction(torch.autograd.Function):
    """out = alpha*x + beta*y with scalar python floats (ratio mean, module sum)."""

    @staticmethod
    def forward(ctx, x, y, alpha: float, beta: float):
        lib = L.load()
        x, y = x.contiguous(), y.contiguous()
        require_gpu(x, y)
        out = torch.empty_like(x)
        L.check(lib.din_axpby(_ptr(x), _ptr(y), _ptr(out), alpha, beta, x.numel(), _stream()), "axpby")
        ctx.ab = (alpha, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        g = g.contiguous()
        a, b = ctx.ab
        ga, gb = torch.empty_like(g), torch.empty_like(g)
        st = _stream()
        L.check(lib.din_axpby(_ptr(g), None, _ptr(ga), a, 0.0, g.numel(), st), "axpby")
        L.check(lib.din_axpby(_ptr(g), None, _ptr(gb), b, 0.0, g.numel(), st), "axpby")
        return ga, gb, None, None


# ------------------------------------------------------------------------------------------------
# context-encoding transformer of Dynamic_TCE_volleyball
# ------------------------------------------------------------------------------------------------
class ContextAttentionFunction(torch.autograd.Function):
    """q [BT,N,H*C] (box queries), kf [BT,P,H*C] (per-pixel keys = values) -> (ctx [BT,N,H*C], att [BT,H,N,P]).
    att = softmax_P(<q, kf>) per head; ctx = att @ kf (TCE_STBiP_module.py:271-277).  att is returned for inspection (att_map)."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, kf: torch.Tensor, heads: int):
        lib = L.load()
        q, kf = q.contiguous(), kf.contiguous()
        require_gpu(q, kf)
        bt, n, hc = q.shape
        p = kf.shape[1]
        assert kf.shape[0] == bt and kf.shape[2] == hc and hc % heads == 0, (q.shape, kf.shape, heads)
        c = hc // heads
        st = _stream()
        att = torch.empty((bt, heads, n, p), dtype=torch.float32, device=q.device)
        out = torch.empty_like(q)
        L.check(lib.din_ctx_scores(_ptr(q), _ptr(kf), _ptr(att), bt, n, p, heads, c, st), "ctx_scores")
        L.check(lib.din_softmax_rows(_ptr(att), bt * heads * n, p, st), "softmax_rows")
        L.check(lib.din_ctx_apply(_ptr(att), _ptr(kf), _ptr(out), bt, n, p, heads, c, st), "ctx_apply")
        ctx.save_for_backward(q, kf, att)
        ctx.dims = (bt, n, p, heads, c)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, gout, _gatt):
        lib = L.load()
        q, kf, att = ctx.saved_tensors
        bt, n, p, heads, c = ctx.dims
        gout = gout.contiguous()
        st = _stream()
        ds = torch.empty_like(att)
        L.check(lib.din_ctx_scores(_ptr(gout), _ptr(kf), _ptr(ds), bt, n, p, heads, c, st), "ctx_scores(dA)")
        L.check(lib.din_softmax_rows_bwd(_ptr(att), _ptr(ds), bt * heads * n, p, st), "softmax_rows_bwd")
        dq, dkf = torch.empty_like(q), torch.empty_like(kf)
        L.check(lib.din_ctx_apply(_ptr(ds), _ptr(kf), _ptr(dq), bt, n, p, heads, c, st), "ctx_apply(dq)")
        L.check(lib.din_ctx_keys_grad(_ptr(att), _ptr(ds), _ptr(gout), _ptr(q), _ptr(dkf), bt, n, p, heads, c, st), "ctx_keys_grad")
        return dq, dkf, None


class AddPositionFunction(torch.autograd.Function):
    """x: backbone output NHWC [BT,OH,OW,C] (fp32 | bf16), pos fp32 [OH,OW,C] -> fp32 x + pos (positional_encoding.py:91).
    relu_masked: the gradient handed back to the backbone graph is multiplied by (x > 0), as the graph expects for ReLU outputs."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, pos: torch.Tensor, relu_masked: bool):
        lib = L.load()
        require_gpu(x, pos)
        assert x.is_contiguous() and tuple(x.shape[1:]) == tuple(pos.shape), (x.shape, pos.shape)
        pos = pos.contiguous().float()
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        L.check(lib.din_add_position(_ptr(x), din_dtype(x), _ptr(pos), _ptr(y), x.shape[0], pos.numel(), _stream()), "add_position")
        ctx.save_for_backward(x)
        ctx.relu_masked = relu_masked
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        (x,) = ctx.saved_tensors
        gy = gy.contiguous().float()
        gx = torch.empty_like(x)
        L.check(lib.din_add_position_bwd(_ptr(gy), _ptr(x), din_dtype(x), _ptr(gx), x.numel(), int(ctx.relu_masked), _stream()),
                "add_position_bwd")
        return gx, None, None


class ActDropoutFunction(torch.autograd.Function):
    """y = dropout(relu(x)) / dropout(x): the mask is a counter-based hash of (seed, element index), regenerated in the backward"""

    @staticmethod
    def forward(ctx, x: torch.Tensor, relu: bool, drop_p: float, seed: int):
        lib = L.load()
        x = x.contiguous()
        require_gpu(x)
        y = torch.empty_like(x)
        L.check(lib.din_act_dropout_fwd(_ptr(x), _ptr(y), x.numel(), int(relu), float(drop_p), int(seed), _ptr(SEED_OFFSET), _stream()),
                "act_dropout_fwd")
        ctx.save_for_backward(x)
        ctx.args = (relu, drop_p, seed)
        ctx.seed_offset = SEED_OFFSET
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        (x,) = ctx.saved_tensors
        relu, drop_p, seed = ctx.args
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        L.check(lib.din_act_dropout_bwd(_ptr(gy), _ptr(x), _ptr(gx), x.numel(), int(relu), float(drop_p), int(seed), _ptr(ctx.seed_offset),
                                        _stream()), "act_dropout_bwd")
        return gx, None, None, None


class ScaleByParamFunction(torch.autograd.Function):
    """out = x * scalar[idx] with the scalar read on the device (no host sync): beta-weighted ratio sum (:144-145)."""

    @staticmethod
    def forward(ctx, x, scalar, idx: int):
        lib = L.load()
        x, scalar = x.contiguous(), scalar.contiguous()
        require_gpu(x, scalar)
        out = torch.empty_like(x)
        L.check(lib.din_scale_by_param(_ptr(x), _ptr(scalar), idx, _ptr(out), 0, x.numel(), _stream()), "scale_by_param")
        ctx.save_for_backward(x, scalar)
        ctx.idx = idx
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        x, scalar = ctx.saved_tensors
        g = g.contiguous()
        st = _stream()
        dx = torch.empty_like(x)
        L.check(lib.din_scale_by_param(_ptr(g), _ptr(scalar), ctx.idx, _ptr(dx), 0, g.numel(), st), "scale_by_param")
        ds = torch.zeros_like(scalar)
        L.check(lib.din_dot_accum(_ptr(g), _ptr(x), _ptr(ds), ctx.idx, g.numel(), st), "dot_accum")
        return dx, ds, None


# ------------------------------------------------------------------------------------------------
# Row H: head
# ------------------------------------------------------------------------------------------------
class HeadFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, weight, bias, n_per_clip):
        lib = L.load()
        s, weight, bias = s.contiguous(), weight.contiguous(), bias.contiguous()
        require_gpu(s, weight, bias, n_per_clip)
        b, t, n, c = s.shape
        a = weight.shape[0]
        buf = torch.empty(b * a + b * t * a, dtype=torch.float32, device=s.device)
        argmax = torch.empty((b, t, c), dtype=torch.int32, device=s.device)
        L.check(lib.din_head_fwd(_ptr(s), _ptr(weight), _ptr(bias), _ptr(n_per_clip), b, t, n, c, a, _ptr(buf), _ptr(argmax), _stream()),
                "head_fwd")
        ctx.save_for_backward(s, weight, argmax)
        return buf[: b * a].reshape(b, a).clone()

    @staticmethod
    def backward(ctx, gscores):
        lib = L.load()
        s, weight, argmax = ctx.saved_tensors
        b, t, n, c = s.shape
        a = weight.shape[0]
        gscores = gscores.contiguous()
        ds = torch.empty_like(s)
        dw = torch.zeros_like(weight)
        db = torch.zeros(a, dtype=torch.float32, device=s.device)
        L.check(lib.din_head_bwd(_ptr(gscores), _ptr(s), _ptr(weight), _ptr(argmax), b, t, n, c, a, _ptr(ds), _ptr(dw), _ptr(db), _stream()),
                "head_bwd")
        return ds, dw, db, None


# ------------------------------------------------------------------------------------------------
# layout views for API parity (NOT on the training path)
# ------------------------------------------------------------------------------------------------
class NHWCToNCHWFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buf, coff: int, c: int, relu_masked: bool = False):
        lib = L.load()
        require_gpu(buf)
        nb, h, w, ld = buf.shape
        out = torch.empty((nb, c, h, w), dtype=torch.float32, device=buf.device)
        L.check(lib.din_nhwc_to_nchw_f32(_ptr(buf), din_dtype(buf), nb, h, w, c, ld, coff, _ptr(out), _stream()), "nhwc_to_nchw")
        ctx.meta = (coff, c, relu_masked)
        ctx.save_for_backward(buf)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        (buf,) = ctx.saved_tensors
        coff, c, relu_masked = ctx.meta
        nb, h, w, ld = buf.shape
        g = g.float().contiguous()
        st = _stream()
        tmp = torch.empty((nb, h, w, c), dtype=torch.float32, device=g.device)
        L.check(lib.din_nchw_f32_to_nhwc(_ptr(g), nb, h, w, c, _ptr(tmp), L.DIN_F32, c, 0, st), "nchw_to_nhwc")
        gb = torch.zeros_like(buf)
        L.check(lib.din_grad_cast_mask(_ptr(tmp), _ptr(buf), _ptr(gb), din_dtype(buf), nb * h * w, c, ld, coff, ld, coff,
                                       int(relu_masked), st), "grad_cast_mask")
        return gb, None, None, None


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    lib = L.load()
    require_gpu(p, g, m, v)
    L.check(lib.din_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                              _stream()), "adam_step")
