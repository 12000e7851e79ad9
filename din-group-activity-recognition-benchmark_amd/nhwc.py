"""NHWC execution graph for the backbone conv stacks (rows V, I, M of SURVEY section 8).

A backbone is described once as a static list of ops over *views* (tensor id, channel offset, channels) of
pixel-major NHWC buffers, then run forward and backward through libdin_hip.so by ONE torch.autograd.Function.
That gives the MI355X design its properties:
  * torch.cat(dim=1) (Inception blocks, infer_model.py:172) costs nothing: producers write channel ranges;
  * ReLU backward is fused into whatever kernel produces a gradient (conv dgrad epilogue / pool backward);
  * BatchNorm(eval) is folded into the packed filters (scale) and the conv bias (shift);
  * images stay uint8 until the fused prep loader (no fp32 image copies, SURVEY 7 hard part 7).
Nothing here falls back to ATen: every arithmetic op is a C-ABI call and raises if the library is absent.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def torch_dtype(dt: int):
    return torch.float32 if dt == L.DIN_F32 else torch.bfloat16


def din_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.DIN_F32
    if t.dtype == torch.bfloat16:
        return L.DIN_BF16
    raise L.DinError(f"unsupported storage dtype {t.dtype}")


def require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise L.DinError("din_amd ops run only on MI355X device tensors (no CPU fallback); got a CPU tensor")
        if not t.is_contiguous():
            raise L.DinError("din_amd ops need contiguous tensors")


# ------------------------------------------------------------------------------------------------
# hooks.  LAUNCH_TIMER: optional context-manager factory (kind, desc, name) wrapped around every conv launch -- the measurement side
# (din_amd/profiling.py, used by bench.py) installs it; the product path itself carries no timing code.
# ------------------------------------------------------------------------------------------------
import contextlib as _contextlib
LAUNCH_TIMER = None
TIMING_ACTIVE = lambda: LAUNCH_TIMER is not None      # noqa: E731  (profiling.install narrows it to "events are being recorded right now")


def _timed(kind, d, name=""):
    return LAUNCH_TIMER(kind, d, name) if LAUNCH_TIMER is not None else _contextlib.nullcontext()


# Called as GRAD_HOOK(weight, dweight) right after a conv layer's weight gradient has been enqueued (backward order: last layer
# first).  parallel.GradBuckets uses it to start a bucket's all-reduce while the rest of the backward pass is still running.
GRAD_HOOK = None
# diagnostics (tools/bf16_grad_cosine.py): GRAD_TAP(tensor id, producing op's name, complete gradient buffer [nb, h, w, c_total]) is called in
# graph_backward each time the reverse pass reaches a producer of a tensor -- its gradient is complete then.  None in production.
GRAD_TAP = None
# Called as GRAD_BUFFER(weight) before a conv layer's weight gradient is computed; may return the fp32 tensor (weight's shape) the kernel
# should write into -- parallel.GradBuckets hands out the weight's slot of its flat all-reduce bucket, so no packing copy is needed.
GRAD_BUFFER = None
# Called as GRAD_ASSIGN(weight, dweight) for every parameter gradient when the backward pass returns; True = the callee installed the
# tensor as weight.grad itself (a gradient that was written in place into its all-reduce bucket slot), autograd then gets None for it:
# handing autograd a tensor that somebody else still references makes AccumulateGrad clone it -- one copy launch per parameter per step.
GRAD_ASSIGN = None


def _dw_buffer(w: torch.Tensor) -> torch.Tensor:
    if GRAD_BUFFER is not None:
        buf = GRAD_BUFFER(w)
        if buf is not None:
            return buf
    return torch.empty_like(w)
# Host-layer fusion switches.  None = ask the option table the library's kernel switches live in (din_set_option, `_lib.host_flag`: production
# sets none, nothing reads the environment); a test may also pin one by assigning True / False to the module attribute.
FUSE_1X1_DGRAD = None        # DIN_FUSE_1X1 (default on): fuse the dgrads of 1x1 convs that read the same tensor
FUSE_FWD_SIBLINGS = None     # DIN_FUSE_FWD (on): run Graph.fwd_groups (sibling 1x1 convs) as one two-destination launch
FUSE_WGRAD_SIBLINGS = None   # DIN_FUSE_WGRAD (on): ... and the wgrads of the members that share the second tensor as one launch
FUSE_WGRAD_1X1 = None        # DIN_FUSE_WGRAD_1X1 (on): weight gradients of ALL 1x1 convs reading one view in one launch (din_conv1x1_wgrad_multi)
FUSE_DGRAD_X = None          # DIN_FUSE_DGRAD_X (on): a lone 1x1's dgrad rides in the strided sibling's launches (din_conv_dgrad_x: Mixed_6a)
GROUP_WGRAD = None           # DIN_GROUP_WGRAD (on): weight gradients of up to GROUP_WGRAD_MAX layers share one launch (din_conv_wgrad_group)
# A queue is flushed at GROUP_WGRAD_MAX layers or when the operands it holds back (input + output gradient of every queued layer) reach
# GROUP_WGRAD_MB: measured on one box (profiles/r06_wgrad_group.txt) -- 4 clips (31 MB per Mixed_6 layer): 12 layers 8.80 ms, 8 layers 8.84,
# no grouping 9.74; 32 clips (248 MB per layer): 4 layers 663.6 clips/s = no grouping 663.0, 8 layers 659.7, 16 layers 650 (a deferred launch
# reads operands that have left the 256 MB last-level cache; the slice partials it saves are ~5 % of a launch at that size).
GROUP_WGRAD_MAX = 12         # (<= the library's WGRAD_GROUP_MAX = 16; option DIN_GROUP_WGRAD_MAX overrides)
GROUP_WGRAD_MB = 768         # (option DIN_GROUP_WGRAD_MB overrides)
_SWITCHES = {"FUSE_1X1_DGRAD": ("DIN_FUSE_1X1", True), "FUSE_FWD_SIBLINGS": ("DIN_FUSE_FWD", True), "FUSE_WGRAD_SIBLINGS": ("DIN_FUSE_WGRAD", True),
             "FUSE_WGRAD_1X1": ("DIN_FUSE_WGRAD_1X1", True), "FUSE_DGRAD_X": ("DIN_FUSE_DGRAD_X", True), "GROUP_WGRAD": ("DIN_GROUP_WGRAD", True), "WGRAD_SIDE_STREAM": ("DIN_WGRAD_STREAM", False)}


def switch(attr: str) -> bool:
    v = globals()[attr]
    if v is not None:
        return bool(v)
    name, default = _SWITCHES[attr]
    return L.host_flag(name, default)



# ------------------------------------------------------------------------------------------------
# shared split-K / wgrad workspace (per device; ops on one stream are serialised, so one buffer is safe)
# ------------------------------------------------------------------------------------------------
_WS: Dict[Tuple[int, str], torch.Tensor] = {}


_SIDE: Dict[int, "torch.cuda.Stream"] = {}
WGRAD_SIDE_STREAM = None     # DIN_WGRAD_STREAM (off), opt-in: measured 472 -> 351 clips/s (the LDS-heavy kernels of the two streams evict each other)


def side_stream(device) -> "torch.cuda.Stream":
    """Second HIP stream per device: wgrad (+ slice reduce, BN parameter gradients) of layer l runs here while the main stream goes
    on with the dgrad chain -- both only depend on the gradient at the layer's output, and each kernel's last partial round of
    workgroups leaves CUs idle that the other stream's kernel can fill."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def workspace(nbytes: int, device, tag: str = "") -> Tuple[Optional[torch.Tensor], int]:
    if nbytes <= 0:
        return None, 0
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws, ws.numel()


# ------------------------------------------------------------------------------------------------
# graph description
# ------------------------------------------------------------------------------------------------
@dataclass
class TensorSpec:
    h: int
    w: int
    c: int                  # pixel stride (total channels incl. padding)
    relu_masked: bool = False   # some producer applies ReLU -> gradients into it are masked by (value > 0)


@dataclass
class View:
    tid: int
    coff: int
    c: int


@dataclass
class Op:
    kind: str               # conv | maxpool | avgpool | bilinear
    src: View
    dst: View
    name: str = ""
    k: Tuple[int, int] = (1, 1)
    s: Tuple[int, int] = (1, 1)
    p: Tuple[int, int] = (0, 0)
    relu: bool = False
    bias: bool = False      # conv has its own bias parameter (VGG)
    bn: bool = False        # conv followed by BatchNorm (Inception BasicConv2d)
    pooled: Optional[Tuple[int, int, int]] = None   # (k, s, p): the reference runs avg_pool2d(k, s, p) on src BEFORE this 1x1 conv
                                                    # (InceptionA/C branch_pool); executed as conv1x1 -> avgpool(+shift, ReLU) on cout channels
    mid: Optional[View] = None                      # pooled convs only: where the raw conv output goes when the conv runs inside a forward group


@dataclass
class Graph:
    tensors: List[TensorSpec] = field(default_factory=list)
    ops: List[Op] = field(default_factory=list)
    input_tid: int = 0
    output_tids: List[int] = field(default_factory=list)
    cin_image: int = 3
    u8_ok: Dict[Tuple[int, int], bool] = field(default_factory=dict)   # (frames, dtype) -> the image layer takes uint8 frames (_accepts_u8_frames)
    fwd_groups: List[Tuple[int, ...]] = field(default_factory=list)   # consecutive sibling conv ops (same source view, same 1x1 geometry) whose
                                                                      # forward runs as ONE launch: first member -> its own view, the rest -> adjacent views of one tensor

    def add_tensor(self, h, w, c) -> int:
        self.tensors.append(TensorSpec(h, w, c))
        return len(self.tensors) - 1

    def param_names(self) -> List[str]:
        """Parameter (and BN buffer) names in the order the autograd Function receives them."""
        names = []
        for op in self.ops:
            if op.kind != "conv":
                continue
            if op.bn:
                names += [op.name + ".conv.weight", op.name + ".bn.weight", op.name + ".bn.bias",
                          op.name + ".bn.running_mean", op.name + ".bn.running_var"]
            else:
                names.append(op.name + ".weight")
                if op.bias:
                    names.append(op.name + ".bias")
        return names


def conv_out(h, k, s, p, d=1):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def group_view(op: Op) -> Optional[View]:
    """where a forward-group member's conv output lands: its dst, or `mid` for a conv whose pool was commuted behind it"""
    return op.mid if op.pooled is not None else op.dst


class GraphBuilder:
    """Tiny helper used by backbone/backbone.py to lay out the layer tables."""

    def __init__(self, h: int, w: int, cpad_image: int):
        self.g = Graph()
        self.g.input_tid = self.g.add_tensor(h, w, cpad_image)

    def tensor(self, h, w, c) -> int:
        return self.g.add_tensor(h, w, c)

    def full(self, tid) -> View:
        return View(tid, 0, self.g.tensors[tid].c)

    def conv(self, name, src: View, cout, k, s=(1, 1), p=(0, 0), relu=True, bias=False, bn=False,
             dst: Optional[View] = None, pooled: Optional[Tuple[int, int, int]] = None, mid: Optional[View] = None) -> View:
        ts = self.g.tensors[src.tid]
        oh, ow = conv_out(ts.h, k[0], s[0], p[0]), conv_out(ts.w, k[1], s[1], p[1])
        if pooled is not None:
            assert tuple(k) == (1, 1) and tuple(s) == (1, 1) and tuple(p) == (0, 0) and pooled[1] == 1 and cout % 8 == 0
            assert conv_out(ts.h, pooled[0], 1, pooled[2]) == ts.h, "commuted pool must keep the grid"
        if dst is None:
            dst = self.full(self.tensor(oh, ow, cout))
        td = self.g.tensors[dst.tid]
        assert (td.h, td.w) == (oh, ow) and dst.c == cout, (name, td, oh, ow)
        td.relu_masked = td.relu_masked or relu
        assert mid is None or pooled is not None
        self.g.ops.append(Op("conv", src, dst, name, tuple(k), tuple(s), tuple(p), relu, bias, bn, pooled, mid))
        return dst

    def fuse_forward(self, n_last: int) -> None:
        """Declare the last n_last conv ops a forward group (see Graph.fwd_groups); checks the layout the fused launch needs.  Members
        with a commuted pool (raw conv output into `mid`, bias + ReLU after the pool) must come last."""
        idx = tuple(range(len(self.g.ops) - n_last, len(self.g.ops)))
        ops = [self.g.ops[i] for i in idx]
        a = ops[0]
        assert a.pooled is None
        assert all(o.kind == "conv" and o.src == a.src and o.k == a.k and o.s == a.s and o.p == a.p and o.bn == a.bn and o.bias == a.bias
                   and o.relu == a.relu for o in ops), "fused siblings must share source and geometry"
        second = [group_view(o) for o in ops[1:]]
        assert all(v is not None and v.tid == second[0].tid for v in second) and second[0].tid != a.dst.tid
        for x, y in zip(second, second[1:]):
            assert y.coff == x.coff + x.c, "second-destination views must be adjacent"
        pooled = [o.pooled is not None for o in ops]
        assert pooled == sorted(pooled), "pooled members last"
        self.g.fwd_groups.append(idx)

    def pool(self, kind, src: View, k, s, p, dst: Optional[View] = None) -> View:
        ts = self.g.tensors[src.tid]
        oh, ow = conv_out(ts.h, k, s, p), conv_out(ts.w, k, s, p)
        if dst is None:
            dst = self.full(self.tensor(oh, ow, src.c))
        td = self.g.tensors[dst.tid]
        assert (td.h, td.w) == (oh, ow) and dst.c == src.c
        self.g.ops.append(Op(kind, src, dst, kind, (k, k), (s, s), (p, p)))
        return dst

    def bilinear(self, src: View, dst: View) -> View:
        assert dst.c == src.c
        self.g.ops.append(Op("bilinear", src, dst, "bilinear"))
        return dst


# ------------------------------------------------------------------------------------------------
# executor
# ------------------------------------------------------------------------------------------------
def _conv_desc(g: Graph, op: Op, nb: int, dt: int, cin: Optional[int] = None) -> L.ConvDesc:
    ts, td = g.tensors[op.src.tid], g.tensors[op.dst.tid]
    d = L.ConvDesc()
    d.nb, d.h, d.w, d.cin = nb, ts.h, ts.w, cin if cin is not None else op.src.c
    d.oh, d.ow, d.cout = td.h, td.w, op.dst.c
    d.kh, d.kw = op.k
    d.sh, d.sw = op.s
    d.ph, d.pw = op.p
    d.dh = d.dw = 1
    d.ldi, d.cioff, d.ldo, d.cooff = ts.c, op.src.coff, td.c, op.dst.coff
    d.dtype = dt
    return d


def _copy_desc(d: L.ConvDesc) -> L.ConvDesc:
    return L.ConvDesc.from_buffer_copy(d)


def _pool_desc(g: Graph, op: Op, nb: int, dt: int) -> L.PoolDesc:
    ts, td = g.tensors[op.src.tid], g.tensors[op.dst.tid]
    d = L.PoolDesc()
    d.nb, d.h, d.w, d.c, d.oh, d.ow = nb, ts.h, ts.w, op.src.c, td.h, td.w
    d.k, d.stride, d.pad = op.k[0], op.s[0], op.p[0]
    d.ldi, d.cioff, d.ldo, d.cooff = ts.c, op.src.coff, td.c, op.dst.coff
    d.dtype = dt
    return d


def _pooled_descs(g: Graph, op: Op, nb: int, dt: int) -> Tuple[L.ConvDesc, L.PoolDesc]:
    """conv1x1 src -> tmp [nb,h,w,cout] and avgpool tmp -> dst view for a conv whose pool was commuted behind it"""
    ts, td = g.tensors[op.src.tid], g.tensors[op.dst.tid]
    d = _conv_desc(g, op, nb, dt)
    d.ldo, d.cooff = op.dst.c, 0
    pd = L.PoolDesc()
    pd.nb, pd.h, pd.w, pd.c, pd.oh, pd.ow = nb, ts.h, ts.w, op.dst.c, td.h, td.w
    pd.k, pd.stride, pd.pad = op.pooled
    pd.ldi, pd.cioff, pd.ldo, pd.cooff = op.dst.c, 0, td.c, op.dst.coff
    pd.dtype = dt
    return d, pd


BN_EPS = 1e-3      # torchvision BasicConv2d


class _BnTables:
    """device-side pointer / offset tables of a graph's BatchNorm layers (parameters keep their addresses across steps)"""
    def __init__(self, key, ptrs, offs, off_list):
        self.key, self.ptrs, self.offs, self.off_list = key, ptrs, offs, off_list
        self.n, self.total = len(off_list) - 1, off_list[-1]


def _bn_tables(g: "Graph", params: Sequence[torch.Tensor], dev) -> Optional[_BnTables]:
    rows, off_list, pos = [], [0], 0
    for op in g.ops:
        if op.kind != "conv":
            continue
        if op.bn:
            gamma, beta, mean, var = params[pos + 1:pos + 5]
            rows.append([gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr()])
            off_list.append(off_list[-1] + gamma.numel())
            pos += 5
        else:
            pos += 2 if op.bias else 1
    if not rows:
        return None
    key = tuple(r[0] for r in rows)
    cached = getattr(g, "_bn_tables", None)
    if cached is not None and cached.key == key and cached.ptrs.device == dev:
        return cached
    tb = _BnTables(key, torch.tensor(rows, dtype=torch.int64).to(dev), torch.tensor(off_list, dtype=torch.int32).to(dev), off_list)
    g._bn_tables = tb
    return tb


PACK_CHUNK = 16384


class _PackCache:
    """persistent packed filter banks of a graph (+ the device table that lets ONE launch repack all of them every step)"""
    pass


def _pack_cache(g: Graph, params: Sequence[torch.Tensor], dt: int, dev, with_transposed: bool, bn: Optional[_BnTables]) -> _PackCache:
    lib = L.load()
    key = (tuple(p.data_ptr() for p in params), dt, bool(with_transposed), str(dev))
    pc = getattr(g, "_pack_cache", None)
    if pc is not None and pc.key == key:
        return pc
    pc = _PackCache()
    pc.key = key
    tdt = torch_dtype(dt)
    pc.bn_scale = torch.empty(bn.total, dtype=torch.float32, device=dev) if bn is not None else None
    pc.bn_shift = torch.empty(bn.total, dtype=torch.float32, device=dev) if bn is not None else None
    pc.wpk, pc.wpt = {}, {}
    descs, layer_of, chunk_index = [], [], []
    pos, bn_i = 0, 0
    src_of: Dict[int, tuple] = {}
    pc.wfused = {}
    for oi, op in enumerate(g.ops):
        if op.kind != "conv":
            continue
        cin = g.cin_image if op.src.tid == g.input_tid else op.src.c
        d = _conv_desc(g, op, 1, dt, cin)
        w = params[pos]
        scale_ptr = None
        if op.bn:
            o0 = bn.off_list[bn_i]
            scale_ptr = C.c_void_p(pc.bn_scale.data_ptr() + 4 * o0)
            bn_i += 1
            pos += 5
        else:
            pos += 2 if op.bias else 1
        src_of[oi] = (w, scale_ptr)
        for transposed in ((0, 1) if (with_transposed and op.src.tid != g.input_tid) else (0,)):
            n_el = lib.din_conv_packed_elems(C.byref(d), transposed)
            buf = torch.empty(n_el, dtype=tdt, device=dev)
            (pc.wpt if transposed else pc.wpk)[oi] = buf
            pd = L.PackDesc()
            L.check(lib.din_conv_pack_desc(C.byref(d), _ptr(w), scale_ptr, _ptr(buf), transposed, C.byref(pd)), "conv_pack_desc")
            li = len(descs)
            descs.append(pd)
            nchunk = (n_el + PACK_CHUNK - 1) // PACK_CHUNK
            layer_of += [li] * nchunk
            chunk_index += list(range(nchunk))
    # forward groups: the siblings' banks packed row after row into ONE bank (each member writes exactly its own rows; the tail rows of
    # the zero-initialised buffer stay zero), so the fused launch sees a single conv with cout = sum of the members
    esz = 4 if dt == L.DIN_F32 else 2
    for grp in g.fwd_groups:
        a = g.ops[grp[0]]
        dF = _conv_desc(g, a, 1, dt, a.src.c)
        dF.cout = sum(g.ops[i].dst.c for i in grp)
        fused = torch.zeros(lib.din_conv_packed_elems(C.byref(dF), 0), dtype=tdt, device=dev)
        pc.wfused[grp[0]] = fused
        row0 = 0
        for oi in grp:
            op = g.ops[oi]
            d = _conv_desc(g, op, 1, dt, op.src.c)
            w, scale_ptr = src_of[oi]
            pd = L.PackDesc()
            L.check(lib.din_conv_pack_desc(C.byref(d), _ptr(w), scale_ptr, _ptr(fused), 0, C.byref(pd)), "conv_pack_desc(fused)")
            pd.out = fused.data_ptr() + row0 * pd.kelems * esz
            pd.rows_pad = pd.rows
            li = len(descs)
            descs.append(pd)
            nchunk = (pd.rows * pd.kelems + PACK_CHUNK - 1) // PACK_CHUNK
            layer_of += [li] * nchunk
            chunk_index += list(range(nchunk))
            row0 += op.dst.c
    raw = (L.PackDesc * len(descs))(*descs)
    host = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone()
    pc.table = host.to(dev)
    pc.layer_of = torch.tensor(layer_of, dtype=torch.int32).to(dev)
    pc.chunk_index = torch.tensor(chunk_index, dtype=torch.int32).to(dev)
    pc.nblocks = len(layer_of)
    g._pack_cache = pc
    return pc


BN_MOMENTUM = 0.1  # torch.nn.BatchNorm2d default (torchvision BasicConv2d)


def _bn_train_forward(lib, raw: torch.Tensor, dt: int, rows: int, cout: int, gamma, beta, mean, var, relu: bool, dst: torch.Tensor,
                      ldd: int, coffd: int, st):
    """batch-statistics BatchNorm (+ReLU) of a raw conv output [rows][cout] into the channel view (ldd, coffd) of dst; updates the
    running statistics in place; returns (batch mean, rstd) for the backward pass"""
    dev = raw.device
    ws = torch.empty(lib.din_bn_workspace(rows, cout) // 8, dtype=torch.float64, device=dev)   # reduced sums + one slab per workgroup
    ab = torch.empty(4 * cout, dtype=torch.float32, device=dev)
    a, b, bmean, rstd = ab[:cout], ab[cout:2 * cout], ab[2 * cout:3 * cout], ab[3 * cout:]
    # fp64 accumulation with exact products, fixed summation order (csrc/bn.hip): no shift needed, bit-reproducible statistics
    L.check(lib.din_bn_stats(_ptr(raw), dt, rows, cout, cout, 0, None, _ptr(ws), st), "bn_stats")
    L.check(lib.din_bn_finalize(_ptr(ws), lib.din_bn_parts(rows), rows, cout, _ptr(gamma), _ptr(beta), BN_EPS, BN_MOMENTUM, _ptr(mean), _ptr(var),
                                _ptr(a), _ptr(b), _ptr(bmean), _ptr(rstd), None, st), "bn_finalize")
    L.check(lib.din_bn_apply(_ptr(raw), dt, rows, cout, cout, 0, _ptr(a), _ptr(b), int(relu), _ptr(dst), ldd, coffd, st), "bn_apply")
    return bmean, rstd


def graph_forward(g: Graph, image_buf: torch.Tensor, params: Sequence[torch.Tensor], dt: int, save_for_backward: bool = True,
                  bn_train: bool = False):
    """Run the graph.  image_buf: NHWC [nb,h,w,cpad] of dtype dt.  Returns (bufs, aux) for backward.
    bn_train: BatchNorm layers normalise with the statistics of this batch and update their running statistics (the reference's stage-2
    default for Inception-v3: model.train() without set_bn_eval, train_net_dynamic.py:98-100,170-172).  Each BasicConv2d then runs as
    conv (unscaled filters, raw output kept for backward) -> statistics -> normalise + ReLU into the consumer's view; the folded-BN
    fusions (sibling launches, shift in the conv / pool epilogue) do not apply."""
    lib = L.load()
    nb = image_buf.shape[0]
    dev = image_buf.device
    tdt = torch_dtype(dt)
    bufs: List[Optional[torch.Tensor]] = [None] * len(g.tensors)
    bufs[g.input_tid] = image_buf
    it = iter(params)
    aux = []
    st = _stream()
    bn = _bn_tables(g, params, dev)
    pc = _pack_cache(g, params, dt, dev, save_for_backward, bn)
    bn_train = bool(bn_train) and bn is not None
    if bn_train:
        bn_scale, bn_shift = pc.bn_scale, pc.bn_shift
        bn_scale.fill_(1.0)                                   # filters are packed unscaled: the normalisation happens after the conv
    elif bn is not None:
        # every BatchNorm layer of the graph folded in ONE launch into flat scale / shift arrays (views per layer below)
        bn_scale, bn_shift = pc.bn_scale, pc.bn_shift
        L.check(lib.din_bn_fold_multi(_ptr(bn.ptrs), _ptr(bn.offs), bn.n, bn.total, BN_EPS, _ptr(bn_scale), _ptr(bn_shift), st), "bn_fold_multi")
    # ... and every filter bank (forward and, when training, dgrad orientation) repacked with the folded scale in ONE launch
    L.check(lib.din_conv_pack_multi(_ptr(pc.table), _ptr(pc.layer_of), _ptr(pc.chunk_index), pc.nblocks, PACK_CHUNK, st), "conv_pack_multi")
    bn_i = 0
    group_of = {grp[0]: grp for grp in g.fwd_groups} if (switch("FUSE_FWD_SIBLINGS") and not bn_train) else {}
    fused_done = set()                                     # members whose output the group launch already produced
    for oi, op in enumerate(g.ops):
        td = g.tensors[op.dst.tid]
        if bufs[op.dst.tid] is None:
            bufs[op.dst.tid] = torch.empty((nb, td.h, td.w, td.c), dtype=tdt, device=dev)
        src, dst = bufs[op.src.tid], bufs[op.dst.tid]
        if op.kind == "conv":
            cin = g.cin_image if op.src.tid == g.input_tid else op.src.c
            d = _conv_desc(g, op, nb, dt, cin)
            if op.src.tid == g.input_tid and src.dtype == torch.uint8:
                d.in_u8, d.ldi, d.cioff = 1, 8, 0                      # raw uint8 frames [nb,3,h,w]: the image layer normalises on load
            w = next(it)
            scale = shift = None
            if op.bn:
                gamma, beta, mean, var = next(it), next(it), next(it), next(it)
                o0, o1 = bn.off_list[bn_i], bn.off_list[bn_i + 1]
                bn_i += 1
                scale, shift = bn_scale[o0:o1], bn_shift[o0:o1]
                bias = shift
            else:
                bias = next(it) if op.bias else None
            if op.bn and bn_train:
                # conv -> raw [nb,oh,ow,cout] (-> commuted pool: also linear, so avgpool(conv(x)) == conv(avgpool(x)) holds before the
                # normalisation) -> batch statistics -> normalise + ReLU into dst's view
                dR = _conv_desc(g, op, nb, dt, cin)
                dR.in_u8, dR.ldi, dR.cioff = d.in_u8, d.ldi, d.cioff      # (uint8 frames feed the image layer in this mode too)
                dR.ldo, dR.cooff = op.dst.c, 0
                raw = torch.empty((nb, g.tensors[op.src.tid].h if op.pooled is not None else td.h,
                                   g.tensors[op.src.tid].w if op.pooled is not None else td.w, op.dst.c), dtype=tdt, device=dev)
                if op.pooled is not None:
                    dR.oh, dR.ow = raw.shape[1], raw.shape[2]
                ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(dR), 0), dev)
                with _timed("fwd", dR, op.name):
                    L.check(lib.din_conv_fwd(C.byref(dR), _ptr(src), _ptr(pc.wpk[oi]), None, _ptr(raw), 0, _ptr(ws), wsb, st), "conv_fwd " + op.name)
                if op.pooled is not None:
                    _, pd = _pooled_descs(g, op, nb, dt)
                    pd.ldo, pd.cooff = op.dst.c, 0
                    pooled_raw = torch.empty((nb, td.h, td.w, op.dst.c), dtype=tdt, device=dev)
                    L.check(lib.din_avgpool_fwd(C.byref(pd), _ptr(raw), _ptr(pooled_raw), None, 0, st), "avgpool_fwd(raw)")
                    raw = pooled_raw
                bmean, rstd = _bn_train_forward(lib, raw, dt, nb * td.h * td.w, op.dst.c, gamma, beta, mean, var, op.relu, dst, td.c,
                                                op.dst.coff, st)
                aux.append((None, raw if save_for_backward else None, bmean, rstd))
                continue
            if oi in fused_done:
                if op.pooled is not None:                   # the group launch left the raw conv output in `mid`: pool + shift + ReLU into dst
                    _, pd = _pooled_descs(g, op, nb, dt)
                    pd.ldi, pd.cioff = g.tensors[op.mid.tid].c, op.mid.coff
                    L.check(lib.din_avgpool_fwd(C.byref(pd), _ptr(bufs[op.mid.tid]), _ptr(dst), _ptr(bias),
                                                L.CONV_BIAS | (L.CONV_RELU if op.relu else 0), st), "avgpool_fwd(epilogue)")
                aux.append((scale,))
                continue
            if oi in group_of and op.bn:
                # sibling convs of one source: ONE launch over the concatenated filter bank, first member into its own view, the others into
                # the adjacent views of their shared tensor (din_conv_fwd2).  Shapes that would run split-K fall through to separate launches.
                grp = group_of[oi]
                rest = [g.ops[i] for i in grp[1:]]
                ctot = op.dst.c + sum(o.dst.c for o in rest)
                craw = ctot - sum(o.dst.c for o in rest if o.pooled is not None)
                dF = _conv_desc(g, op, nb, dt, cin)
                dF.cout = ctot
                v2 = group_view(rest[0])
                t2 = v2.tid
                td2 = g.tensors[t2]
                if lib.din_conv_workspace_bytes(C.byref(dF), 0) == 0 and bn.off_list[bn_i - 1 + len(grp)] - o0 == ctot:
                    if bufs[t2] is None:
                        bufs[t2] = torch.empty((nb, td2.h, td2.w, td2.c), dtype=tdt, device=dev)
                    flags = L.CONV_BIAS | (L.CONV_RELU if op.relu else 0)
                    with _timed("fwd", dF, "+".join(g.ops[i].name for i in grp)):
                        L.check(lib.din_conv_fwd2(C.byref(dF), _ptr(src), _ptr(pc.wfused[oi]), _ptr(bn_shift[o0:o0 + ctot]), _ptr(dst),
                                                  _ptr(bufs[t2]), td2.c, v2.coff, op.dst.c, craw if craw < ctot else 0, flags, None, 0, st),
                                "conv_fwd2 " + op.name)
                    fused_done.update(grp[1:])
                    aux.append((scale,))
                    continue
            wpk = pc.wpk[oi]
            flags = (L.CONV_BIAS if bias is not None else 0) | (L.CONV_RELU if op.relu else 0)
            if op.pooled is not None:
                d, pd = _pooled_descs(g, op, nb, dt)
                tmp = torch.empty((nb, pd.h, pd.w, pd.c), dtype=tdt, device=dev)
                ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 0), dev)
                with _timed("fwd", d, op.name):
                    L.check(lib.din_conv_fwd(C.byref(d), _ptr(src), _ptr(wpk), None, _ptr(tmp), 0, _ptr(ws), wsb, st), "conv_fwd " + op.name)
                L.check(lib.din_avgpool_fwd(C.byref(pd), _ptr(tmp), _ptr(dst), _ptr(bias), flags, st), "avgpool_fwd(epilogue)")
            else:
                ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 0), dev)
                with _timed("fwd", d, op.name):
                    L.check(lib.din_conv_fwd(C.byref(d), _ptr(src), _ptr(wpk), _ptr(bias), _ptr(dst), flags, _ptr(ws), wsb, st),
                            "conv_fwd " + op.name)
            aux.append((scale,))
        elif op.kind == "maxpool":
            d = _pool_desc(g, op, nb, dt)
            # save the winning-tap map when the pooled tensor is a ReLU output: backward then never re-reads the input
            amax = None
            if save_for_backward and g.tensors[op.src.tid].relu_masked and op.src.tid != g.input_tid:
                amax = torch.empty((nb, td.h, td.w, op.src.c), dtype=torch.uint8, device=dev)
            L.check(lib.din_maxpool_fwd(C.byref(d), _ptr(src), _ptr(dst), _ptr(amax), st), "maxpool_fwd")
            aux.append((amax,))
        elif op.kind in ("avgpool", "bilinear"):
            d = _pool_desc(g, op, nb, dt)
            if op.kind == "avgpool":
                L.check(lib.din_avgpool_fwd(C.byref(d), _ptr(src), _ptr(dst), None, 0, st), "avgpool_fwd")
            else:
                L.check(lib.din_bilinear_fwd(C.byref(d), _ptr(src), _ptr(dst), st), "bilinear_fwd")
            aux.append(())
        else:
            raise L.DinError("unknown op " + op.kind)
    return bufs, aux


def graph_backward(g: Graph, bufs, aux, params: Sequence[torch.Tensor], dt: int,
                   out_grads: Dict[int, torch.Tensor], need_param_grad: Sequence[bool], bn_train: bool = False):
    """Reverse pass.  out_grads: tid -> gradient buffer (same geometry/dtype as the tensor, already ReLU-masked where
    the tensor is relu_masked).  Returns the list of parameter gradients aligned with `params` (None for buffers)."""
    lib = L.load()
    nb = bufs[g.input_tid].shape[0]
    dev = bufs[g.input_tid].device
    tdt = torch_dtype(dt)
    st = _stream()
    main = torch.cuda.current_stream()
    side = side_stream(dev) if (switch("WGRAD_SIDE_STREAM") and not TIMING_ACTIVE()) else None
    wtag = ""
    gbufs: Dict[int, torch.Tensor] = dict(out_grads)

    def on_side(tensors):
        """the main stream's work so far is visible to the side stream; the listed tensors stay alive for it"""
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        for t_ in tensors:
            if t_ is not None:
                t_.record_stream(side)
        return C.c_void_p(side.cuda_stream)

    gwritten: Dict[int, List[Tuple[int, int]]] = {}          # tid -> channel ranges already written (buffers this pass allocated un-zeroed)

    def partitioned(tid: int) -> bool:
        """True when the readers of tensor tid use pairwise identical-or-disjoint channel views that together cover every producer's view:
        then each channel range of the gradient is first WRITTEN by its first reader (no zero fill, no accumulate), and nothing is read
        before it was written (temporaries of sibling convs that share one tensor)."""
        cons = sorted({(o.src.coff, o.src.c) for o in g.ops if o.src.tid == tid})
        for (a0, ac), (b0, bc_) in zip(cons, cons[1:]):
            if b0 < a0 + ac:
                return False
        for o in g.ops:
            if o.dst.tid == tid:
                lo, hi = o.dst.coff, o.dst.coff + o.dst.c
                for c0, cc in cons:
                    if c0 <= lo < c0 + cc:
                        lo = min(hi, c0 + cc)
                if lo < hi:
                    return False
        return True

    def grad_target(view: View) -> Tuple[torch.Tensor, bool]:
        """gradient buffer of view.tid and whether to accumulate into it"""
        ts = g.tensors[view.tid]
        rng = (view.coff, view.c)
        if view.tid in gbufs:
            wr = gwritten.get(view.tid)
            if wr is None:                                  # handed in by the caller or zero-filled: always add
                return gbufs[view.tid], True
            acc = any(rng[0] < r0 + rc and r0 < rng[0] + rng[1] for r0, rc in wr)
            wr.append(rng)
            return gbufs[view.tid], acc
        full = view.coff == 0 and view.c == ts.c
        if full or partitioned(view.tid):
            buf = torch.empty((nb, ts.h, ts.w, ts.c), dtype=tdt, device=dev)
            gwritten[view.tid] = [rng]
            gbufs[view.tid] = buf
            return buf, False
        buf = torch.zeros((nb, ts.h, ts.w, ts.c), dtype=tdt, device=dev)
        gbufs[view.tid] = buf
        return buf, True

    # index params per conv op
    offsets, pos = [], 0
    for op in g.ops:
        if op.kind == "conv":
            offsets.append(pos)
            pos += 5 if op.bn else (2 if op.bias else 1)
        else:
            offsets.append(-1)
    grads: List[Optional[torch.Tensor]] = [None] * len(params)
    bn = _bn_tables(g, params, dev)
    pcache = _pack_cache(g, params, dt, dev, True, bn)          # the forward pass of this step packed both orientations
    if bn is not None:
        # flat, pre-zeroed accumulators for every layer's shift gradient and <W, dW> dot (one memset instead of two per layer)
        bn_acc = torch.zeros(2 * bn.total, dtype=torch.float32, device=dev)
        bn_dshift, bn_wdot = bn_acc[:bn.total], bn_acc[bn.total:]
        bn_index = {}
        k_ = 0
        for oi_, op_ in enumerate(g.ops):
            if op_.kind == "conv" and op_.bn:
                bn_index[oi_] = k_
                k_ += 1
        bn_touched = False

    # 1x1 / stride-1 convs that read the same view (branch-entry convs of the Inception blocks): their dgrads are fused into ONE
    # multi-source launch, issued when the last member (first in program order) has been visited
    groups: Dict[Tuple[int, int, int], List[int]] = {}
    if switch("FUSE_1X1_DGRAD"):
        for oi, op in enumerate(g.ops):
            if op.kind == "conv" and op.k == (1, 1) and op.s == (1, 1) and op.p == (0, 0) and op.src.tid != g.input_tid:
                groups.setdefault((op.src.tid, op.src.coff, op.src.c), []).append(oi)
    groups = {k: v for k, v in groups.items() if 2 <= len(v) <= 4}
    member_of = {oi: key for key, v in groups.items() for oi in v}
    remaining = {key: len(v) for key, v in groups.items()}
    pending: Dict[Tuple[int, int, int], list] = {key: [] for key in groups}
    # a lone 1x1 / stride-1 conv whose input view is also read by a STRIDED conv earlier in program order (InceptionB: branch3x3dbl_1 beside
    # branch3x3): the reverse pass meets the 1x1 first, parks its dgrad operands, and the strided conv's dgrad carries them (din_conv_dgrad_x)
    x_host: Dict[int, int] = {}                               # 1x1 op index -> strided op index that will carry its dgrad
    x_parked: Dict[int, tuple] = {}                           # strided op index -> (1x1 op index, its gout, pixel stride, channel offset)
    if switch("FUSE_DGRAD_X") and dt == L.DIN_BF16:
        for oi, op in enumerate(g.ops):
            if (op.kind == "conv" and op.k == (1, 1) and op.s == (1, 1) and op.p == (0, 0) and op.src.tid != g.input_tid and oi not in member_of
                    and op.pooled is None):
                for oj in range(oi - 1, -1, -1):
                    o2 = g.ops[oj]
                    if o2.kind == "conv" and o2.src == op.src and (o2.s[0] > 1 or o2.s[1] > 1) and oj not in x_host.values():
                        x_host[oi] = oj
                        break

    def flush_wgrad_multi(key, items):
        """the deferred weight gradients of a 1x1 group (multi_w): ONE launch reads the block input once and produces every member's dW"""
        nonlocal bn_touched
        by_oi = {it[0]: it for it in items}
        op0 = g.ops[items[0][0]]
        ts0 = g.tensors[op0.src.tid]
        plan, wsb_need = multi_w[key]
        if any(i not in by_oi for mem in plan for i in mem):
            # A member received no gradient (a graph that consumes only some branch outputs -- partial heads; not the backbones here): the
            # fused launch's plan does not apply, so the members that DID get one degrade to the per-layer kernel (this used
            # to raise in the middle of backward, after the weight gradients had already been deferred)
            for oi_, gout_, w_, scale_, ldj, coffj, pre in items:
                opj = g.ops[oi_]
                dj = _conv_desc(g, opj, nb, dt)
                dj.ldo, dj.cooff = ldj, coffj
                o0, o1 = bn.off_list[bn_index[oi_]], bn.off_list[bn_index[oi_] + 1]
                dwj = _dw_buffer(w_)
                ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(dj), 2), dev, wtag)
                with _timed("wgrad", dj, opj.name):
                    L.check(lib.din_conv_wgrad(C.byref(dj), _ptr(bufs[opj.src.tid]), _ptr(gout_), _ptr(dwj), None if pre else _ptr(bn_dshift[o0:o1]),
                                               _ptr(scale_), _ptr(w_), _ptr(bn_wdot[o0:o1]), 2, _ptr(ws), wsb, st), "conv_wgrad " + opj.name)
                grads[offsets[oi_]] = dwj
                if GRAD_HOOK is not None:
                    GRAD_HOOK(w_, dwj)
            bn_touched = True
            return
        srcs = (L.ConvWSrc * len(plan))()
        keep, outs = [], []
        for j, mem in enumerate(plan):
            _oi, gout_, _w, _scale, ldj, coffj, pre = by_oi[mem[0]]
            o0, o1 = bn.off_list[bn_index[mem[0]]], bn.off_list[bn_index[mem[-1]] + 1]
            if len(mem) == 1:
                wj = params[offsets[mem[0]]]
                dwj = _dw_buffer(wj)
            else:
                wj = torch.cat([params[offsets[i]] for i in mem])
                dwj = torch.empty_like(wj)
            keep += [wj, dwj, gout_]
            srcs[j].dout, srcs[j].dw = gout_.data_ptr(), dwj.data_ptr()
            srcs[j].dbias = 0 if pre else bn_dshift[o0:o1].data_ptr()
            srcs[j].scale, srcs[j].w, srcs[j].wdot = pcache.bn_scale[o0:o1].data_ptr(), wj.data_ptr(), bn_wdot[o0:o1].data_ptr()
            srcs[j].cout, srcs[j].ldo, srcs[j].cooff = o1 - o0, ldj, coffj
            outs.append((mem, dwj))
        bn_touched = True
        ws, wsb = workspace(wsb_need, dev, wtag)
        dM = _conv_desc(g, op0, nb, dt)
        dM.cout = sum(s_.cout for s_ in srcs)                    # FLOP accounting of the fused launch
        with _timed("wgrad", dM, "1x1multi:" + "+".join(g.ops[i].name for mem in plan for i in mem)):
            L.check(lib.din_conv1x1_wgrad_multi(len(plan), srcs, dt, nb * ts0.h * ts0.w, op0.src.c, ts0.c, op0.src.coff, _ptr(bufs[op0.src.tid]), 2,
                                                _ptr(ws), wsb, st), "conv1x1_wgrad_multi")
        for mem, dwj in outs:
            r0 = 0
            for i in mem:
                c = g.ops[i].dst.c
                grads[offsets[i]] = dwj if len(mem) == 1 else dwj[r0:r0 + c]
                r0 += c
                if GRAD_HOOK is not None:
                    GRAD_HOOK(params[offsets[i]], grads[offsets[i]])

    # ---- layer-grouped weight gradients (din_conv_wgrad_group).  A weight gradient has no consumer inside the reverse pass, so layers whose
    # kernel allows it (the pipelined wide-bank kernel, equal tile instantiation) are QUEUED with their operands and launched up to
    # GROUP_WGRAD_MAX at a time: the launch's 256 workgroups are shared, every layer is cut into a fraction of the pixel slices it would get
    # alone, and its fp32 slice partials (slices x |dW|, ~50 MB per layer and launch whatever the batch) shrink by the same factor.
    wq: Dict[int, list] = {}
    group_on = switch("GROUP_WGRAD") and dt == L.DIN_BF16 and side is None
    group_max = max(2, min(16, int(L.get_option("DIN_GROUP_WGRAD_MAX") or GROUP_WGRAD_MAX))) if group_on else 0
    group_bytes = int(L.get_option("DIN_GROUP_WGRAD_MB") or GROUP_WGRAD_MB) << 20 if group_on else 0
    wq_bytes: Dict[int, int] = {}

    def wgrad_now(d, x, go, dw_, dshift_, scale_, w_, wdot_, acc, name, after):
        ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 2), dev, wtag)
        with _timed("wgrad", d, name):
            L.check(lib.din_conv_wgrad(C.byref(d), _ptr(x), _ptr(go), _ptr(dw_), _ptr(dshift_), _ptr(scale_), _ptr(w_), _ptr(wdot_), acc,
                                       _ptr(ws), wsb, st), "conv_wgrad " + name)
        if after is not None:
            after()

    def flush_wq(key):
        q = wq.pop(key, None)
        wq_bytes.pop(key, None)
        if not q:
            return
        need = 0
        if len(q) > 1:
            items = (L.ConvWgradItem * len(q))()
            for it, (d, x, go, dw_, dshift_, scale_, w_, wdot_, acc, _name, _after) in zip(items, q):
                it.desc = d
                it.in_, it.dout, it.dw = x.data_ptr(), go.data_ptr(), dw_.data_ptr()
                it.dbias = dshift_.data_ptr() if dshift_ is not None else None
                it.scale = scale_.data_ptr() if scale_ is not None else None
                it.w = w_.data_ptr() if w_ is not None else None
                it.wdot = wdot_.data_ptr() if wdot_ is not None else None
                it.accumulate = acc
            need = lib.din_conv_wgrad_group_workspace(len(q), items)
        if need <= 0:                                           # a single layer (or a list the library does not take as one launch)
            for args in q:
                wgrad_now(*args)
            return
        ws, wsb = workspace(need, dev, "wgroup")
        dG = _copy_desc(q[0][0])
        dG.flops_override = float(sum(2.0 * a[0].nb * a[0].oh * a[0].ow * a[0].cout * a[0].cin * a[0].kh * a[0].kw for a in q))
        with _timed("wgrad", dG, "group:" + "+".join(a[9] for a in q)):
            L.check(lib.din_conv_wgrad_group(len(q), items, _ptr(ws), wsb, st), "conv_wgrad_group " + q[0][9] + "+")
        for a in q:
            if a[10] is not None:
                a[10]()

    def wgrad_launch(d, x, go, dw_, dshift_, scale_, w_, wdot_, acc, name, after=None):
        """one layer's weight gradient: queued for a grouped launch when its kernel allows it, launched at once otherwise"""
        key = lib.din_conv_wgrad_group_key(C.byref(d)) if group_on else 0
        if not key:
            wgrad_now(d, x, go, dw_, dshift_, scale_, w_, wdot_, acc, name, after)
            return
        q = wq.setdefault(key, [])
        q.append((_copy_desc(d), x, go, dw_, dshift_, scale_, w_, wdot_, acc, name, after))
        wq_bytes[key] = wq_bytes.get(key, 0) + 2 * (d.nb * d.h * d.w * d.cin + d.nb * d.oh * d.ow * d.cout)      # bf16 operands held back
        if len(q) >= group_max or wq_bytes[key] >= group_bytes:
            flush_wq(key)

    def flush_group(key):
        items = pending[key]
        if not items:
            return
        if key in multi_w:
            flush_wgrad_multi(key, items)
        op0 = g.ops[items[0][0]]
        ts0 = g.tensors[op0.src.tid]
        gsrc, acc = grad_target(op0.src)
        srcs = (L.ConvSrc * len(items))()
        keep = []
        for j, (oi_, gout_, w_, scale_, ldj, coffj, _pre) in enumerate(items):
            opj = g.ops[oi_]
            dj = _conv_desc(g, opj, nb, dt)
            wpt = pcache.wpt[oi_]
            keep.append(wpt)
            srcs[j].dout, srcs[j].wpk_t = gout_.data_ptr(), wpt.data_ptr()
            srcs[j].cout, srcs[j].ldo, srcs[j].cooff = opj.dst.c, ldj, coffj
        flags = (L.CONV_ACCUM if acc else 0) | (L.CONV_MASK if ts0.relu_masked else 0)
        d0 = _conv_desc(g, op0, nb, dt)
        d0.cout = sum(g.ops[it[0]].dst.c for it in items)          # FLOP accounting of the fused launch
        d0.src_couts = tuple(g.ops[it[0]].dst.c for it in items)   # (measurement side: profiling.LaunchTimer names the instantiation from it)
        with _timed("dgrad", d0, "+".join(g.ops[it[0]].name for it in items)):
            L.check(lib.din_conv1x1_dgrad_multi(len(items), srcs, dt, nb, ts0.h, ts0.w, op0.src.c, ts0.c, op0.src.coff, _ptr(gsrc),
                                                _ptr(bufs[op0.src.tid]) if ts0.relu_masked else None, ts0.c, op0.src.coff, flags, st),
                    "conv1x1_dgrad_multi")
        pending[key] = []

    # sibling groups: the members behind the first one share a tensor -> one wgrad launch, issued when the LAST of them comes up (the
    # reverse pass reaches it first; by then the consumers of every member have written their slice of the shared gradient buffer)
    wgrad_group = {}
    bn_train = bool(bn_train) and bn is not None
    if switch("FUSE_WGRAD_SIBLINGS") and not bn_train:
        for grp in g.fwd_groups:
            mem = tuple(i for i in grp[1:] if g.ops[i].pooled is None)      # (a commuted-pool member keeps its own wgrad: its gradient
            if len(mem) >= 2:                                               #  operand is the un-pooled map, a separate buffer)
                wgrad_group[mem[-1]] = mem
    wgrad_done = set()
    # 1x1 groups whose weight gradients run as ONE launch (din_conv1x1_wgrad_multi): sources = the members, sibling pairs that share a
    # tensor (wgrad_group) counted as one source; the library says whether the group fits its kernel (0 bytes of workspace: it does not)
    multi_w: Dict[Tuple[int, int, int], tuple] = {}
    if switch("FUSE_WGRAD_1X1") and bn is not None and not bn_train and side is None and dt == L.DIN_BF16:
        pair_of = {i: mem for mem in wgrad_group.values() for i in mem}
        for key, members in groups.items():
            if not all(g.ops[i].bn for i in members):
                continue
            plan, seen = [], set()
            for i in members:
                if i in seen:
                    continue
                mem = pair_of.get(i, (i,))
                if not all(j in members for j in mem):
                    mem = (i,)
                plan.append(tuple(mem))
                seen.update(mem)
            probe = (L.ConvWSrc * len(plan))()
            for j, mem in enumerate(plan):
                o = g.ops[mem[0]]
                probe[j].cout = sum(g.ops[i].dst.c for i in mem)
                probe[j].ldo, probe[j].cooff = (o.dst.c, 0) if o.pooled is not None else (g.tensors[o.dst.tid].c, o.dst.coff)
            ts_ = g.tensors[g.ops[members[0]].src.tid]
            need = lib.din_conv1x1_wgrad_multi_workspace(len(plan), probe, dt, nb * ts_.h * ts_.w, g.ops[members[0]].src.c) if 2 <= len(plan) <= 4 else 0
            if need > 0:
                multi_w[key] = (plan, need)
    def dgrad_parked(host_oi):
        """the strided conv that was to carry a parked 1x1 dgrad received no gradient itself: the 1x1's dgrad runs alone"""
        oi_, gout_, ldj, coffj = x_parked.pop(host_oi)
        opj = g.ops[oi_]
        tsj = g.tensors[opj.src.tid]
        dj = _conv_desc(g, opj, nb, dt)
        dj.ldo, dj.cooff = ldj, coffj
        gsrc, acc = grad_target(opj.src)
        flags = (L.CONV_ACCUM if acc else 0) | (L.CONV_MASK if tsj.relu_masked else 0)
        ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(dj), 1), dev)
        with _timed("dgrad", dj, opj.name):
            L.check(lib.din_conv_dgrad(C.byref(dj), _ptr(gout_), _ptr(pcache.wpt[oi_]), _ptr(gsrc),
                                       _ptr(bufs[opj.src.tid]) if tsj.relu_masked else None, tsj.c, opj.src.coff, flags,
                                       _ptr(ws), wsb, st), "conv_dgrad " + opj.name)

    for oi in range(len(g.ops) - 1, -1, -1):
        op = g.ops[oi]
        if op.dst.tid not in gbufs:
            if oi in member_of:                       # a group member without gradient still counts as visited
                remaining[member_of[oi]] -= 1
                if remaining[member_of[oi]] == 0:
                    flush_group(member_of[oi])
            if oi in x_parked:
                dgrad_parked(oi)
            continue                                  # nothing flows back through this op
        gout = gbufs[op.dst.tid]
        if GRAD_TAP is not None:
            GRAD_TAP(op.dst.tid, op.name, gout)
        src_needs_grad = op.src.tid != g.input_tid
        ts = g.tensors[op.src.tid]
        if op.kind == "conv":
            cin = g.cin_image if op.src.tid == g.input_tid else op.src.c
            d = _conv_desc(g, op, nb, dt, cin)
            if op.src.tid == g.input_tid and bufs[g.input_tid].dtype == torch.uint8:
                d.in_u8, d.ldi, d.cioff = 1, 8, 0
            po = offsets[oi]
            w = params[po]
            scale = aux[oi][0]
            g_ld, g_coff = g.tensors[op.dst.tid].c, op.dst.coff        # where the gradient at the conv output lives
            dshift_pre = None
            bn_live = op.bn and bn_train
            if bn_live:
                # batch-statistics BatchNorm backward: gz (masked by the ReLU that follows) -> dy at the raw conv output, dgamma, dbeta
                _, raw, bmean, rstd = aux[oi]
                td_ = g.tensors[op.dst.tid]
                rows = nb * td_.h * td_.w
                sums = torch.empty(lib.din_bn_workspace(rows, op.dst.c) // 8, dtype=torch.float64, device=dev)
                L.check(lib.din_bn_bwd_stats(_ptr(gout), g_ld, g_coff, _ptr(raw), op.dst.c, 0, dt, rows, op.dst.c, _ptr(bmean), _ptr(rstd),
                                             _ptr(sums), st), "bn_bwd_stats")
                dy = torch.empty((nb, td_.h, td_.w, op.dst.c), dtype=tdt, device=dev)
                dgb = torch.empty(2 * op.dst.c, dtype=torch.float32, device=dev)
                L.check(lib.din_bn_bwd_apply(_ptr(gout), g_ld, g_coff, _ptr(raw), op.dst.c, 0, dt, rows, op.dst.c, _ptr(params[po + 1]), _ptr(bmean),
                                             _ptr(rstd), _ptr(sums), _ptr(dy), op.dst.c, 0, _ptr(dgb), _ptr(dgb[op.dst.c:]), st), "bn_bwd_apply")
                grads[po + 1], grads[po + 2] = dgb[:op.dst.c], dgb[op.dst.c:]
                gout, g_ld, g_coff = dy, op.dst.c, 0
                d.ldo, d.cooff = op.dst.c, 0                           # the gradient operand of wgrad / dgrad is the contiguous dy
                if op.pooled is not None:
                    d, pd = _pooled_descs(g, op, nb, dt)
                    gtmp = torch.empty((nb, pd.h, pd.w, pd.c), dtype=tdt, device=dev)
                    pd.ldo, pd.cooff = op.dst.c, 0                     # the pooled-side gradient is the contiguous dy
                    L.check(lib.din_avgpool_bwd(C.byref(pd), _ptr(gout), _ptr(gtmp), None, 0, st), "avgpool_bwd(raw)")
                    gout = gtmp
            elif op.pooled is not None:
                # y = relu(avgpool(conv1x1(x)) + shift): shift gradient = column sums of gout, conv-output gradient = avgpool^T(gout)
                d, pd = _pooled_descs(g, op, nb, dt)
                td_ = g.tensors[op.dst.tid]
                if op.bn or op.bias:
                    if op.bn:                                   # straight into this layer's slice of the flat accumulator
                        dshift_pre = bn_dshift[bn.off_list[bn_index[oi]]:bn.off_list[bn_index[oi] + 1]]
                    else:
                        dshift_pre = torch.empty(op.dst.c, dtype=torch.float32, device=dev)
                    L.check(lib.din_colsum(_ptr(gout), dt, nb * td_.h * td_.w, op.dst.c, td_.c, op.dst.coff, _ptr(dshift_pre), st), "colsum")
                gtmp = torch.empty((nb, pd.h, pd.w, pd.c), dtype=tdt, device=dev)
                L.check(lib.din_avgpool_bwd(C.byref(pd), _ptr(gout), _ptr(gtmp), None, 0, st), "avgpool_bwd(epilogue)")
                gout, g_ld, g_coff = gtmp, pd.c, 0
            # ---- wgrad (+ bias / BN parameter gradients): on the side stream when enabled
            defer_w = oi in member_of and member_of[oi] in multi_w and src_needs_grad      # produced when its 1x1 group is flushed (flush_wgrad_multi)
            hooked = False
            if defer_w or oi in wgrad_done:
                pass                                              # produced by the group launch of a sibling (below)
            elif oi in wgrad_group and op.bn and side is None:
                # the members of a forward group that share the second tensor: their output gradients are adjacent channel views of ONE
                # buffer, their BatchNorm slots adjacent -> one wgrad over the concatenated filter rows; dW lands in one buffer whose
                # row ranges are the members' gradients
                mem = wgrad_group[oi]
                mops = [g.ops[i] for i in mem]
                ctot = sum(o.dst.c for o in mops)
                dF = _conv_desc(g, mops[0], nb, dt, cin)
                dF.cout = ctot
                o0, o1 = bn.off_list[bn_index[mem[0]]], bn.off_list[bn_index[mem[-1]] + 1]
                assert o1 - o0 == ctot
                wcat = torch.cat([params[offsets[i]] for i in mem])
                dwf = torch.empty_like(wcat)
                bn_touched = True
                r0 = 0
                for i, o in zip(mem, mops):
                    grads[offsets[i]] = dwf[r0:r0 + o.dst.c]
                    r0 += o.dst.c

                def _hooks(mem=mem):
                    if GRAD_HOOK is not None:
                        for i in mem:
                            GRAD_HOOK(params[offsets[i]], grads[offsets[i]])
                wgrad_launch(dF, bufs[op.src.tid], gout, dwf, bn_dshift[o0:o1], pcache.bn_scale[o0:o1], wcat, bn_wdot[o0:o1], 2,
                             "+".join(o.name for o in mops), _hooks)
                wgrad_done.update(mem)
            if defer_w:
                dw = None
            elif oi in wgrad_done:
                dw = grads[po]
            elif bn_live:
                dw = _dw_buffer(w)
                ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 2), dev, wtag)
                with _timed("wgrad", d, op.name):
                    L.check(lib.din_conv_wgrad(C.byref(d), _ptr(bufs[op.src.tid]), _ptr(gout), _ptr(dw), None, None, None, None, 0,
                                               _ptr(ws), wsb, st), "conv_wgrad " + op.name)
                grads[po] = dw
            elif op.bn:
                dw = _dw_buffer(w)
                wsbytes = lib.din_conv_workspace_bytes(C.byref(d), 2)
                o0, o1 = bn.off_list[bn_index[oi]], bn.off_list[bn_index[oi] + 1]
                dshift, wdot = bn_dshift[o0:o1], bn_wdot[o0:o1]          # views of the pre-zeroed flat accumulators
                bn_touched = True
                if side is not None:
                    stw = on_side([bufs[op.src.tid], gout, dw, bn_acc, scale])
                    with torch.cuda.stream(side):
                        ws, wsb = workspace(wsbytes, dev, "side")
                    with _timed("wgrad", d, op.name):
                        L.check(lib.din_conv_wgrad(C.byref(d), _ptr(bufs[op.src.tid]), _ptr(gout), _ptr(dw),
                                                   None if dshift_pre is not None else _ptr(dshift), _ptr(scale),
                                                   _ptr(w), _ptr(wdot), 2, _ptr(ws), wsb, stw), "conv_wgrad " + op.name)
                else:
                    hooked = True                                  # (the hook fires when the launch that produces dw has been enqueued)
                    wgrad_launch(d, bufs[op.src.tid], gout, dw, None if dshift_pre is not None else dshift, scale, w, wdot, 2, op.name,
                                 (lambda w=w, dw=dw: GRAD_HOOK(w, dw)) if GRAD_HOOK is not None else None)
                grads[po] = dw
            else:
                dw = _dw_buffer(w)
                wsbytes = lib.din_conv_workspace_bytes(C.byref(d), 2)
                db = (dshift_pre if dshift_pre is not None else torch.empty_like(params[po + 1])) if op.bias else None
                if side is not None:
                    stw = on_side([bufs[op.src.tid], gout, dw, db])
                    with torch.cuda.stream(side):
                        ws, wsb = workspace(wsbytes, dev, "side")
                else:
                    stw = st
                    ws, wsb = workspace(wsbytes, dev, wtag)
                with _timed("wgrad", d, op.name):
                    L.check(lib.din_conv_wgrad(C.byref(d), _ptr(bufs[op.src.tid]), _ptr(gout), _ptr(dw),
                                               None if dshift_pre is not None else _ptr(db), None, None, None, 0,
                                               _ptr(ws), wsb, stw), "conv_wgrad " + op.name)
                grads[po] = dw
                if op.bias:
                    grads[po + 1] = db
            if GRAD_HOOK is not None and side is None and oi not in wgrad_done and not defer_w and not hooked:
                GRAD_HOOK(w, dw)
            # ---- dgrad
            if src_needs_grad and oi in member_of:
                key = member_of[oi]
                pending[key].append((oi, gout, w, scale, g_ld, g_coff, dshift_pre is not None))
                remaining[key] -= 1
                if remaining[key] == 0:
                    flush_group(key)
            elif src_needs_grad and oi in x_host and dshift_pre is None:
                x_parked[x_host[oi]] = (oi, gout, g_ld, g_coff)        # carried by the strided sibling's dgrad (below, when the pass reaches it)
            elif src_needs_grad:
                gsrc, acc = grad_target(op.src)
                wpt = pcache.wpt[oi]
                flags = (L.CONV_ACCUM if acc else 0) | (L.CONV_MASK if ts.relu_masked else 0)
                ws, wsb = workspace(lib.din_conv_workspace_bytes(C.byref(d), 1), dev)
                if oi in x_parked:
                    xi, xg, xld, xcoff = x_parked.pop(oi)
                    xs = L.ConvSrc()
                    xs.dout, xs.wpk_t = xg.data_ptr(), pcache.wpt[xi].data_ptr()
                    xs.cout, xs.ldo, xs.cooff = g.ops[xi].dst.c, xld, xcoff
                    with _timed("dgrad", d, op.name + "+" + g.ops[xi].name):
                        L.check(lib.din_conv_dgrad_x(C.byref(d), _ptr(gout), _ptr(wpt), _ptr(gsrc),
                                                     _ptr(bufs[op.src.tid]) if ts.relu_masked else None, ts.c, op.src.coff, flags,
                                                     C.byref(xs), _ptr(ws), wsb, st), "conv_dgrad_x " + op.name)
                else:
                    with _timed("dgrad", d, op.name):
                        L.check(lib.din_conv_dgrad(C.byref(d), _ptr(gout), _ptr(wpt), _ptr(gsrc),
                                                   _ptr(bufs[op.src.tid]) if ts.relu_masked else None, ts.c, op.src.coff, flags,
                                                   _ptr(ws), wsb, st), "conv_dgrad " + op.name)
        elif src_needs_grad:
            d = _pool_desc(g, op, nb, dt)
            gsrc, acc = grad_target(op.src)
            mask = _ptr(bufs[op.src.tid]) if ts.relu_masked else None
            if op.kind == "maxpool":
                (amax,) = aux[oi]
                L.check(lib.din_maxpool_bwd(C.byref(d), _ptr(bufs[op.src.tid]), _ptr(amax), _ptr(gout), _ptr(gsrc),
                                            int(ts.relu_masked), int(acc), st), "maxpool_bwd")
            elif op.kind == "avgpool":
                L.check(lib.din_avgpool_bwd(C.byref(d), _ptr(gout), _ptr(gsrc), mask, int(acc), st), "avgpool_bwd")
            else:
                L.check(lib.din_bilinear_bwd(C.byref(d), _ptr(gout), _ptr(gsrc), mask, int(acc), st), "bilinear_bwd")
        # the gradient of dst is dead once all its producers ran; producers of one tensor are contiguous in program
        # order for our graphs only per view, so free conservatively when no earlier op writes this tensor
        if not any(o.dst.tid == op.dst.tid for o in g.ops[:oi]):
            gbufs.pop(op.dst.tid, None)
    #  a 1x1's parked data gradient is only consumed when its strided host op reaches its own dgrad branch: a graph whose host
    # takes another path must not drop it silently
    for key in list(wq):
        flush_wq(key)                                  # weight gradients still queued for a grouped launch
    if x_parked:
        raise L.DinError(f"graph_backward: parked 1x1 data gradients were never carried by their strided sibling (ops {sorted(x_parked)}): "
                         "the fused strided + 1x1 dgrad (din_conv_dgrad_x) does not serve this graph shape")
    if side is not None:
        main.wait_stream(side)                        # parameter gradients are complete for whoever runs next on the main stream
    if bn is not None and bn_touched and not bn_train:
        # BatchNorm parameter gradients of every layer in one launch: dgamma = (wdot - dshift * mean) * rstd, dbeta = dshift
        bn_out = torch.empty(2 * bn.total, dtype=torch.float32, device=dev)
        L.check(lib.din_bn_fold_bwd_multi(_ptr(bn.ptrs), _ptr(bn.offs), bn.n, bn.total, BN_EPS, _ptr(bn_wdot), _ptr(bn_dshift),
                                          _ptr(bn_out), _ptr(bn_out[bn.total:]), st), "bn_fold_bwd_multi")
        for oi_, k_ in bn_index.items():
            if grads[offsets[oi_]] is not None:       # the layer took part in this backward
                o0, o1 = bn.off_list[k_], bn.off_list[k_ + 1]
                grads[offsets[oi_] + 1], grads[offsets[oi_] + 2] = bn_out[o0:o1], bn_out[bn.total + o0:bn.total + o1]
    return grads


def _accepts_u8_frames(g: Graph, nb: int, dt: int) -> bool:
    """True when the only reader of the graph input is a conv whose forward and weight-gradient launches take the image-layer kernels
    that read uint8 frames directly (din_conv_accepts_u8): Inception's Conv2d_1a_3x3 at full frame size in bf16."""
    readers = [op for op in g.ops if op.src.tid == g.input_tid]
    if len(readers) != 1 or readers[0].kind != "conv" or readers[0].pooled is not None:
        return False
    key = (nb, dt)
    cache = g.u8_ok
    if key not in cache:
        d = _conv_desc(g, readers[0], nb, dt, g.cin_image)
        d.ldi, d.cioff = 8, 0
        cache[key] = bool(L.load().din_conv_accepts_u8(C.byref(d)))
    return cache[key]


class NHWCGraphFunction(torch.autograd.Function):
    """images (uint8|fp32 NCHW, 0..255) -> output NHWC buffers.  One autograd node for the whole conv stack."""

    @staticmethod
    def forward(ctx, graph: Graph, dt: int, images: torch.Tensor, prenormalised: bool, bn_train: bool, *params):
        lib = L.load()
        require_gpu(images, *params)
        nb, _, h, w = images.shape
        ti = graph.tensors[graph.input_tid]
        assert (h, w) == (ti.h, ti.w), f"image {h}x{w} does not match the graph ({ti.h}x{ti.w})"
        st = _stream()
        if not prenormalised and images.dtype == torch.uint8 and _accepts_u8_frames(graph, nb, dt):
            # the image layer (forward and weight gradient) reads the uint8 frames itself: no prepared NHWC copy of the clip batch
            img = images.contiguous()
        elif prenormalised:
            img = torch.empty((nb, h, w, ti.c), dtype=torch_dtype(dt), device=images.device)
            if ti.c > 3:
                img.zero_()
            L.check(lib.din_nchw_f32_to_nhwc(_ptr(images.float().contiguous()), nb, h, w, 3, _ptr(img), dt, ti.c, 0, st),
                    "nchw_to_nhwc")
        else:
            img = torch.empty((nb, h, w, ti.c), dtype=torch_dtype(dt), device=images.device)
            if images.dtype == torch.uint8:
                L.check(lib.din_prep_images_nhwc(_ptr(images), 1, _ptr(img), dt, nb, h, w, ti.c, st), "prep_nhwc")
            else:
                L.check(lib.din_prep_images_nhwc(_ptr(images.float().contiguous()), 0, _ptr(img), dt, nb, h, w, ti.c, st),
                        "prep_nhwc")
        with torch.no_grad():
            bufs, aux = graph_forward(graph, img, params, dt, save_for_backward=any(p.requires_grad for p in params), bn_train=bn_train)
        ctx.graph, ctx.dt, ctx.bufs, ctx.aux, ctx.bn_train = graph, dt, bufs, aux, bn_train
        ctx.params = params
        ctx.need = [p.requires_grad for p in params]
        outs = tuple(bufs[t] for t in graph.output_tids)
        ctx.set_materialize_grads(False)                   # an unused output must not cost a zero-filled gradient map (0.5 GB for Mixed_6e)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *gouts):
        graph = ctx.graph
        og = {}
        for tid, go in zip(graph.output_tids, gouts):
            if go is not None:
                og[tid] = go.contiguous()
        with torch.no_grad():
            grads = graph_backward(graph, ctx.bufs, ctx.aux, ctx.params, ctx.dt, og, ctx.need, ctx.bn_train)
        ctx.bufs = ctx.aux = None
        grads = [gr if need else None for gr, need in zip(grads, ctx.need)]
        if GRAD_ASSIGN is not None:
            grads = [None if (gr is not None and GRAD_ASSIGN(p_, gr)) else gr for p_, gr in zip(ctx.params, grads)]
        return (None, None, None, None, None, *grads)
