"""Dynamic_volleyball / Dynamic_collective -- drop-in for the reference's infer_model.py:15-234 and :1135-1319.

Same constructor (`Model(cfg)`), same `forward((images, boxes[, bboxes_num])) -> {'activities': [B, A]}`, same
`loadmodel(path)` and state_dict keys (backbone.*, fc_emb_1, nl_emb_1, point_conv, point_ln, DPI.*, dpi_nl,
fc_activities), so stage-1 / stage-2 checkpoints of the reference load unchanged.  Every arithmetic step runs in
hand-written gfx950 kernels behind the C ABI; see DESIGN.md for the kernel map.

Deliberate differences from the reference (all documented in DESIGN.md):
  * `cfg.backbone == 'inv3'` works (the reference has no head branch for it, infer_model.py:203-216 -> crash);
    it uses the vgg16 residual -> LN -> ReLU -> dropout order.
  * no torch.cuda.empty_cache() per step (infer_model.py:200 is a perf bug), images may be uint8.
  * new optional cfg field `backbone_dtype` ('fp32' parity mode | 'bf16' throughput mode), default 'fp32'.
"""
from __future__ import annotations

import collections

import torch
import torch.nn as nn

from . import ops
from .backbone.backbone import MyInception_v3, MyVGG16
from .infer_module.dynamic_infer_module import (Dynamic_Person_Inference, Hierarchical_Dynamic_Inference,
                                                Multi_Dynamic_Inference)
from .infer_module.positional_encoding import Context_PositionEmbeddingSine
from .infer_module.TCE_STBiP_module import MultiHeadLayerEmbfeatureContextEncoding
from .roi_align.roi_align import RoIAlign
from .utils import print_log


def _as_kernel_list(k):
    if isinstance(k, (list,)) and len(k) and isinstance(k[0], (tuple, list)):
        return [tuple(x) for x in k]
    if isinstance(k, int):
        return [(k, k)]
    return [tuple(k)]


class _DynamicBase(nn.Module):
    def _build_trunk(self, cfg):
        D, K, NFB = cfg.emb_features, cfg.crop_size[0], cfg.num_features_boxes
        dt = getattr(cfg, "backbone_dtype", "fp32")
        if cfg.backbone == "inv3":
            self.backbone = MyInception_v3(transform_input=False, pretrained=True, compute_dtype=dt)
        elif cfg.backbone == "vgg16":
            self.backbone = MyVGG16(pretrained=True, compute_dtype=dt)
        else:
            raise NotImplementedError(f"backbone {cfg.backbone!r}: the MI355X hot path covers 'vgg16' and 'inv3' "
                                      f"(BASELINE.json configs); vgg19/res18/alex are out of scope")
        if not cfg.train_backbone:
            for p in self.backbone.parameters():
                p.requires_grad = False
        self._step = 0                  # dropout-mask counter (ops.mask_seed); train_net saves / restores it with the checkpoint
        self.roi_align = RoIAlign(*cfg.crop_size)
        self.fc_emb_1 = nn.Linear(K * K * D, NFB)
        self.nl_emb_1 = nn.LayerNorm([NFB])

    def _init_linears(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def loadmodel(self, filepath):
        state = torch.load(filepath, map_location="cpu")
        self.backbone.load_state_dict(state["backbone_state_dict"])
        self.fc_emb_1.load_state_dict(state["fc_emb_state_dict"])
        print("Load model states from: ", filepath)

    def loadpart(self, pretrained_state_dict, model, prefix):
        """reference infer_model.py:358-368 (Dynamic_TCE_volleyball.loadpart): copy every entry of `pretrained_state_dict` whose key, with
        `prefix` removed, exists in `model` -- partial initialisation from another checkpoint"""
        num = 0
        model_state_dict = model.state_dict()
        picked = collections.OrderedDict()
        for k, v in pretrained_state_dict.items():
            if k.replace(prefix, "") in model_state_dict:
                picked[k.replace(prefix, "")] = v
                num += 1
        model_state_dict.update(picked)
        model.load_state_dict(model_state_dict)
        print(str(num) + " parameters loaded for " + prefix)

    # ---- shared front: images -> per-box embeddings [B,T,N,NFB] ------------------------------------------
    def _embed(self, images_in, boxes_in, N, return_context: bool = False):
        cfg = self.cfg
        B, T = images_in.shape[0], images_in.shape[1]
        H, W = cfg.image_size
        OH, OW = cfg.out_size
        D, K = cfg.emb_features, cfg.crop_size[0]
        images_flat = images_in.reshape(B * T, 3, H, W)
        boxes_flat = boxes_in.reshape(B * T * N, 4)
        boxes_idx = ops.boxes_frame_index(B * T, N, boxes_in.device)                  # infer_model.py:155-157
        bufs, graph = self.backbone.forward_nhwc(images_flat)                        # prep fused (:161-162)
        fm = bufs[0]
        assert tuple(fm.shape[1:3]) == (OH, OW), f"backbone output {tuple(fm.shape[1:3])} != cfg.out_size {(OH, OW)}"   # (:164)
        views = self.backbone.output_views(graph)
        if fm.shape[3] >= D:                                                         # one map holds all D channels (VGG; materialised fuse)
            tid = graph.output_tids[0]
            crops = self.roi_align(fm, boxes_flat, boxes_idx, nhwc=True, channels=D,
                                   relu_masked=graph.tensors[tid].relu_masked)      # [BTN, D, K, K]  (:178-180)
        else:
            # multi-scale fuse + RoIAlign in one step (:165-180): every backbone output is sampled through its virtual align_corners
            # resize to (OH, OW); the resized / concatenated map is never built, forward or backward
            assert sum(c for _, _, c in views) == D, f"backbone outputs hold {sum(c for _, _, c in views)} channels, cfg.emb_features = {D}"
            maps = [(b, c, graph.tensors[tid].relu_masked) for b, (tid, _coff, c) in zip(bufs, views)]
            crops = self.roi_align.forward_multiscale(maps, boxes_flat, boxes_idx, (OH, OW))
        feats = crops.reshape(B, T, N, D * K * K)
        x = ops.linear(feats, self.fc_emb_1.weight, self.fc_emb_1.bias,
                       lowp=getattr(self.cfg, "backbone_dtype", "fp32") == "bf16")              # :184
        x = ops.layer_norm(x, self.nl_emb_1.weight, self.nl_emb_1.bias, relu=True)   # :185-186
        if return_context:                                                           # the LAST backbone output, pixel-major (:404)
            tid_last, coff, c = views[-1]
            assert coff == 0 and bufs[-1].shape[3] == c
            return x, bufs[-1], graph.tensors[tid_last].relu_masked
        return x

    def _dropout_seed(self):
        self._step += 1
        return ops.mask_seed(int(getattr(self.cfg, "train_random_seed", 0)), self._step)


class Dynamic_volleyball(_DynamicBase):
    """main module of DIN for the volleyball dataset (reference infer_model.py:15-234)"""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        T, N = cfg.num_frames, cfg.num_boxes
        NFB = cfg.num_features_boxes
        self._build_trunk(cfg)
        in_dim = cfg.lite_dim if cfg.lite_dim else NFB
        print_log(getattr(cfg, "log_path", None), ("Activate" if cfg.lite_dim else "Deactivate") + " lite model inference.")
        kernels = _as_kernel_list(cfg.ST_kernel_size)
        if not cfg.hierarchical_inference:
            self.DPI = Multi_Dynamic_Inference(in_dim=in_dim, person_mat_shape=(10, 12), stride=cfg.stride,
                                               kernel_size=kernels, dynamic_sampling=cfg.dynamic_sampling,
                                               sampling_ratio=cfg.sampling_ratio, group=cfg.group,
                                               scale_factor=cfg.scale_factor, beta_factor=cfg.beta_factor,
                                               parallel_inference=cfg.parallel_inference, num_DIM=cfg.num_DIM, cfg=cfg)
        else:
            self.DPI = Hierarchical_Dynamic_Inference(in_dim=in_dim, person_mat_shape=(T, N), stride=cfg.stride,
                                                      kernel_size=kernels, dynamic_sampling=cfg.dynamic_sampling,
                                                      sampling_ratio=cfg.sampling_ratio, group=cfg.group,
                                                      scale_factor=cfg.scale_factor, beta_factor=cfg.beta_factor,
                                                      parallel_inference=cfg.parallel_inference, cfg=cfg)
        print_log(getattr(cfg, "log_path", None), "Hierarchical Inference : " + str(cfg.hierarchical_inference))
        self.dpi_nl = nn.LayerNorm([T, N, in_dim])
        self.dropout_global = nn.Dropout(p=cfg.train_dropout_prob)      # holder of p; the mask is fused in the LN kernel
        if cfg.lite_dim:
            self.point_conv = nn.Conv2d(NFB, in_dim, kernel_size=1, stride=1)
            self.point_ln = nn.LayerNorm([T, N, in_dim])
            self.fc_activities = nn.Linear(in_dim, cfg.num_activities)
        else:
            self.fc_activities = nn.Linear(cfg.num_features_gcn, cfg.num_activities)
        self._init_linears()

    def forward(self, batch_data):
        images_in, boxes_in = batch_data
        cfg = self.cfg
        B, T, N = images_in.shape[0], images_in.shape[1], cfg.num_boxes
        x = self._embed(images_in, boxes_in, N)                                       # [B,T,N,NFB]
        if cfg.lite_dim:                                                              # :188-193
            x = ops.GridConvFunction.apply(x, self.point_conv.weight, self.point_conv.bias, 1)
            x = ops.layer_norm(x, self.point_ln.weight, self.point_ln.bias, relu=True)
        graph, _mad = self.DPI(x)                                                     # :199
        p = cfg.train_dropout_prob if self.training else 0.0
        head_mode = "res18" if cfg.backbone == "res18" else "vgg16"
        if head_mode == "vgg16":                                                      # :210-216 (also used for inv3)
            s = ops.layer_norm(graph, self.dpi_nl.weight, self.dpi_nl.bias, res=x, relu=True, drop_p=p,
                               seed=self._dropout_seed())
        else:                                                                         # :203-209
            raise NotImplementedError
        scores = ops.HeadFunction.apply(s, self.fc_activities.weight, self.fc_activities.bias, None)   # :224-232
        return {"activities": scores}


class Dynamic_TCE_volleyball(_DynamicBase):
    """DIN behind a context-encoding transformer (reference infer_model.py:237-468): every box embedding attends over
    the pixels of its frame's last backbone map (+ sine position embedding) with 4 heads of 128 features; [embedding | context
    encoding] (NFB + 512 channels) then goes through the same DIN modules and head as Dynamic_volleyball.

    As in the reference, only the vgg16 / res18 head branches exist (:430-442; 'inv3' leaves `boxes_states` unbound there) and this
    package has no res18 trunk, so cfg.backbone must be 'vgg16'; and lite_dim cannot be combined with it (the reference sizes
    fc_activities for lite_dim but feeds it lite_dim + 512 channels, :341-343 vs :410-456)."""

    NUM_HEADS_CONTEXT, NUM_FEATURES_CONTEXT = 4, 128                                   # :244-245

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if cfg.backbone != "vgg16":
            raise NotImplementedError("Dynamic_TCE_volleyball: the reference's forward only handles the vgg16 / res18 trunks "
                                      "(infer_model.py:430-442) and its context transformer expects 512 context channels")
        if cfg.lite_dim:
            raise NotImplementedError("Dynamic_TCE_volleyball with lite_dim: the reference model cannot run (fc_activities is sized for "
                                      "lite_dim, its input has lite_dim + 512 channels)")
        T, N = cfg.num_frames, cfg.num_boxes
        NFB, K = cfg.num_features_boxes, cfg.crop_size[0]
        self._build_trunk(cfg)
        print_log(getattr(cfg, "log_path", None), "Deactivate lite model inference.")
        self.multilayer_head_embfeature_context_encoding = MultiHeadLayerEmbfeatureContextEncoding(
            self.NUM_HEADS_CONTEXT, 1, self.NUM_FEATURES_CONTEXT, NFB, K, N, context_dropout_ratio=0.1)       # :287-292
        self.context_positionembedding1 = Context_PositionEmbeddingSine(16, 512 / 2)                          # :293
        context_dim = NFB + self.NUM_HEADS_CONTEXT * self.NUM_FEATURES_CONTEXT                                # :296
        kernels = _as_kernel_list(cfg.ST_kernel_size)
        if not cfg.hierarchical_inference:
            self.DPI = Multi_Dynamic_Inference(in_dim=context_dim, person_mat_shape=(10, 12), stride=cfg.stride,
                                               kernel_size=kernels, dynamic_sampling=cfg.dynamic_sampling,
                                               sampling_ratio=cfg.sampling_ratio, group=cfg.group,
                                               scale_factor=cfg.scale_factor, beta_factor=cfg.beta_factor,
                                               parallel_inference=cfg.parallel_inference, num_DIM=cfg.num_DIM, cfg=cfg)
        else:
            self.DPI = Hierarchical_Dynamic_Inference(in_dim=context_dim, person_mat_shape=(T, N), stride=cfg.stride,
                                                      kernel_size=kernels, dynamic_sampling=cfg.dynamic_sampling,
                                                      sampling_ratio=cfg.sampling_ratio, group=cfg.group,
                                                      scale_factor=cfg.scale_factor, beta_factor=cfg.beta_factor,
                                                      parallel_inference=cfg.parallel_inference, cfg=cfg)
        print_log(getattr(cfg, "log_path", None), "Hierarchical Inference : " + str(cfg.hierarchical_inference))
        self.dpi_nl = nn.LayerNorm([T, N, context_dim])                                                       # :334
        self.dropout_global = nn.Dropout(p=cfg.train_dropout_prob)
        self.fc_activities = nn.Linear(context_dim, cfg.num_activities)                                       # :343
        self._init_linears()

    def forward(self, batch_data):
        images_in, boxes_in = batch_data
        cfg = self.cfg
        B, T, N = images_in.shape[0], images_in.shape[1], cfg.num_boxes
        x, context, masked = self._embed(images_in, boxes_in, N, return_context=True)       # [B,T,N,NFB], NHWC [BT,OH,OW,512]
        context = self.context_positionembedding1(context, nhwc=True, relu_masked=masked)    # :404-406 (fp32)
        enc = self.multilayer_head_embfeature_context_encoding
        enc.seed_base = int(getattr(cfg, "train_random_seed", 0)) + 101
        states = enc(x.reshape(B * T * N, -1), context, nhwc=True)                           # :408
        xc = torch.cat((x, states.reshape(B, T, N, -1)), dim=3)                              # :409-410
        graph, _mad = self.DPI(xc)                                                           # :414
        p = cfg.train_dropout_prob if self.training else 0.0
        s = ops.layer_norm(graph, self.dpi_nl.weight, self.dpi_nl.bias, res=xc, relu=True, drop_p=p,
                           seed=self._dropout_seed())                                        # vgg16 branch :436-442
        scores = ops.HeadFunction.apply(s, self.fc_activities.weight, self.fc_activities.bias, None)   # :452-466
        return {"activities": scores}


class Dynamic_collective(_DynamicBase):
    """DIN for the Collective Activity dataset (reference infer_model.py:1135-1319), variable actors per clip.

    The reference crashes here (tuple/tensor mismatch, SURVEY section 0 bug 3); this implements the intended dataflow:
    per clip b with N_b = bboxes_num[b,0] valid boxes: DIN on [1,T,N_b,C] -> +x -> LayerNorm([T,C]) per actor -> ReLU ->
    dropout -> max over actors -> fc -> mean over T -- as ONE launch set per batch: the DIN walk, and the head take the per-clip
    actor counts as a device array (`n_per_clip`), padding actors are zeroed once (din_mask_actors)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        T, N = cfg.num_frames, cfg.num_boxes
        NFB = cfg.num_features_boxes
        self._build_trunk(cfg)
        in_dim = cfg.lite_dim if cfg.lite_dim else NFB
        k = _as_kernel_list(cfg.ST_kernel_size)[0]
        self.DPI = Dynamic_Person_Inference(in_dim=in_dim, person_mat_shape=(10, 12), stride=cfg.stride, kernel_size=k,
                                            dynamic_sampling=cfg.dynamic_sampling, sampling_ratio=cfg.sampling_ratio,
                                            group=cfg.group, scale_factor=cfg.scale_factor, beta_factor=cfg.beta_factor,
                                            parallel_inference=cfg.parallel_inference, cfg=cfg)
        self.dpi_nl = nn.LayerNorm([T, in_dim])
        self.dropout_global = nn.Dropout(p=cfg.train_dropout_prob)
        if cfg.lite_dim:
            self.point_conv = nn.Conv2d(NFB, in_dim, kernel_size=1, stride=1)
            self.point_ln = nn.LayerNorm([T, N, in_dim])
            self.fc_activities = nn.Linear(in_dim, cfg.num_activities)
        else:
            self.fc_activities = nn.Linear(cfg.num_features_gcn, cfg.num_activities)
        self._init_linears()

    def forward(self, batch_data):
        images_in, boxes_in, bboxes_num_in = batch_data
        cfg = self.cfg
        B, T, MAX_N = images_in.shape[0], images_in.shape[1], cfg.num_boxes
        x = self._embed(images_in, boxes_in, MAX_N)                                   # [B,T,MAX_N,NFB]
        if cfg.lite_dim:
            x = ops.GridConvFunction.apply(x, self.point_conv.weight, self.point_conv.bias, 1)
            x = ops.layer_norm(x, self.point_ln.weight, self.point_ln.bias, relu=True)
        # Variable actors per clip WITHOUT a host loop or a device->host read (the reference loops over clips and slices
        # boxes_features_all[b, :, :N], infer_model.py:1284-1316): the per-clip counts stay on the device and the kernels take them.
        n_per_clip = bboxes_num_in.reshape(B, T)[:, 0].to(torch.int32).clamp(1, MAX_N).contiguous()
        xm = ops.MaskActorsFunction.apply(x, n_per_clip)                              # padding actors -> 0 (== zero padding of the grid)
        g, _ = self.DPI(xm, n_per_clip)                                               # clip b: DIN on its T x n_b grid (:1291)
        p = cfg.train_dropout_prob if self.training else 0.0
        # (g + x) -> [B, N, T, C]: LayerNorm([T, C]) per actor (:1295-1297), ReLU, dropout fused; padding actors' rows are never read
        s = ops.AxpbyFunction.apply(g, xm, 1.0, 1.0).permute(0, 2, 1, 3).contiguous()
        s = ops.layer_norm(s, self.dpi_nl.weight, self.dpi_nl.bias, relu=True, drop_p=p, seed=self._dropout_seed())
        s = s.permute(0, 2, 1, 3).contiguous()                                        # [B, T, N, C]
        # max over the clip's n_b actors, fc, mean over T (:1306-1309)
        scores = ops.HeadFunction.apply(s, self.fc_activities.weight, self.fc_activities.bias, n_per_clip)
        return {"activities": scores}
