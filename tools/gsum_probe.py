#!/usr/bin/env python3
"""gsum deviations of one golden model case (diagnostic): prints the worst keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.test_gpu_din_model import load_model_case
from din_amd.config import Config
from din_amd.infer_model import Dynamic_volleyball
path = sys.argv[1]
z, ocfg, p, images, boxes, labels = load_model_case(path)
cfg = Config("volleyball")
cfg.backbone, cfg.image_size, cfg.out_size, cfg.emb_features = ocfg.backbone, ocfg.image_size, ocfg.out_size, ocfg.emb_features
cfg.num_boxes, cfg.num_frames = ocfg.num_boxes, ocfg.num_frames
cfg.num_features_boxes = cfg.num_features_gcn = ocfg.num_features_boxes
cfg.ST_kernel_size, cfg.sampling_ratio, cfg.num_DIM = ocfg.ST_kernel_size, ocfg.sampling_ratio, ocfg.num_DIM
cfg.beta_factor, cfg.lite_dim, cfg.hierarchical_inference = ocfg.beta_factor, ocfg.lite_dim, ocfg.hierarchical_inference
cfg.train_backbone, cfg.backbone_dtype = True, "fp32"
cfg.hier_dropout_p = 0.0
model = Dynamic_volleyball(cfg)
model.load_state_dict(p, strict=False)
model = model.cuda().eval()
ret = model((images.cuda(), boxes.cuda()))
F.cross_entropy(ret["activities"], labels.cuda()).backward()
named = dict(model.named_parameters())
rows = []
for k in z.files:
    if k.startswith("gsum."):
        name = k[5:]
        got = named[name].grad.double().sum().item()
        rows.append((abs(got - float(z[k])) / (float(z["gabs." + name]) + 1e-30), name, got, float(z[k]), float(z["gabs." + name])))
rows.sort(reverse=True)
for r in rows[:8]:
    print("%.2e %s got %.6g want %.6g gabs %.4g" % r)
