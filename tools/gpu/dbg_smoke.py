import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from din_amd.config import Config
from din_amd.infer_model import Dynamic_volleyball
from oracle import din_oracle as O
dev = torch.device("cuda:0")
ocfg = O.OracleCfg(image_size=(64, 96), out_size=(2, 3), num_boxes=4, num_frames=3, num_features_boxes=64)
p = O.synth_params(O.model_param_shapes(ocfg), seed=1, din_std=0.05)
images, boxes, labels = O.synth_inputs(2, 3, 4, 64, 96, 2, 3, 8, seed=2)
cfg = Config("volleyball")
cfg.backbone, cfg.image_size, cfg.out_size, cfg.emb_features = "vgg16", (64, 96), (2, 3), 512
cfg.num_boxes, cfg.num_frames, cfg.num_features_boxes, cfg.num_features_gcn = 4, 3, 64, 64
cfg.ST_kernel_size, cfg.sampling_ratio, cfg.beta_factor, cfg.train_backbone = [(3, 3)], [1], False, True
model = Dynamic_volleyball(cfg); model.load_state_dict(p); model = model.to(dev).eval()
ret = model((images.to(dev), boxes.to(dev)))
F.cross_entropy(ret["activities"], labels.to(dev)).backward()
po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
ref = O.dynamic_volleyball_forward(ocfg, po, images.float(), boxes)
F.cross_entropy(ref["activities"], labels).backward()
for k, v in model.named_parameters():
    g, r = v.grad.cpu().double(), po[k].grad.double()
    print(f"{k:40s} maxrel {float((g-r).abs().max()/r.abs().max()):.3e}  l2rel {float((g-r).norm()/r.norm()):.3e}")
