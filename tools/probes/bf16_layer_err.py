import sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from oracle import din_oracle as O
from din_amd.config import Config
from din_amd.infer_model import Dynamic_volleyball
H, W, OH, OW = 192, 320, 6, 10
ocfg = O.OracleCfg(image_size=(H, W), out_size=(OH, OW), num_boxes=6, num_frames=3, num_features_boxes=64)
p = O.synth_params(O.model_param_shapes(ocfg), seed=6, din_std=0.02)
images, boxes, labels = O.synth_inputs(2, 3, 6, H, W, OH, OW, 8, seed=10)
outs = {}
for dt in ("fp32", "bf16"):
    cfg = Config("volleyball")
    cfg.backbone, cfg.image_size, cfg.out_size, cfg.emb_features = "vgg16", (H, W), (OH, OW), 512
    cfg.num_boxes, cfg.num_frames, cfg.num_features_boxes, cfg.num_features_gcn = 6, 3, 64, 64
    cfg.ST_kernel_size, cfg.sampling_ratio, cfg.beta_factor, cfg.train_backbone = [(3, 3)], [1], False, True
    cfg.backbone_dtype = dt
    model = Dynamic_volleyball(cfg); model.load_state_dict(p); model = model.cuda().eval()
    ret = model((images.cuda(), boxes.cuda()))
    F.cross_entropy(ret["activities"], labels.cuda()).backward()
    outs[dt] = (ret["activities"].detach().double().cpu(), {k: v.grad.detach().double().cpu() for k, v in model.named_parameters()})
print("logits", ((outs["bf16"][0]-outs["fp32"][0]).abs().max()/outs["fp32"][0].abs().max()).item())
for k in outs["fp32"][1]:
    a, b = outs["bf16"][1][k], outs["fp32"][1][k]
    l2 = ((a-b).norm()/b.norm()).item(); mx = ((a-b).abs().max()/b.abs().max()).item()
    cos = (a.flatten() @ b.flatten() / (a.norm()*b.norm())).item()
    print(f"{k:40s} relL2 {l2:.3e} relMax {mx:.3e} cos {cos:.5f} |ref| {b.norm().item():.3e}")
