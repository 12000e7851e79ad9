import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from oracle import din_oracle as O
wl = sys.argv[1] if len(sys.argv) > 1 else "inv3"
OH, OW, D = ((87, 157, 1056) if wl == "inv3" else (22, 40, 512))
ocfg = O.OracleCfg(backbone=wl, image_size=(720, 1280), out_size=(OH, OW), emb_features=D)
p = O.synth_params(O.model_param_shapes(ocfg), seed=3, din_std=0.02)
p = {k: v.requires_grad_("running_" not in k) for k, v in p.items()}
images, boxes, labels = O.synth_inputs(1, 3, 12, 720, 1280, OH, OW, 8, seed=0)
images = images.float()
print("cpu_count", os.cpu_count())
for th in [int(a) for a in sys.argv[2:]]:
    torch.set_num_threads(th)
    ts = []
    for it in range(2):
        for v in p.values(): v.grad = None
        t0 = time.time()
        out, inter = O.dynamic_volleyball_forward(ocfg, p, images, boxes, return_intermediates=True)
        t1 = time.time()
        F.cross_entropy(out["activities"], labels).backward()
        ts.append((t1 - t0, time.time() - t1))
    print(wl, "threads", th, "fwd/bwd s:", ["%.2f/%.2f" % t for t in ts], flush=True)
