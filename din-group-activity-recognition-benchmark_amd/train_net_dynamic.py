"""Stage-2 trainer -- drop-in for the reference's train_net_dynamic.py (train_net :27-157, train_volleyball :159-235,
test_volleyball :238-312, train_collective :315-389, test_collective :392-471) on MI355X: one process per GPU (torchrun), clips
sharded over ranks, gradients all-reduced by RCCL (din_amd.parallel) instead of nn.DataParallel, fused Adam.  The info-dict keys of
the reference are kept; train / test loops are picked by cfg.dataset_name as in the reference (:106-109).

Checkpoints follow the reference: `cfg.load_backbone_stage2` loads the stage-1 backbone + embedding (`model.loadmodel`, :80-81),
`cfg.load_stage2model` resumes a stage-2 `{'epoch', 'state_dict', 'optimizer'}` file (:82-88; a `module.` prefix left by nn.DataParallel
is stripped), and the file written after every test epoch has the same three keys with torch.optim.Adam's optimizer format.
BatchNorm follows `cfg.set_bn_eval` exactly like :98-100 / :170-172: model.train(), then BatchNorm modules back to eval() when the flag is
set (train_collective does that unconditionally, :323-324).

Datasets are out of scope: `train_net(cfg, training_set, validation_set)` takes any torch Dataset yielding the
reference's tuples (volleyball: images [T,3,H,W] 0..255, boxes [T,N,4] feature px, actions [T,N], activities [T]; collective: the same
with MAX_N boxes plus bboxes_num [T]); with none given it trains on a synthetic set of that shape.
"""
from __future__ import annotations

import os

import random
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as data

from . import parallel
from .input_feed import DeviceFeed
from .infer_model import Dynamic_collective, Dynamic_TCE_volleyball, Dynamic_volleyball
from .optim import FusedAdam
from .utils import AverageMeter, Timer, print_log


def set_bn_eval(m):
    """reference train_net_dynamic.py:17-20"""
    if m.__class__.__name__.find("BatchNorm") != -1:
        m.eval()


def adjust_lr(optimizer, new_lr):
    print("change learning rate:", new_lr)
    for group in optimizer.param_groups:
        group["lr"] = new_lr


class SyntheticVolleyball(data.Dataset):
    """Volleyball-shaped random clips: uint8 images, player-shaped boxes in feature px, labels."""

    def __init__(self, cfg, length=8, seed=0):
        self.cfg, self.length, self.seed = cfg, length, seed

    def __len__(self):
        return self.length

    def _clip(self, i):
        cfg = self.cfg
        T, N = cfg.num_frames, cfg.num_boxes
        H, W = cfg.image_size
        OH, OW = cfg.out_size
        r = np.random.default_rng(self.seed * 100003 + i)
        images = torch.from_numpy(r.integers(0, 256, size=(T, 3, H, W), dtype=np.uint8))
        cx, cy = r.uniform(0.05, 0.95, (T, N)) * OW, r.uniform(0.3, 0.9, (T, N)) * OH
        bw, bh = r.uniform(0.03, 0.08, (T, N)) * OW, r.uniform(0.15, 0.35, (T, N)) * OH
        boxes = np.stack([np.clip(cx - bw / 2, 0, OW), np.clip(cy - bh / 2, 0, OH), np.clip(cx + bw / 2, 0, OW),
                          np.clip(cy + bh / 2, 0, OH)], -1).astype(np.float32)
        actions = torch.from_numpy(r.integers(0, cfg.num_actions, size=(T, N)).astype(np.int64))
        activities = torch.full((T,), int(r.integers(0, cfg.num_activities)), dtype=torch.int64)
        return r, images, torch.from_numpy(boxes), actions, activities

    def __getitem__(self, i):
        return self._clip(i)[1:]


class SyntheticCollective(SyntheticVolleyball):
    """Collective-Activity-shaped clips (reference collective.py:150-215): cfg.num_boxes is MAX_N, each clip has 1..MAX_N actors
    (the same count in all of its frames), padding boxes are zero, plus the `bboxes_num [T]` entry."""

    def __getitem__(self, i):
        r, images, boxes, actions, activities = self._clip(i)
        T, N = self.cfg.num_frames, self.cfg.num_boxes
        n = int(r.integers(1, N + 1))
        boxes[:, n:] = 0.0
        actions[:, n:] = -1
        return images, boxes, actions, activities, torch.full((T,), n, dtype=torch.int32)


def build_model(cfg):
    registry = {"dynamic_volleyball": Dynamic_volleyball, "dynamic_tce_volleyball": Dynamic_TCE_volleyball,
                "dynamic_collective": Dynamic_collective}                            # reference train_net_dynamic.py:66-73
    if cfg.inference_module_name not in registry:
        raise NotImplementedError(f"{cfg.inference_module_name}: only the DIN models are on the MI355X hot path")
    return registry[cfg.inference_module_name](cfg)


class _Epoch:
    """meters of one pass: loss / accuracy / confusion matrix, accumulated on the device and read ONCE at the end of the pass (the
    reference reads .item() every batch, a host sync per step)"""

    def __init__(self, cfg, device):
        self.conf = torch.zeros(cfg.num_activities, cfg.num_activities, dtype=torch.int64, device=device)
        self.loss_sum = torch.zeros((), dtype=torch.float64, device=device)
        self.clips = 0
        self.timer = Timer()

    def add(self, scores, target, loss):
        pred = torch.argmax(scores, dim=1)
        self.conf.index_put_((target.long(), pred), torch.ones_like(pred, dtype=torch.int64), accumulate=True)
        self.loss_sum += loss.detach().double() * scores.shape[0]
        self.clips += scores.shape[0]

    def info(self, epoch):
        conf = self.conf.cpu().float()
        total = max(self.clips, 1)
        per_class = conf.diag() / conf.sum(1).clamp(min=1)
        return {"time": self.timer.timeit(), "epoch": epoch, "loss": float(self.loss_sum.item()) / total,
                "activities_acc": float(conf.diag().sum()) / total * 100, "activities_conf": conf.numpy(),
                "activities_MPCA": float(per_class.mean() * 100)}


def _train_pass(data_loader, model, device, optimizer, epoch, cfg, grad_buckets, collective, max_batches=None):
    meters = _Epoch(cfg, device)
    for bi, batch_data in enumerate(DeviceFeed(data_loader, device)):      # batch k+1 crosses PCIe on the copy stream during step k
        if max_batches is not None and bi >= max_batches:
            break
        model.train()
        if cfg.set_bn_eval or collective:                              # reference :170-172 / :323-324
            model.apply(set_bn_eval)
        batch_size, num_frames = batch_data[0].shape[0], batch_data[0].shape[1]
        activities_in = batch_data[3].reshape((batch_size, num_frames))[:, 0].reshape((batch_size,))
        inputs = (batch_data[0], batch_data[1], batch_data[4]) if collective else (batch_data[0], batch_data[1])
        ret = model(inputs)
        activities_scores = ret["activities"]
        total_loss = F.cross_entropy(activities_scores, activities_in)
        optimizer.zero_grad()
        total_loss.backward()
        if grad_buckets is not None:
            grad_buckets.allreduce(scale_in_optimizer=True)        # the 1 / world factor rides in the fused Adam kernel
            optimizer.step(grad_scale=grad_buckets.grad_scale)
        else:
            optimizer.step()
        meters.add(activities_scores.detach(), activities_in, total_loss)
    return meters.info(epoch)


def _test_pass(data_loader, model, device, epoch, cfg, collective):
    model.eval()
    meters = _Epoch(cfg, device)
    with torch.no_grad():
        for batch_data in DeviceFeed(data_loader, device):
            batch_size, num_frames = batch_data[0].shape[0], batch_data[0].shape[1]
            activities_in = batch_data[3].reshape((batch_size, num_frames))[:, 0].reshape((batch_size,))
            inputs = (batch_data[0], batch_data[1], batch_data[4]) if collective else (batch_data[0], batch_data[1])
            scores = model(inputs)["activities"]
            meters.add(scores, activities_in, F.cross_entropy(scores, activities_in))
    return meters.info(epoch)


def train_volleyball(data_loader, model, device, optimizer, epoch, cfg, grad_buckets=None, max_batches=None):
    return _train_pass(data_loader, model, device, optimizer, epoch, cfg, grad_buckets, False, max_batches)


def test_volleyball(data_loader, model, device, epoch, cfg):
    return _test_pass(data_loader, model, device, epoch, cfg, False)


def train_collective(data_loader, model, device, optimizer, epoch, cfg, grad_buckets=None, max_batches=None):
    """reference :315-389 with `ret['activities']` (the reference hands the model's dict to cross_entropy, SURVEY 0 bug 3)"""
    return _train_pass(data_loader, model, device, optimizer, epoch, cfg, grad_buckets, True, max_batches)


def test_collective(data_loader, model, device, epoch, cfg):
    return _test_pass(data_loader, model, device, epoch, cfg, True)


def load_stage2_state(model, path):
    """reference :82-88: `{'epoch', 'state_dict', 'optimizer'}`; keys saved through nn.DataParallel carry a 'module.' prefix"""
    state = torch.load(path, map_location="cpu", weights_only=False)
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state["state_dict"].items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing if "num_batches_tracked" not in k]
    if bad or unexpected:
        raise RuntimeError(f"stage-2 checkpoint {path} does not match the model: missing {bad[:5]}, unexpected {list(unexpected)[:5]}")
    return state


def _mask_counters(model):
    return [(n, m) for n, m in model.named_modules() if hasattr(m, "_step") and isinstance(getattr(m, "_step"), int)]


def train_net(cfg, training_set=None, validation_set=None, max_steps=None):
    """Reference train_net (:27-157).  Launch one process per GPU with torchrun; a single process works too.  max_steps bounds the number
    of epochs (smoke runs)."""
    rank, local_rank, world = parallel.init_from_env()
    np.random.seed(cfg.train_random_seed)
    torch.manual_seed(cfg.train_random_seed)
    random.seed(cfg.train_random_seed)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    collective = cfg.dataset_name == "collective"
    synth = SyntheticCollective if collective else SyntheticVolleyball
    if training_set is None and validation_set is None and getattr(cfg, "data_path", None) and os.path.isdir(cfg.data_path):
        # the real datasets (reference train_net_dynamic.py:57-58 -> dataset.return_dataset): annotation trees + JPEG frames -> uint8 clips,
        # feature-px boxes, padded tracks (din_amd/volleyball.py, collective.py); without a data tree the synthetic clips stand in
        from .dataset import return_dataset
        training_set, validation_set = return_dataset(cfg)
    training_set = training_set or synth(cfg, length=max(cfg.batch_size * 2, 4))
    validation_set = validation_set or synth(cfg, length=max(cfg.test_batch_size, 2), seed=1)
    if cfg.batch_size % world != 0:
        raise ValueError(f"batch_size {cfg.batch_size} must be divisible by the number of ranks {world}: the gradient all-reduce averages "
                         f"per-rank means, which is the global mean only for equal shards")
    per_rank = cfg.batch_size // world
    sampler = data.distributed.DistributedSampler(training_set, world, rank, shuffle=True) if world > 1 else None
    training_loader = data.DataLoader(training_set, batch_size=per_rank, shuffle=sampler is None, sampler=sampler, num_workers=0)
    validation_loader = data.DataLoader(validation_set, batch_size=cfg.test_batch_size, shuffle=False, num_workers=0)
    log_path = getattr(cfg, "log_path", None)
    if cfg.training_stage != 2:
        raise NotImplementedError("training_stage 1 (Basenet) is out of scope: the MI355X hot path is the stage-2 DIN step")
    model = build_model(cfg)
    resume = None
    if cfg.load_backbone_stage2:                                       # :80-81
        model.loadmodel(cfg.stage1_model_path)
    elif cfg.load_stage2model:                                         # :82-88
        resume = load_stage2_state(model, cfg.stage2model)
        print_log(log_path, "Loading stage2 model: " + str(cfg.stage2model))
    else:
        print_log(log_path, "Not loading stage1 or stage2 model.")
    model = model.to(device)
    parallel.broadcast_parameters(model)
    model.train()
    if cfg.set_bn_eval:
        model.apply(set_bn_eval)
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = FusedAdam(params, lr=cfg.train_learning_rate, weight_decay=cfg.weight_decay)
    start_epoch = 1
    if resume is not None and getattr(cfg, "resume_optimizer", True) and "optimizer" in resume:
        # (the reference reloads only the weights; resuming the moments, the epoch and the dropout-mask counters too is the extension
        #  `cfg.resume_optimizer`, default on)
        try:
            optimizer.load_state_dict(resume["optimizer"])
            start_epoch = int(resume.get("epoch", 0)) + 1
            for name, m in _mask_counters(model):
                if name in resume.get("mask_steps", {}):
                    m._step = int(resume["mask_steps"][name])
        except (ValueError, KeyError) as e:
            print_log(log_path, f"optimizer state of {cfg.stage2model} not resumed: {e}")
    buckets = parallel.GradBuckets(params) if world > 1 else None
    train = train_collective if collective else train_volleyball
    test = test_collective if collective else test_volleyball
    if cfg.test_before_train:
        print(test(validation_loader, model, device, 0, cfg))
    infos = []
    best_result = {"epoch": 0, "activities_acc": 0}
    for epoch in range(start_epoch, start_epoch + cfg.max_epoch):
        if epoch in cfg.lr_plan:
            adjust_lr(optimizer, cfg.lr_plan[epoch])
        if sampler is not None:
            sampler.set_epoch(epoch)
        info = train(training_loader, model, device, optimizer, epoch, cfg, buckets)
        if rank == 0:
            print_log(log_path, "Train epoch %d: loss %.5f acc %.2f%%" % (epoch, info["loss"], info["activities_acc"]))
        if epoch % cfg.test_interval_epoch == 0:
            tinfo = test(validation_loader, model, device, epoch, cfg)
            if tinfo["activities_acc"] > best_result["activities_acc"]:
                best_result = tinfo
            if rank == 0:
                print_log(log_path, "Test epoch %d: loss %.5f acc %.2f%%" % (epoch, tinfo["loss"], tinfo["activities_acc"]))
                print_log(log_path, "Best group activity accuracy: %.2f%% at epoch #%d." % (best_result["activities_acc"], best_result["epoch"]))
                result_path = getattr(cfg, "result_path", None)
                if result_path:                                        # :141-147
                    state = {"epoch": epoch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
                             "mask_steps": {name: m._step for name, m in _mask_counters(model)}}
                    torch.save(state, result_path + "/stage%d_epoch%d_%.2f%%.pth" % (cfg.training_stage, epoch, tinfo["activities_acc"]))
            info = dict(train=info, test=tinfo)
        infos.append(info)
        if max_steps is not None and len(infos) >= max_steps:
            break
    return infos
