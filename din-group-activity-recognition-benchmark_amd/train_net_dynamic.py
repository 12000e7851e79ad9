"""Stage-2 trainer -- drop-in for the reference's train_net_dynamic.py (train_net :27-157, train_volleyball :159-235,
test_volleyball :238-312) on MI355X: one process per GPU (torchrun), clips sharded over ranks, gradients all-reduced by
RCCL (din_amd.parallel) instead of nn.DataParallel, fused Adam.  The info-dict keys of the reference are kept.

Datasets are out of scope: `train_net(cfg, training_set, validation_set)` takes any torch Dataset
yielding the reference's tuples (images [T,3,H,W] 0..255, boxes [T,N,4] feature px, actions [T,N], activities [T]);
with none given it trains on a synthetic set of that shape.
"""
from __future__ import annotations

import random
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as data

from . import parallel
from .infer_model import Dynamic_collective, Dynamic_volleyball
from .optim import FusedAdam
from .utils import AverageMeter, Timer, print_log


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

    def __getitem__(self, i):
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
        return images, torch.from_numpy(boxes), actions, activities


def build_model(cfg):
    registry = {"dynamic_volleyball": Dynamic_volleyball, "dynamic_collective": Dynamic_collective}
    if cfg.inference_module_name not in registry:
        raise NotImplementedError(f"{cfg.inference_module_name}: only the DIN models are on the MI355X hot path")
    return registry[cfg.inference_module_name](cfg)


def train_volleyball(data_loader, model, device, optimizer, epoch, cfg, grad_buckets=None):
    activities_meter, loss_meter, epoch_timer = AverageMeter(), AverageMeter(), Timer()
    conf = torch.zeros(cfg.num_activities, cfg.num_activities, dtype=torch.int64)
    for batch_data in data_loader:
        model.train()
        batch_data = [b.to(device=device, non_blocking=True) for b in batch_data]
        batch_size, num_frames = batch_data[0].shape[0], batch_data[0].shape[1]
        activities_in = batch_data[3].reshape((batch_size, num_frames))[:, 0].reshape((batch_size,))
        ret = model((batch_data[0], batch_data[1]))
        activities_scores = ret["activities"]
        total_loss = F.cross_entropy(activities_scores, activities_in)
        labels = torch.argmax(activities_scores, dim=1)
        correct = torch.sum(torch.eq(labels.int(), activities_in.int()).float())
        activities_meter.update(correct.item() / activities_scores.shape[0], activities_scores.shape[0])
        for t, p in zip(activities_in.tolist(), labels.tolist()):
            conf[t, p] += 1
        loss_meter.update(total_loss.item(), batch_size)
        optimizer.zero_grad()
        total_loss.backward()
        if grad_buckets is not None:
            grad_buckets.allreduce()
        optimizer.step()
    confn = conf.float()
    per_class = confn.diag() / confn.sum(1).clamp(min=1)
    return {"time": epoch_timer.timeit(), "epoch": epoch, "loss": loss_meter.avg,
            "activities_acc": activities_meter.avg * 100, "activities_conf": confn.numpy(),
            "activities_MPCA": float(per_class.mean() * 100)}


def test_volleyball(data_loader, model, device, epoch, cfg):
    model.eval()
    activities_meter, loss_meter, epoch_timer = AverageMeter(), AverageMeter(), Timer()
    conf = torch.zeros(cfg.num_activities, cfg.num_activities, dtype=torch.int64)
    with torch.no_grad():
        for batch_data in data_loader:
            batch_data = [b.to(device=device) for b in batch_data]
            batch_size, num_frames = batch_data[0].shape[0], batch_data[0].shape[1]
            activities_in = batch_data[3].reshape((batch_size, num_frames))[:, 0].reshape((batch_size,))
            ret = model((batch_data[0], batch_data[1]))
            scores = ret["activities"]
            loss = F.cross_entropy(scores, activities_in)
            labels = torch.argmax(scores, dim=1)
            correct = torch.sum(torch.eq(labels.int(), activities_in.int()).float())
            activities_meter.update(correct.item() / scores.shape[0], scores.shape[0])
            for t, p in zip(activities_in.tolist(), labels.tolist()):
                conf[t, p] += 1
            loss_meter.update(loss.item(), batch_size)
    confn = conf.float()
    per_class = confn.diag() / confn.sum(1).clamp(min=1)
    return {"time": epoch_timer.timeit(), "epoch": epoch, "loss": loss_meter.avg,
            "activities_acc": activities_meter.avg * 100, "activities_conf": confn.numpy(),
            "activities_MPCA": float(per_class.mean() * 100)}


def train_net(cfg, training_set=None, validation_set=None, max_steps=None):
    """Reference train_net (:27-157).  Launch one process per GPU with torchrun; single process works too."""
    rank, local_rank, world = parallel.init_from_env()
    np.random.seed(cfg.train_random_seed)
    torch.manual_seed(cfg.train_random_seed)
    random.seed(cfg.train_random_seed)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    training_set = training_set or SyntheticVolleyball(cfg, length=max(cfg.batch_size * 2, 4))
    validation_set = validation_set or SyntheticVolleyball(cfg, length=max(cfg.test_batch_size, 2), seed=1)
    per_rank = max(cfg.batch_size // world, 1)
    sampler = data.distributed.DistributedSampler(training_set, world, rank, shuffle=True) if world > 1 else None
    training_loader = data.DataLoader(training_set, batch_size=per_rank, shuffle=sampler is None, sampler=sampler, num_workers=0)
    validation_loader = data.DataLoader(validation_set, batch_size=cfg.test_batch_size, shuffle=False, num_workers=0)
    model = build_model(cfg)
    if cfg.training_stage == 2 and cfg.stage1_model_path:
        model.loadmodel(cfg.stage1_model_path)
    model = model.to(device)
    parallel.broadcast_parameters(model)
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = FusedAdam(params, lr=cfg.train_learning_rate, weight_decay=cfg.weight_decay)
    buckets = parallel.GradBuckets(params) if world > 1 else None
    infos = []
    for epoch in range(1, cfg.max_epoch + 1):
        if epoch in cfg.lr_plan:
            adjust_lr(optimizer, cfg.lr_plan[epoch])
        info = train_volleyball(training_loader, model, device, optimizer, epoch, cfg, buckets)
        if rank == 0:
            print_log(getattr(cfg, "log_path", None), "Train epoch %d: loss %.5f acc %.2f%%" % (epoch, info["loss"], info["activities_acc"]))
        if epoch % cfg.test_interval_epoch == 0:
            tinfo = test_volleyball(validation_loader, model, device, epoch, cfg)
            if rank == 0:
                print_log(getattr(cfg, "log_path", None), "Test epoch %d: loss %.5f acc %.2f%%" % (epoch, tinfo["loss"], tinfo["activities_acc"]))
                result_path = getattr(cfg, "result_path", None)
                if result_path:
                    state = {"epoch": epoch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}
                    torch.save(state, result_path + "/stage%d_epoch%d_%.2f%%.pth" % (cfg.training_stage, epoch, tinfo["activities_acc"]))
            info = dict(train=info, test=tinfo)
        infos.append(info)
        if max_steps is not None and epoch >= max_steps:
            break
    return infos
