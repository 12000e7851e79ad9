#!/usr/bin/env python3
"""bench.py -- clips/sec (fwd+bwd+Adam) of the DIN stage-2 hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run, one
rank per GPU over RCCL.  W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + torch.cuda.synchronize()
on both sides, MAX over ranks, rank 0 prints ONE JSON line.

Workload (BASELINE.json): default = configs[1] "Volleyball stage-2 DIN, Inception-v3, T=3 ST_kernel=(3,3) N=12, bf16"
with the global batch of configs[2] (32 clips) sharded over the ranks (strong scaling: total work fixed).
`--workload vgg16_fp32|vgg16_bf16|inv3_fp32` select the other single-GPU configurations.
A step = forward + cross-entropy + backward + gradient all-reduce (N>1) + fused Adam over one batch of synthetic,
HBM-resident uint8 clips (inputs are on the device before the timed region starts).

Extra objects on the JSON line:
  roofline      dominant kernel (the conv kernel with the largest summed launch time of a surveyed step; Inception bf16: the pipelined
                weight-gradient kernel conv_wgrad_pipe_kernel<192, 256, true, 8>): algorithmic FLOPs per launch / live HIP-event launch
                time, against the dense MFMA peak of the compute dtype
  cpu_baseline  the CPU oracle (oracle/din_oracle.py, torch-CPU fp32 "port") timed on this box's host cores on a bounded
                sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL otherwise fails with hipIpcGetMemHandle: invalid argument);
# the launch environment normally exports it already -- must be in place before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist
import torch.nn.functional as F

_DEFAULT_THREADS = torch.get_num_threads()          # what torch picked for this process before the CPU baseline probes thread counts

WORKLOADS = {
    # name: (backbone, dtype, (OH, OW), D)
    "inv3_bf16": ("inv3", "bf16", (87, 157), 1056),
    "inv3_fp32": ("inv3", "fp32", (87, 157), 1056),
    "vgg16_bf16": ("vgg16", "bf16", (22, 40), 512),
    "vgg16_fp32": ("vgg16", "fp32", (22, 40), 512),
    # BASELINE configs[4]: Collective Activity stage-2 DIN (scripts/train_collective_stage2_dynamic.py: 480x720 frames, T = 10, up to 13
    # actors, 4 activities, dropout 0.5; its VGG16 option -- the res18 default is not a hot-path backbone), variable actors per clip
    "collective_bf16": ("vgg16", "bf16", (15, 22), 512),
    "collective_fp32": ("vgg16", "fp32", (15, 22), 512),
}
COLLECTIVE = {"H": 480, "W": 720, "N": 13, "T": 10, "activities": 4, "global_batch": 8, "dropout": 0.5}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0                               # HBM3E spec peak, MI355X_MICROARCH.md (6.3 TB/s is what a plain copy achieves)
# newest committed per-kernel HBM traffic table (profiles/rNN_pmc_traffic.json: tools/pmc_traffic.py over two rocprofv3 --pmc passes)
TRAFFIC_FILE = (sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")) or ["r03_pmc_traffic.json"])[-1]


def make_cfg(workload, T=3, N=12, H=720, W=1280, lite=None, hierarchical=False):
    from din_amd.config import Config
    backbone, dt, out_size, D = WORKLOADS[workload]
    collective = workload.startswith("collective")
    cfg = Config("collective" if collective else "volleyball")
    cfg.backbone, cfg.backbone_dtype, cfg.out_size, cfg.emb_features = backbone, dt, out_size, D
    cfg.image_size, cfg.num_frames, cfg.num_boxes = (H, W), T, N
    # BASELINE configs[3]: ST-factorised kernels [(1,3),(3,1)] run hierarchically (DPI_1 -> LN -> ReLU -> dropout -> DPI_2)
    cfg.ST_kernel_size = [(1, 3), (3, 1)] if hierarchical else ((3, 3) if collective else [(3, 3)])
    cfg.sampling_ratio, cfg.num_DIM = [1], 1
    cfg.dynamic_sampling, cfg.scale_factor, cfg.beta_factor = True, True, False
    cfg.lite_dim, cfg.hierarchical_inference, cfg.train_backbone = lite, hierarchical, True
    cfg.train_dropout_prob, cfg.set_bn_eval = 0.3, True
    if collective:
        cfg.inference_module_name, cfg.num_activities, cfg.train_dropout_prob = "dynamic_collective", COLLECTIVE["activities"], COLLECTIVE["dropout"]
    return cfg


def synth_weights(model, seed=3):
    """seeded synthetic weights: He-normal convs/linears are the module defaults; DIN predictors N(0, 0.02) so the
    dynamic-walk path is exercised (the reference zero-inits them); BN running stats non-trivial."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "p_conv" in name or "scale_conv" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        for name, b in model.named_buffers():
            if name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            elif name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))


def synth_boxes_labels(b, t, n, oh, ow, num_classes=8, seed=0):
    """player-shaped boxes in feature-map pixels + activity labels (SURVEY 8d recipe; the same for every rank count)"""
    import numpy as np
    r1 = np.random.default_rng(seed + 1)
    cx, cy = r1.uniform(0.05, 0.95, (b, t, n)) * ow, r1.uniform(0.3, 0.9, (b, t, n)) * oh
    bw, bh = r1.uniform(0.03, 0.08, (b, t, n)) * ow, r1.uniform(0.15, 0.35, (b, t, n)) * oh
    boxes = np.stack([np.clip(cx - bw / 2, 0, ow), np.clip(cy - bh / 2, 0, oh), np.clip(cx + bw / 2, 0, ow), np.clip(cy + bh / 2, 0, oh)], -1)
    labels = np.random.default_rng(seed + 2).integers(0, num_classes, size=(b,)).astype(np.int64)
    return torch.from_numpy(boxes.astype(np.float32)), torch.from_numpy(labels)


def cpu_baseline(workload, T, N, H, W, budget_s=12.0, lite=None, hierarchical=False, B=1, timed=3):
    """Oracle fwd+bwd on the host cores for a bounded sample (B clips per step; default 1).

    The baseline gets its best shot: the thread count is probed (below), then >= `timed` full-size steps are timed at the fastest one;
    `cores` reports the thread count actually used for the quoted number."""
    from oracle import din_oracle as O
    backbone, _dt, (OH, OW), D = WORKLOADS[workload]
    ncpu = os.cpu_count() or 1
    collective = workload.startswith("collective")
    ocfg = O.OracleCfg(backbone=backbone, image_size=(H, W), out_size=(OH, OW), emb_features=D, num_boxes=N, num_frames=T,
                       lite_dim=lite, hierarchical_inference=hierarchical,
                       ST_kernel_size=[(1, 3), (3, 1)] if hierarchical else ((3, 3) if collective else [(3, 3)]),
                       collective=collective, num_activities=COLLECTIVE["activities"] if collective else 8)
    p = O.synth_params(O.model_param_shapes(ocfg), seed=3, din_std=0.02)
    p = {k: v.requires_grad_("running_" not in k) for k, v in p.items()}
    images, boxes, labels = O.synth_inputs(B, T, N, H, W, OH, OW, ocfg.num_activities, seed=0)
    images = images.float()
    counts = torch.full((B, T), max(1, N // 2), dtype=torch.int32)
    if collective:
        boxes[:, :, N // 2:] = 0.0

    def step():
        for v in p.values():
            v.grad = None
        if collective:
            out = O.dynamic_collective_forward(ocfg, p, images, boxes, counts)
        else:
            out = O.dynamic_volleyball_forward(ocfg, p, images, boxes)
        F.cross_entropy(out["activities"], labels).backward()

    # Thread count: probed on a PROXY of the same model at a quarter of the pixels (one step each, ascending 16 / 32 / 64 / 128 / 256 capped
    # at the host's logical CPUs; the climb stops once a count is > 1.3x slower than the best so far -- torch-CPU does not scale to
    # every SMT thread of a 2-socket host: 256 threads were measured 40x slower than 64 on a 2 x EPYC 9575F box, which is why the
    # full-size step is not used for probing).  The quoted number is then the MEAN of `timed` full-size steps (>= 3) after one
    # untimed warm-up step at the fastest count; `cores` = that count.
    cands = sorted({max(1, min(ncpu, c)) for c in (16, 32, 64, 128, 256)})
    def fm_size(x):                                            # feature-map extent of the trunk for an image extent x (layer arithmetic)
        if backbone == "vgg16":
            return x // 32
        x = (x - 3) // 2 + 1 - 2                               # Conv2d_1a (3x3 / 2), 2a (3x3), 2b (3x3 pad 1)
        x = (x - 3) // 2 + 1 - 2                               # max-pool 3 / 2, 3b (1x1), 4a (3x3)
        return (x - 3) // 2 + 1                                # max-pool 3 / 2 -> Mixed_5 grid
    pcfg = O.OracleCfg(backbone=backbone, image_size=(H // 2, W // 2), out_size=(fm_size(H // 2), fm_size(W // 2)), emb_features=D, num_boxes=N,
                       num_frames=T, lite_dim=lite, hierarchical_inference=hierarchical, ST_kernel_size=ocfg.ST_kernel_size,
                       collective=collective, num_activities=ocfg.num_activities)
    pimages, pboxes, plabels = O.synth_inputs(1, T, N, H // 2, W // 2, pcfg.out_size[0], pcfg.out_size[1], ocfg.num_activities, seed=0)
    pimages = pimages.float()
    pcounts = torch.full((1, T), max(1, N // 2), dtype=torch.int32)

    def proxy_step():
        for v in p.values():
            v.grad = None
        if collective:
            out = O.dynamic_collective_forward(pcfg, p, pimages, pboxes, pcounts)
        else:
            out = O.dynamic_volleyball_forward(pcfg, p, pimages, pboxes)
        F.cross_entropy(out["activities"], plabels).backward()

    probe = {}
    best_t, best_th = None, None
    try:
        for th in cands:
            torch.set_num_threads(th)
            if best_t is None:
                proxy_step()                                   # first touch (allocator, thread pool) is not a measurement
            t0 = time.time()
            proxy_step()
            dt = time.time() - t0
            probe[str(th)] = round(dt, 3)
            if best_t is None or dt < best_t:
                best_t, best_th = dt, th
            elif dt > 1.3 * best_t:
                break
    except (RuntimeError, AssertionError):                     # the proxy frame is too small for this backbone's arithmetic: fall back
        best_th = min(cands, key=lambda c: abs(c - 64))
    torch.set_num_threads(best_th)
    step()                                                     # untimed warm-up at the chosen count
    t0 = time.time()
    n = 0
    while n < timed or (time.time() - t0 < budget_s and n < 6):
        step()
        n += 1
    dt = (time.time() - t0) / n
    try:
        allowed = len(os.sched_getaffinity(0))                 # CPUs this process may run on (container cpuset), not the host's count
    except (AttributeError, OSError):
        allowed = None
    return {"value": B / dt, "unit": "clips/sec", "cores": best_th, "kind": "port", "host_logical_cpus": ncpu, "cpus_allowed": allowed,
            "torch_intraop_threads_default": _DEFAULT_THREADS, "thread_probe_s": probe,
            "sample": f"{n} timed fwd+bwd step(s) (mean; 1 untimed warm-up) of B={B} clip(s) (T={T}, {H}x{W}, {backbone}" + (f", lite_dim={lite}" if lite else "") +
                      (", hierarchical" if hierarchical else "") + (f", collective with {max(1, N // 2)} of {N} actors" if collective else "") +
                      f", fp32 torch-CPU oracle) at the fastest of threads={list(probe)} (thread_probe_s: seconds per step of the half-resolution proxy)"}


def parity_mode_sample(a, dev, T, N, H, W, clips=8, steps=5):
    """clips/sec of the SAME workload in the fp32 parity mode (fp32 storage, exact-fp32 MFMA): the only mode that meets north_star's
    1e-4 logits bar (tests/test_gpu_din_model.py::test_full_size_fp32_model_matches_reference_golden).  Bounded sample: `clips` clips,
    one warm-up + `steps` timed steps of fwd + CE + bwd + fused Adam, HBM-resident uint8 clips."""
    from din_amd.infer_model import Dynamic_volleyball
    from din_amd.optim import FusedAdam
    from din_amd.train_net_dynamic import set_bn_eval
    wl = a.workload.replace("bf16", "fp32")
    backbone, dtype, (OH, OW), D = WORKLOADS[wl]
    cfg = make_cfg(wl, T, N, H, W, lite=a.lite_dim, hierarchical=a.hierarchical)
    torch.manual_seed(0)
    model = Dynamic_volleyball(cfg)
    synth_weights(model)
    model = model.to(dev).train()
    model.apply(set_bn_eval)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, weight_decay=0.0)
    g = torch.Generator().manual_seed(2000)
    images = torch.randint(0, 256, (clips, T, 3, H, W), dtype=torch.uint8, generator=g).to(dev)
    boxes, labels = synth_boxes_labels(clips, T, N, OH, OW, cfg.num_activities, seed=0)
    boxes, labels = boxes.to(dev), labels.to(dev)

    def step():
        opt.zero_grad()
        loss = F.cross_entropy(model((images, boxes))["activities"], labels)
        loss.backward()
        opt.step()
        return loss
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"value": round(clips / dt, 2), "unit": "clips/sec", "dtype": "fp32", "clips_per_step": clips, "steps": steps,
           "ms_per_step": round(dt * 1e3, 2), "final_loss": round(float(loss.item()), 5),
           "note": "same workload in the fp32 parity mode (logits within 1e-4 of the reference: tests/golden/full_*.npz), "
                   "fwd + CE + bwd + fused Adam, 1 warm-up step"}
    del model, opt, images
    torch.cuda.empty_cache()
    return out


class ClockSampler:
    """Samples the GPU's shader clock and socket power (amdsmi) on a background thread while the timed region runs.  The MI355X runs under a
    1400 W package cap: under these kernels the firmware lowers sclk below the 2.4 GHz the 2.5 PFLOP/s peak is quoted at (measured with
    tools/clock_probe.sh: 2.03 GHz under the gather kernel alone, 1.76 GHz under the Conv2d_2b stem kernel, ~2.28 GHz over the step), so the
    bench line states the clock its roofline fractions were reached at.  Reporting only: nothing here changes what is timed (one amdsmi query
    per 100 ms, made by ctypes calls that release the GIL)."""
    def __init__(self, index):
        self.samples, self._stop, self._thr, self.cap_w, self.error = [], None, None, None, None
        try:
            import threading
            import amdsmi
            amdsmi.amdsmi_init()
            self._smi, self._h = amdsmi, amdsmi.amdsmi_get_processor_handles()[index]
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(self._h)
                self.cap_w = float(cap.get("power_cap", 0)) / (1e6 if float(cap.get("power_cap", 0)) > 1e5 else 1.0) or None
            except Exception:
                pass
            self._stop = threading.Event()
            self._thr = threading.Thread(target=self._run, daemon=True)
        except Exception as e:                               # no amdsmi / no permission: the bench line says so and carries on
            self.error = f"{type(e).__name__}: {e}"

    def _one(self):
        m = self._smi.amdsmi_get_gpu_metrics_info(self._h)
        clk = m.get("current_gfxclks") or [m.get("current_gfxclk")]
        clk = [c for c in clk if isinstance(c, (int, float)) and 0 < c < 10000]
        pw = m.get("current_socket_power") or m.get("average_socket_power")
        return (sum(clk) / len(clk) if clk else None, float(pw) if isinstance(pw, (int, float)) and 0 < pw < 10000 else None)

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._one())
            except Exception as e:
                self.error = f"{type(e).__name__}: {e}"
                return
            self._stop.wait(0.1)

    def start(self):
        if self._thr is not None:
            self._thr.start()

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=2.0)
        clk = [c for c, _ in self.samples if c]
        pw = [w for _, w in self.samples if w]
        if not clk:
            return {"sclk_mhz_avg": None, "note": f"not sampled ({self.error or 'timed region shorter than one sample'})"}
        return {"sclk_mhz_avg": round(sum(clk) / len(clk)), "sclk_mhz_min": round(min(clk)), "sclk_mhz_max": round(max(clk)),
                "socket_power_w_avg": round(sum(pw) / len(pw)) if pw else None, "power_cap_w": self.cap_w, "samples": len(clk),
                "source": "amdsmi gpu_metrics (mean of the XCDs' current_gfxclks), one sample per 100 ms inside the timed region"}


def _self_launch(n: int) -> int:
    """`python bench.py --gpus N` without torchrun's environment: re-execute under torch.distributed.run, one rank per GPU (the driver's own
    form, BASELINE.json metric at 2 / 4 / 8 GPUs).  Before round 5 this case silently measured ONE GPU (WORLD_SIZE unset -> world 1)."""
    import socket
    import subprocess
    if os.environ.get("DIN_DIST_BACKEND") != "gloo" and os.environ.get("DIN_SINGLE_DEVICE") != "1":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:                                           # say so HERE, once, instead of N tracebacks from inside the ranks
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {have} (one rank per GPU over RCCL; "
                  f"DIN_SINGLE_DEVICE=1 DIN_DIST_BACKEND=gloo runs the N-rank control flow on one device for debugging)", file=sys.stderr, flush=True)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC for RCCL; must be set before the ranks start their HIP runtime
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def _rank_device_check(rank: int, world: int, gpus: int, dev) -> dict:
    """Every multi-rank run proves on its first step that the collective library sees `world` ranks on `world` distinct devices and that
    an all-reduce really sums over them (replaces nn.DataParallel, reference train_net_dynamic.py:95-96).  Returns the `rccl` block of the
    JSON line."""
    assert dist.is_initialized() and dist.get_world_size() == gpus == world, (dist.is_initialized() and dist.get_world_size(), gpus, world)
    backend = dist.get_backend()
    on_gpu = dev.type == "cuda"
    ident = (rank, torch.cuda.current_device() if on_gpu else -1, torch.cuda.get_device_properties(dev).name if on_gpu else "cpu",
             os.getpid())
    ids = [None] * world
    dist.all_gather_object(ids, ident)
    assert len({i[3] for i in ids}) == world, f"ranks share a process: {ids}"
    if on_gpu and os.environ.get("DIN_SINGLE_DEVICE") != "1":
        assert backend == "nccl", f"GPU ranks must talk through RCCL (backend 'nccl'), got {backend}"
        assert len({i[1] for i in ids}) == world, f"ranks share devices: {ids}"
    probe = torch.ones(1, device=dev) * (rank + 1)
    dist.all_reduce(probe)
    assert float(probe) == world * (world + 1) / 2, float(probe)
    info = {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "world_size": world,
            "devices": [i[1] for i in ids], "device_name": ids[0][2], "pids_distinct": True, "allreduce_probe_sum": float(probe)}
    if rank == 0:
        print(f"rank/device check: {info}", file=sys.stderr, flush=True)
    return info


def _dry_run_dist(a) -> None:
    """--dry-run-dist: the launch / rendezvous / sharding / all-reduce / JSON plumbing of a multi-rank run WITHOUT the model (CPU test of
    the N > 1 entry path under gloo; the product path itself needs the MI355X and is not touched here)."""
    from din_amd import parallel
    rank, local, world = parallel.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local) if (torch.cuda.is_available() and os.environ.get("DIN_DIST_BACKEND") != "gloo") else torch.device("cpu")
    rccl = _rank_device_check(rank, world, a.gpus, dev) if world > 1 else None
    gb = a.global_batch or 32
    mine = parallel.shard_range(gb, rank, world)
    torch.manual_seed(0)
    net = torch.nn.Linear(16, 8).to(dev)
    parallel.broadcast_parameters(net)
    clips = (torch.arange(gb * 16, dtype=torch.float32).reshape(gb, 16) / 100.0).to(dev)
    buckets = parallel.GradBuckets(net.parameters(), overlap=False) if world > 1 else None
    t0 = time.perf_counter()
    for _ in range(a.warmup + a.steps):
        net.zero_grad()
        (net(clips[mine.start:mine.stop]).pow(2).sum() / gb).backward()
        if buckets is not None:
            buckets.allreduce()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ref = torch.nn.Linear(16, 8).to(dev)
    ref.load_state_dict(net.state_dict())
    (ref(clips).pow(2).sum() / gb).backward()
    scale = float(world) if world > 1 else 1.0                  # local losses are divided by the GLOBAL batch: the rank average is 1/world of the full gradient
    err = max(float((p.grad * scale - q.grad).abs().max()) for p, q in zip(net.parameters(), ref.parameters()))
    assert err <= 1e-5, err
    if rank == 0:
        print(json.dumps({"metric": "dry run (no model): launch + rendezvous + shard + gradient all-reduce", "value": None, "unit": "clips/sec",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / max(a.steps + a.warmup, 1) * 1e3, 3),
                          "higher_is_better": True, "scaling": "strong", "dry_run": True, "rccl": rccl,
                          "config": {"global_batch": gb, "clips_per_gpu": len(mine), "parallelism": f"dp{world}"},
                          "allreduce_max_abs_err_vs_full_batch": err}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="inv3_bf16", choices=sorted(WORKLOADS))
    ap.add_argument("--global-batch", type=int, default=None, help="clips per step over all ranks (default 32; collective workloads 8)")
    ap.add_argument("--frames", type=int, default=None, help="frames per clip T (default 3; collective and --hierarchical 10)")
    ap.add_argument("--lite-dim", type=int, default=None, help="BASELINE configs[2]: lite-DIN projection width (128 in the reference)")
    ap.add_argument("--hierarchical", action="store_true", help="BASELINE configs[3]: ST-factorised [(1,3),(3,1)] hierarchical DIN (T defaults to 10)")
    ap.add_argument("--tce", action="store_true", help="Dynamic_TCE_volleyball (SURVEY 8(f)-4; scripts/train_volleyball_stage2_dynamic_tce.py: "
                    "vgg16 trunk, T defaults to 10): DIN behind the 4-head context-encoding transformer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 parity-mode sample, the HBM-group survey and the config-1 CPU baseline")
    ap.add_argument("--no-adam", action="store_true")
    ap.add_argument("--sustain", type=int, default=200, help="extra untimed-by-contract steps after the timed region (reported as `sustained`; 0: none)")
    ap.add_argument("--host-images", action="store_true", help="clips start in pinned host memory every step (uint8): the PCIe-inclusive rate, never the headline value")
    ap.add_argument("--forward-only", action="store_true", help="evaluation path (SURVEY 8f-3): model.eval(), torch.no_grad(), forward + loss only")
    ap.add_argument("--per-layer", default="", help="write the per-layer conv launch table of the sampled step to this file")
    ap.add_argument("--force-buckets", action="store_true",
                    help="single process: run the gradient-bucket path anyway (hook bookkeeping, one concatenation per bucket, a 1-rank all-reduce) "
                         "to measure what it costs on the host and in copy kernels -- compare ms_per_step with and without")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step's forward + loss + backward from a captured HIP graph (din_amd.graph_step; the all-reduce and the "
                         "optimizer stay eager calls).  auto = off (measured: no gain at 4 clips, a loss on 80-frame VGG16 steps); on: opt in")
    ap.add_argument("--bn-mode", default="eval", choices=["eval", "batch"],
                    help="Inception BatchNorm: 'eval' = cfg.set_bn_eval (running statistics, folded; results independent of the GPU count), "
                         "'batch' = the reference's stage-2 default (batch statistics of the rank's frames + running-stat update)")
    ap.add_argument("--dry-run-dist", action="store_true",
                    help="no model: only the multi-rank launch, rendezvous, clip sharding, a gradient all-reduce and the JSON line "
                         "(tests/test_host_cpu.py runs `bench.py --gpus 2 --dry-run-dist` under gloo)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(a.gpus))
    if a.dry_run_dist:
        return _dry_run_dist(a)

    from din_amd import parallel, profiling
    profiling.install()
    from din_amd.infer_model import Dynamic_collective, Dynamic_TCE_volleyball, Dynamic_volleyball
    from din_amd.optim import FusedAdam

    rank, local, world = parallel.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {a.gpus}` (it starts its own ranks) or torchrun"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    collective = a.workload.startswith("collective")
    if a.frames is None:
        a.frames = COLLECTIVE["T"] if collective else (10 if (a.hierarchical or a.tce) else 3)
    if a.global_batch is None:
        a.global_batch = COLLECTIVE["global_batch"] if collective else 32
    T, N, H, W = (a.frames, COLLECTIVE["N"], COLLECTIVE["H"], COLLECTIVE["W"]) if collective else (a.frames, 12, 720, 1280)
    backbone, dtype, (OH, OW), D = WORKLOADS[a.workload]
    cfg = make_cfg(a.workload, T, N, H, W, lite=a.lite_dim, hierarchical=a.hierarchical)
    cfg.set_bn_eval = a.bn_mode == "eval"
    torch.manual_seed(0)
    if a.tce:
        assert a.workload.startswith("vgg16") and not a.lite_dim, "--tce: the reference model only runs on the vgg16 trunk without lite_dim"
        a.no_cpu_baseline = True                           # (the bounded CPU sample is defined for the DIN models of BASELINE.json only)
    model = (Dynamic_TCE_volleyball if a.tce else Dynamic_collective if collective else Dynamic_volleyball)(cfg)
    synth_weights(model)
    model = model.to(dev).train()
    if cfg.set_bn_eval:
        from din_amd.train_net_dynamic import set_bn_eval
        model.apply(set_bn_eval)                               # reference train_net_dynamic.py:98-100
    parallel.broadcast_parameters(model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdam(params, lr=1e-4, weight_decay=0.0)
    if a.force_buckets and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
    buckets = parallel.GradBuckets(params, force=a.force_buckets) if (world > 1 or a.force_buckets) else None
    # always (not only under DIN_CHECK_ALLREDUCE): N ranks, N distinct devices, an all-reduce that sums over all of them
    rccl_info = _rank_device_check(rank, world, a.gpus, dev) if world > 1 else None

    mine = parallel.shard_range(a.global_batch, rank, world)
    B = len(mine)
    g = torch.Generator().manual_seed(1000 + rank)
    images = torch.randint(0, 256, (B, T, 3, H, W), dtype=torch.uint8, generator=g).to(dev)     # uint8, HBM-resident
    boxes, labels = synth_boxes_labels(a.global_batch, T, N, OH, OW, cfg.num_activities, seed=0)
    counts = None
    if collective:                                         # 1..N actors per clip, the same count in all of a clip's frames; padding boxes zero
        import numpy as np
        cn = np.random.default_rng(5).integers(1, N + 1, size=(a.global_batch,))
        for b_ in range(a.global_batch):
            boxes[b_, :, int(cn[b_]):] = 0.0
        counts = torch.from_numpy(np.repeat(cn[:, None], T, 1).astype(np.int32))[mine.start:mine.stop].to(dev)
    boxes, labels = boxes[mine.start:mine.stop].to(dev), labels[mine.start:mine.stop].to(dev)
    batch = lambda im: (im, boxes, counts) if collective else (im, boxes)

    if a.forward_only:
        model.eval()

    feed = None
    if a.host_images:
        # every step's clips start in pinned host memory and cross PCIe on the copy stream of din_amd.input_feed.DeviceFeed while the
        # previous step computes (double buffer); 8.3 MB per clip
        from din_amd.input_feed import DeviceFeed
        images_host = images.cpu().pin_memory()
        def _forever():
            while True:
                yield images_host
        feed = iter(DeviceFeed(_forever(), dev))

    cap = None                                             # graph_step.CapturedStep once the warm-up steps are done

    def step(eager=False):
        nonlocal images
        if cap is not None and not eager:
            loss = cap.replay()                            # hipGraphLaunch: forward + cross-entropy + backward
            if buckets is not None:
                buckets.allreduce(scale_in_optimizer=not a.no_adam)
            if not a.no_adam:
                opt.step(grad_scale=buckets.grad_scale if buckets is not None else 1.0)
            return loss
        if feed is not None:
            images = next(feed)
        if a.forward_only:
            with torch.no_grad():
                return F.cross_entropy(model(batch(images))["activities"], labels)
        opt.zero_grad()
        ret = model(batch(images))
        loss = F.cross_entropy(ret["activities"], labels)
        loss.backward()
        if buckets is not None:
            buckets.allreduce(scale_in_optimizer=not a.no_adam)
            if os.environ.get("DIN_CHECK_ALLREDUCE") == "1":     # debugging aid: every rank must hold the same averaged gradients
                chk = torch.stack([p.grad.double().sum() for p in params if p.grad is not None]).sum().reshape(1)
                lo, hi = chk.clone(), chk.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                assert float((hi - lo).abs()) <= 1e-9 * max(1.0, float(hi.abs())), (float(lo), float(hi))
                if rank == 0:
                    print(f"allreduce check ok: grad checksum {float(chk):.6e}", file=sys.stderr, flush=True)
        if not a.no_adam:
            opt.step(grad_scale=buckets.grad_scale if buckets is not None else 1.0)
        return loss

    # The LAST warm-up step is run with every conv launch bracketed by HIP events: it names the dominant kernel (largest summed launch
    # time) and fills the per-kernel survey.  The timed region then brackets ONLY that kernel's launches (in its last step), so the
    # headline number pays for ~20 event pairs instead of ~190 (a fully bracketed step costs ~10 % of its own duration).
    survey, hbm = None, None
    for wi in range(a.warmup):
        if wi == a.warmup - 1:
            profiling.PROFILE = []
            if not a.no_extras:
                hbm = profiling.hbm_survey()
                hbm.__enter__()
        step()
        if wi == a.warmup - 1:
            torch.cuda.synchronize()
            survey, profiling.PROFILE = profiling.PROFILE, None
            if hbm is not None:
                hbm.__exit__(None, None, None)
    # auto = off: measured no gain where it was expected (4 clips: 10.27 -> 10.24 ms; the step is GPU-bound) and a 3x LOSS on the 80-frame VGG16
    # workloads (T = 10, 8 clips: 172 -> 514 ms per step when replayed; profiles/r03_wgrad_stationary.txt) -- opt-in only
    use_graph = a.graph == "on"
    if use_graph and not (a.forward_only or a.host_images or a.bn_mode == "batch") and a.warmup >= 2:
        from din_amd import graph_step
        try:
            cap = graph_step.CapturedStep(lambda: F.cross_entropy(model(batch(images))["activities"], labels), params,
                                          graph_step.dropout_counters(model))
            step()                                         # one untimed replay: the capture itself executed nothing
            torch.cuda.synchronize()
        except Exception as e:                             # stay on the eager path (and say so)
            print(f"HIP graph capture failed, running eagerly: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            cap = None
            opt.zero_grad()
    if os.environ.get("DIN_BENCH_TORCH_PROFILE"):          # tuning aid: which host-side torch ops launch the small fill / copy kernels
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            step()
            torch.cuda.synchronize()
        with open(os.environ["DIN_BENCH_TORCH_PROFILE"], "w") as f:
            f.write(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=200, max_shapes_column_width=90))
            f.write("\n\n==== ATen ops that launch something, by call site ====\n")
            sites = {}
            for ev in prof.events():
                if ev.device_type.name != "CPU" or not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
                    continue
                frames = [fr for fr in (ev.stack or []) if "din_amd" in fr or "din-group" in fr or "bench.py" in fr][:3]
                key = (ev.name, str(ev.input_shapes)[:80], " <- ".join(fr.split("/")[-1] for fr in frames))
                c = sites.setdefault(key, [0, 0.0])
                c[0] += 1; c[1] += ev.self_device_time_total
            for key, (n, us) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{n:4d} x {us:9.1f} us  {key[0]:28s} {key[1]:80s} {key[2]}\n")

    # calibrate what a HIP event pair adds around ONE launch (two marker packets; kernels otherwise run back to back):
    # per-launch cost of a trivial kernel bracketed individually minus its cost inside one long bracket
    from din_amd import ops as _ops
    tiny = torch.zeros(64, device=dev)
    def _tiny():
        _ops.L.check(_ops.L.load().din_axpby(tiny.data_ptr(), None, tiny.data_ptr(), 1.0, 0.0, 64, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    singles = []
    for _ in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _tiny(); e1.record()
        singles.append((e0, e1))
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record()
    for _ in range(64):
        _tiny()
    b1.record()
    torch.cuda.synchronize()
    single_ms = sorted(x.elapsed_time(y) for x, y in singles)[len(singles) // 2]
    event_overhead_ms = max(0.0, single_ms - b0.elapsed_time(b1) / 64)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    def aggregate(records):
        out = {}
        for kind, variant, flops, dt_, e0, e1, _name in records:
            rec = out.setdefault(variant, [0.0, 0.0, 0])
            rec[0] += flops
            rec[1] += max(e0.elapsed_time(e1) - event_overhead_ms, 0.0) * 1e-3
            rec[2] += 1
        return out

    # dominant kernel = the kernel (as rocprofv3 names it, tile template included) with the largest summed launch time in the survey
    survey_agg = aggregate(survey) if survey else {}
    dom_survey = max(survey_agg, key=lambda kname: survey_agg[kname][1]) if survey_agg else None
    # Per-launch HIP events cost a few us of pipeline bubble each (two marker packets).  Inside the timed region they are recorded
    # during the LAST step only and only around the dominant kernel's launches (every conv launch when there was no warm-up survey).
    sampler = ClockSampler(dev.index or 0) if rank == 0 else None
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    for it in range(a.steps):
        last = it == a.steps - 1
        if last:
            profiling.PROFILE, profiling.PROFILE_ONLY = [], dom_survey
            if cap is not None:
                opt.zero_grad()                            # the bracketed step runs eagerly (HIP events cannot time nodes of a replayed graph)
        loss = step(eager=last)
    host_enqueue_s = time.perf_counter() - t0                # host time to ENQUEUE the timed steps (nothing waits on the GPU inside a step)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    clocks = sampler.stop() if sampler is not None else None
    prof, profiling.PROFILE, profiling.PROFILE_ONLY = profiling.PROFILE, None, None
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # ---- sustained phase (outside the timed region; `value` is NOT taken from it) ------------------------------
    # The contract's K steps last ~1 s on this workload, shorter than the sampling period of a coarse utilisation monitor (the driver's
    # gpu_busy sampler recorded 0 active GPUs around the round-3 run).  The same step is therefore repeated for >= 200 more steps
    # (~10 s) right after the timed region -- GPU first, CPU baselines later -- and its rate is reported beside the headline one.
    sustained = None
    if world == 1 and rank == 0 and not (a.no_extras or a.forward_only) and a.sustain > 0:
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(a.sustain):
            step()
        torch.cuda.synchronize()
        sdt = time.perf_counter() - ts
        sustained = {"steps": a.sustain, "seconds": round(sdt, 2), "ms_per_step": round(sdt / a.sustain * 1e3, 3),
                     "value": round(B * a.sustain / sdt, 2), "unit": "clips/sec",
                     "note": "same step repeated after the timed region (not part of `value`): a long enough GPU phase for coarse utilisation samplers"}

    # ---- roofline of the dominant kernel from the live HIP events -------------------------------------------
    every = survey if survey else prof                     # all conv launches of one step: the warm-up survey, else the timed step itself
    if a.per_layer and rank == 0:
        rows = sorted(((max(e0.elapsed_time(e1) - event_overhead_ms, 0.0), kind, name, variant, flops) for kind, variant, flops, dt_, e0, e1, name in every),
                      reverse=True)
        with open(a.per_layer, "w") as f:
            for ms, kind, name, variant, flops in rows:
                f.write(f"{ms * 1e3:9.1f} us  {flops / max(ms, 1e-6) / 1e9:7.1f} TF  {kind:5s} {name:60s} {variant}\n")
    agg = survey_agg if survey_agg else aggregate(prof)
    timed_agg = aggregate(prof)                            # launches bracketed inside the timed region
    dom = dom_survey if dom_survey in timed_agg else (max(timed_agg, key=lambda kname: timed_agg[kname][1]) if timed_agg else "none")
    fl, sec, cnt = timed_agg.get(dom, [0.0, 1e-9, 1])
    achieved = fl / sec / 1e12
    peak = PEAK_TFLOPS[dtype]
    roofline = {"bound": "mfma", "kernel": dom,
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": None, "launches": cnt, "avg_launch_ms": round(sec / max(cnt, 1) * 1e3, 4),
                "event_pair_overhead_ms_subtracted": round(event_overhead_ms, 4), "sampled_steps": 1,
                "sampled_in": "last timed step (this kernel only)" if survey_agg else "last timed step (all conv launches)",
                "flops_per_launch_avg": fl / max(cnt, 1),
                "survey_of_all_conv_launches": "last warm-up step" if survey_agg else "last timed step",
                "all_conv_launches": {"TFLOP/s": round(sum(v[0] for v in agg.values()) / max(sum(v[1] for v in agg.values()), 1e-9) / 1e12, 2),
                                      "time_s": round(sum(v[1] for v in agg.values()), 4)},
                "other_kernels": {kname: {"TFLOP/s": round(v[0] / max(v[1], 1e-9) / 1e12, 2), "launches": v[2],
                                          "time_s": round(v[1], 4)} for kname, v in sorted(agg.items(), key=lambda kv: -kv[1][1]) if kname != dom}}
    # HBM traffic of the dominant kernel: PMC counters cannot be collected inside this process; they come from the separate rocprofv3
    # --pmc passes over this same command whose per-kernel result is committed under profiles/ (tools/pmc_traffic.py)
    try:
        tr = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", TRAFFIC_FILE)))
        if tr.get("workload") == a.workload and tr.get("global_batch") == a.global_batch and world == 1:
            prefix = dom.split(", ...>")[0]
            hits = [v for k, v in tr["kernels"].items() if k.startswith(prefix)]
            n = sum(h["launches"] for h in hits)
            if n:
                rd = sum(h["fetch_bytes_per_launch"] * h["launches"] for h in hits) / n
                wr = sum((h["write_bytes_per_launch"] or 0.0) * h["launches"] for h in hits) / n
                roofline["traffic"] = round(rd + wr)
                roofline["traffic_detail"] = {"unit": "HBM bytes per launch", "read": round(rd), "write": round(wr),
                                              "source": f"profiles/{TRAFFIC_FILE} (committed result of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                        "passes over this command, tools/pmc_traffic.py; not measured in this run)"}
    except (OSError, ValueError, KeyError):
        pass
    if clocks and clocks.get("sclk_mhz_avg"):
        # the same fraction against the MFMA peak AT THE CLOCK THE RUN HAD (the 2.5 PFLOP/s figure is 256 CUs x 4 SIMDs x 1024 FLOP/clk at 2.4 GHz);
        # `frac` above stays the contract's number (against the nominal peak)
        scaled = peak * clocks["sclk_mhz_avg"] / 2400.0
        roofline["peak_at_sampled_sclk"] = round(scaled, 1)
        roofline["frac_at_sampled_sclk"] = round(achieved / scaled, 4)
    conv_time = sum(v[1] for v in agg.values())

    if rank == 0:
        clips = a.global_batch * a.steps if world > 1 else B * a.steps
        out = {
            "metric": ("clips/sec (fwd only, eval), " if a.forward_only else "clips/sec (fwd+bwd), ") +
                      (f"Collective DIN stage-2, BxTx<={N} actors" if collective else "Volleyball DIN stage-2, BxTx12 actors"),
            "value": round(clips / elapsed, 3), "unit": "clips/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic" + (" (uint8 clips copied from pinned host memory every step)" if a.host_images else ""),
            "config": {"workload": (f"Collective stage-2 DIN, {backbone}, T={T}, ST_kernel=(3,3), 1..{N} actors per clip, {H}x{W}, {dtype}" if collective else
                                    ("Volleyball stage-2 TCE + DIN (Dynamic_TCE_volleyball)" if a.tce else "Volleyball stage-2 DIN") + f", {backbone}, T={T}, " +
                                    ("ST_kernel=[(1,3),(3,1)] hierarchical" if a.hierarchical else "ST_kernel=(3,3)") +
                                    (f", lite_dim={a.lite_dim}" if a.lite_dim else "") + f", N=12, {H}x{W}, {dtype}"),
                       "global_batch": a.global_batch, "clips_per_gpu": B, "frames": T, "parallelism": f"dp{world}",
                       "includes": "fwd + cross-entropy (eval mode, no_grad)" if a.forward_only else
                                   "fwd + cross-entropy + bwd" + (f" + grad all-reduce ({'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()})"
                                                                  if (world > 1 or buckets is not None) else "")
                                   + ("" if a.no_adam else " + fused Adam"),
                       "launch_mode": ("HIP graph replay of fwd + loss + bwd (din_amd.graph_step), eager all-reduce / optimizer; the last timed "
                                       "step runs eagerly for the per-launch HIP events" if cap is not None else "eager launches"),
                       "bn_mode": ("running statistics (set_bn_eval)" if cfg.set_bn_eval else "batch statistics (reference stage-2 default)")
                                  if backbone == "inv3" else "n/a"},
            "roofline": roofline,
            "rccl": rccl_info,
            "clocks": clocks,
            "conv_time_frac_sampled_step": round(conv_time / (elapsed / a.steps), 4),
            "host_enqueue_ms_per_step": round(host_enqueue_s / a.steps * 1e3, 3),
            "final_loss": round(float(loss.item()), 5),
        }
        if sustained is not None:
            out["sustained"] = sustained
        if hbm is not None:
            # HBM-bound kernel groups of the surveyed (last warm-up) step: ALGORITHMIC bytes / HIP-event time vs 8 TB/s
            out["roofline_hbm"] = dict(hbm.summary(event_overhead_ms, PEAK_HBM_GBS),
                                       source="algorithmic bytes per launch / HIP-event launch time, last warm-up step (din_amd/profiling.py::hbm_survey)")
        # ---- parity check of the benchmarked launch geometry (outside the timed region) ----------------------------
        # The B-clip forward (production planner at this batch size: register-resident 1x1, persistent stem kernels, ...) against the SAME
        # model run on clip 0 alone (the small-batch kernels the golden-tied tests see; clips are independent under running-statistics BN
        # and in eval mode).  tests/test_gpu_din_model.py::test_benchmarked_dispatch_matches_golden_and_single_clip_run ties the same
        # comparison to the reference's numbers; this line says the run that was just timed computes what those tests checked.
        if True:
            was_training = model.training
            model.eval()
            # (the hierarchical module's dropout is ALWAYS on, also under eval() -- the reference's `F.dropout(x)` with functional defaults,
            #  infer_module/dynamic_infer_module.py:495 -- and its mask depends on the position in the batch: switched off for this comparison)
            always_on = [(m, m.hier_dropout_p) for m in model.modules() if hasattr(m, "hier_dropout_p")]
            for m, _p in always_on:
                m.hier_dropout_p = 0.0
            with torch.no_grad():
                full = model(batch(images))["activities"].float()
                one = model(tuple(None if t is None else t[:1] for t in batch(images)))["activities"].float()
            for m, p_ in always_on:
                m.hier_dropout_p = p_
            if was_training:
                model.train()
                if cfg.set_bn_eval:
                    model.apply(set_bn_eval)
            top = float(full.abs().max())
            diff = float((full[:1] - one).abs().max()) / max(top, 1e-30)
            bar = 8e-3 if dtype == "bf16" else 1e-4
            out["parity_check"] = {"what": f"eval-mode logits of clip 0 inside the {B}-clip batch vs the same model on clip 0 alone "
                                           "(max abs difference / largest logit)", "value": float(f"{diff:.3e}"), "bar": bar,
                                   "finite": bool(torch.isfinite(full).all()), "ok": bool(diff <= bar and torch.isfinite(full).all())}
        plain = world == 1 and not (a.forward_only or a.tce or collective or a.no_extras or a.host_images or a.force_buckets)
        if plain and dtype == "bf16":
            out["parity_mode"] = parity_mode_sample(a, dev, T, N, H, W)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload, T, N, H, W, lite=a.lite_dim, hierarchical=a.hierarchical)
            out["vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            if plain and not (a.workload.startswith("vgg16") or a.lite_dim or a.hierarchical):
                # the CPU reference path's own configuration -- configs[0], VGG16, T=3, B=2 (one probe + timed steps)
                out["cpu_baseline_config1"] = cpu_baseline("vgg16_fp32", 3, 12, 720, 1280, budget_s=0.0, B=2, timed=3)
        print(json.dumps(out))
    if world > 1 or (a.force_buckets and dist.is_initialized()):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
