"""Host-side helpers mirroring the names the DIN path uses from the reference's utils.py (prep_images :8-19, print_log
:101-105, AverageMeter :161-179, Timer :181-191).  prep_images runs on the device through the C ABI."""
from __future__ import annotations

import time

from . import ops


def prep_images(images):
    """y = ((x / 255) - 0.5) * 2 with the reference's three fp32 roundings (utils.py:8-19)."""
    return ops.prep_images_f32(images)


def print_log(file_path, *args):
    print(*args)
    if file_path is not None:
        with open(file_path, "a") as f:
            print(*args, file=f)


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)


class Timer:
    def __init__(self):
        self.last_time = time.time()

    def timeit(self):
        old, self.last_time = self.last_time, time.time()
        return self.last_time - old
