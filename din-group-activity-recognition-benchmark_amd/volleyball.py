"""Volleyball input contract: annotation files + normalised tracks + JPEG frames -> the tensors `Dynamic_volleyball` takes.

Mirrors the reference's volleyball.py interface for the stage-2 path (names, argument meaning, tuple layout), written for the MI355X feed:
  * frames stay **uint8** [T, 3, H, W] (reference: `torch.from_numpy(images).float()`, volleyball.py:270 -- 4x the bytes over PCIe and a
    float image the first conv would only re-read); the HIP image layer normalises uint8 itself (csrc image loaders, utils.prep_images).
    `uint8_images=False` restores the reference's float tensor.  The VALUES are the same either way: PIL decode + bilinear resize
    (torchvision 0.4.0 `transforms.functional.resize(img, (h, w))` == `img.resize((w, h), Image.BILINEAR)`), HWC -> CHW.
  * track boxes: normalised (y1, x1, y2, x2) -> feature-map pixels (x1*OW, y1*OH, x2*OW, y2*OH) (volleyball.py:246-251).  The reference
    multiplies in the tracks' dtype (float64 in tracks_normalized.pkl) and casts to float32 at the end (:271); so does this, in one
    vectorised expression -- bit-identical boxes.
  * short tracks are padded BY REPETITION of the leading boxes / actions up to num_boxes (volleyball.py:258-260).  The reference's
    expression only reaches num_boxes when at least half of the players are tracked (n >= num_boxes / 2), and crashes in its reshape
    otherwise; here fewer players wrap around cyclically (same result wherever the reference has one).
Checked against the reference's own `VolleyballDataset` on a synthetic annotation tree: tests/golden/dataset_volleyball.npz
(tools/gen_golden.py --only dataset), tests/test_host_cpu.py::test_volleyball_dataset_matches_reference_golden.
"""
from __future__ import annotations

import os
import random
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from PIL import Image
from torch.utils import data

ACTIVITIES = ["r_set", "r_spike", "r-pass", "r_winpoint", "l_set", "l-spike", "l-pass", "l_winpoint"]     # volleyball.py:20-21
ACTIONS = ["blocking", "digging", "falling", "jumping", "moving", "setting", "spiking", "standing", "waiting"]   # volleyball.py:25-27
NUM_ACTIVITIES, NUM_ACTIONS = len(ACTIVITIES), len(ACTIONS)
_ACTIVITY_ID = {n: i for i, n in enumerate(ACTIVITIES)}
_ACTION_ID = {n: i for i, n in enumerate(ACTIONS)}


def volley_read_annotations(path: str) -> Dict[int, dict]:
    """one sequence's annotations.txt: `<fid>.jpg <activity> {x y w h <action>}*` per line -> {fid: {file_name, group_activity,
    actions [n], bboxes int [n, 4] as (y1, x1, y2, x2) in source pixels}} (volleyball.py:31-66)"""
    out: Dict[int, dict] = {}
    with open(path) as fh:
        for line in fh:
            tok = line.rstrip("\n").split(" ")
            if len(tok) < 2 or not tok[0]:
                continue
            people = tok[2:]
            n = len(people) // 5
            xywh = np.array([[int(v) for v in people[5 * i:5 * i + 4]] for i in range(n)], dtype=np.int64).reshape(n, 4)
            x, y, w, h = (xywh[:, j] for j in range(4))
            out[int(tok[0].split(".")[0])] = {
                "file_name": tok[0],
                "group_activity": _ACTIVITY_ID[tok[1]],
                "actions": [_ACTION_ID[people[5 * i + 4]] for i in range(n)],
                "bboxes": np.stack([y, x, y + h, x + w], axis=1),
            }
    return out


def volley_read_dataset(path: str, seqs: Sequence[int]) -> Dict[int, Dict[int, dict]]:
    return {sid: volley_read_annotations(os.path.join(path, str(sid), "annotations.txt")) for sid in seqs}


def volley_all_frames(anns: Dict[int, Dict[int, dict]]) -> List[Tuple[int, int]]:
    return [(sid, fid) for sid, per_seq in anns.items() for fid in per_seq]


def volley_frames_around(frame: Tuple[int, int], num_before: int = 5, num_after: int = 4) -> List[Tuple[int, int, int]]:
    sid, src = frame
    return [(sid, src, fid) for fid in range(src - num_before, src + num_after + 1)]


def load_frame_u8(path: str, image_size: Tuple[int, int]) -> np.ndarray:
    """JPEG -> uint8 [3, H, W]: decode, bilinear resize to image_size = (H, W), HWC -> CHW (volleyball.py:239-243)"""
    with Image.open(path) as img:
        img = img.convert("RGB") if img.mode != "RGB" else img
        arr = np.asarray(img.resize((image_size[1], image_size[0]), Image.BILINEAR))
    return np.ascontiguousarray(arr.transpose(2, 0, 1))


def tracks_to_boxes(track: np.ndarray, feature_size: Tuple[int, int]) -> np.ndarray:
    """normalised (y1, x1, y2, x2) [n, 4] -> (x1*OW, y1*OH, x2*OW, y2*OH) in the track's own dtype (volleyball.py:246-251)"""
    oh, ow = feature_size
    t = np.asarray(track)
    return t[:, [1, 0, 3, 2]] * np.array([ow, oh, ow, oh], dtype=t.dtype if t.dtype.kind == "f" else np.float64)


def pad_by_repetition(rows, num: int):
    """first rows repeated cyclically up to `num` (volleyball.py:258-260 for n >= num / 2)"""
    n = len(rows)
    if n == 0 or n > num:
        raise ValueError(f"pad_by_repetition: {n} annotated boxes cannot be padded to num_boxes = {num}")
    if n == num:
        return rows
    idx = np.arange(num) % n
    return rows[idx] if isinstance(rows, np.ndarray) else [rows[i] for i in idx]


class VolleyballDataset(data.Dataset):
    """reference volleyball.py:146-275 (constructor arguments in the same order); item = (images, bboxes, actions, activities):
    images uint8 [T, 3, H, W] (float32 when uint8_images=False), bboxes float32 [T, N, 4] in feature px, actions int64 [T, N],
    activities int64 [T]."""

    def __init__(self, anns, tracks, frames, images_path, image_size, feature_size, inference_module_name="dynamic_volleyball", num_boxes=12,
                 num_before=4, num_after=4, is_training=True, is_finetune=False, uint8_images=True):
        self.anns, self.tracks, self.frames = anns, tracks, frames
        self.images_path, self.image_size, self.feature_size = images_path, tuple(image_size), tuple(feature_size)
        if inference_module_name == "arg_volleyball":
            # (reference volleyball.py:207-212: ARG samples 3 random frames in training and a fixed 9-frame order in test -- a stage-2 baseline
            #  outside the DIN path; refusing is better than silently feeding it full windows)
            raise NotImplementedError("VolleyballDataset: the 'arg_volleyball' frame sampling is not ported (DIN stage-2 path only)")
        self.inference_module_name = inference_module_name
        self.num_boxes, self.num_before, self.num_after = num_boxes, num_before, num_after
        self.is_training, self.is_finetune, self.uint8_images = is_training, is_finetune, uint8_images

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, index):
        return self.load_samples_sequence(self.volley_frames_sample(self.frames[index]))

    def volley_frames_sample(self, frame):
        """stage 2 (DIN): the whole window src-num_before .. src+num_after, training and test alike (volleyball.py:214-219); stage 1
        (is_finetune): one random frame of the window in training (:187-193)"""
        sid, src = frame
        if self.is_finetune and self.is_training:
            return [(sid, src, random.randint(src - self.num_before, src + self.num_after))]
        return volley_frames_around(frame, self.num_before, self.num_after)

    def load_samples_sequence(self, select_frames):
        images = np.stack([load_frame_u8(os.path.join(self.images_path, str(sid), str(src), f"{fid}.jpg"), self.image_size)
                           for sid, src, fid in select_frames])
        boxes = np.stack([pad_by_repetition(tracks_to_boxes(self.tracks[(sid, src)][fid], self.feature_size), self.num_boxes)
                          for sid, src, fid in select_frames])
        actions = np.array([pad_by_repetition(list(self.anns[sid][src]["actions"]), self.num_boxes) for sid, src, _ in select_frames],
                           dtype=np.int64)
        activities = np.array([self.anns[sid][src]["group_activity"] for sid, src, _ in select_frames], dtype=np.int64)
        img = torch.from_numpy(images)
        return (img if self.uint8_images else img.float(), torch.from_numpy(boxes.astype(np.float32)),
                torch.from_numpy(actions), torch.from_numpy(activities))
