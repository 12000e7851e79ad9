"""Collective Activity input contract: annotations + JPEG frames -> the tensors `Dynamic_collective` takes.

Mirrors the reference's collective.py interface for the stage-2 path.  Differences by design (values identical): frames stay uint8
(`uint8_images=False` restores float32, collective.py:218), the box arithmetic is one vectorised float64 expression cast to float32 at the
end exactly as the reference's `np.array(..., dtype=np.float)` -> `.float()` (:212,219).  Rules kept:
  * a frame id is a clip anchor when fid % 10 == 1 and fid + 9 <= FRAMES_NUM[sid]; its group activity is the most common person action,
    skipping 'NA' (collective.py:49-56,71-79), then 5 -> 4 classes (Activity5to4, :37); person actions 6 -> 5 (Action6to5, :36);
  * boxes are normalised with the SEQUENCE's frame size (FRAMES_SIZE, :67-69) as (y1, x1, y2, x2) and scaled to feature px as
    (x1*OW, y1*OH, x2*OW, y2*OH) (:193-196); every frame of a clip uses the ANCHOR frame's boxes (:192: anns[sid][src_fid]);
  * frames with fewer than num_boxes people are padded with ZERO boxes and action -1, and `bboxes_num` carries the real count (:199-203).
Checked against the reference's own `CollectiveDataset`: tests/golden/dataset_collective.npz (tools/gen_golden.py --only dataset).
"""
from __future__ import annotations

import os
import random
from collections import Counter
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch.utils import data

from .volleyball import load_frame_u8

# per-sequence frame counts and frame sizes of the Collective Activity dataset (data, collective.py:12-24)
FRAMES_NUM = dict(zip(range(1, 45), [
    302, 347, 194, 257, 536, 401, 968, 221, 356, 302, 1813, 1084, 851, 723, 464, 1021, 905, 600, 203, 342, 650, 361, 311, 321, 617, 734, 1804,
    470, 635, 356, 690, 194, 193, 395, 707, 914, 1049, 653, 518, 401, 707, 420, 410, 356]))
FRAMES_SIZE = {sid: ((450, 800) if sid in (15, 20, 21, 22, 23, 24) else (480, 720)) for sid in range(1, 45)}

ACTIONS = ["NA", "Crossing", "Waiting", "Queueing", "Walking", "Talking"]
ACTIVITIES = ["Crossing", "Waiting", "Queueing", "Walking", "Talking"]
ACTIONS_ID = {a: i for i, a in enumerate(ACTIONS)}
ACTIVITIES_ID = {a: i for i, a in enumerate(ACTIVITIES)}
Action6to5 = {0: 0, 1: 1, 2: 2, 3: 3, 4: 1, 5: 4}
Activity5to4 = {0: 0, 1: 1, 2: 2, 3: 0, 4: 3}


def _group_activity(actions: List[int]) -> int:
    top = Counter(actions).most_common(2)
    return (top[0][0] if top[0][0] != 0 else top[1][0]) - 1


def collective_read_annotations(path: str, sid: int) -> Dict[int, dict]:
    """seq%02d/annotations.txt, tab-separated `frame x y w h action ...` rows grouped by frame -> {anchor fid: {frame_id,
    group_activity, actions [n], bboxes [n] of (y1, x1, y2, x2) normalised by the sequence's frame size}} (collective.py:40-81)"""
    per_frame: Dict[int, Tuple[List[int], List[Tuple[float, float, float, float]]]] = {}
    order: List[int] = []
    fh_, fw_ = FRAMES_SIZE[sid]
    with open(os.path.join(path, "seq%02d" % sid, "annotations.txt")) as fh:
        for line in fh:
            v = line.rstrip("\n").split("\t")
            if len(v) < 6:
                continue
            fid = int(v[0])
            if not order or order[-1] != fid:               # rows of a frame are contiguous; a frame id that comes back starts a new group
                order.append(fid)
                per_frame[fid] = ([], [])
            x, y, w, h = (int(t) for t in v[1:5])
            per_frame[fid][0].append(int(v[5]) - 1)
            per_frame[fid][1].append((y / fh_, x / fw_, (y + h) / fh_, (x + w) / fw_))
    out = {}
    for fid in order:
        if fid % 10 == 1 and fid + 9 <= FRAMES_NUM[sid]:
            actions, boxes = per_frame[fid]
            out[fid] = {"frame_id": fid, "group_activity": _group_activity(actions), "actions": actions, "bboxes": boxes}
    return out


def collective_read_dataset(path: str, seqs: Sequence[int]) -> Dict[int, Dict[int, dict]]:
    return {sid: collective_read_annotations(path, sid) for sid in seqs}


def collective_all_frames(anns) -> List[Tuple[int, int]]:
    return [(sid, fid) for sid in anns for fid in anns[sid]]


class CollectiveDataset(data.Dataset):
    """reference collective.py:96-225; item = (images, bboxes, actions, activities, bboxes_num): images uint8 [T, 3, H, W], bboxes float32
    [T, MAX_N, 4] (zero rows beyond the real count), actions int64 [T, MAX_N] (-1 beyond), activities int64 [T], bboxes_num int32 [T]."""

    def __init__(self, anns, frames, images_path, image_size, feature_size, num_boxes=13, num_frames=10, is_training=True, is_finetune=False,
                 uint8_images=True):
        self.anns, self.frames, self.images_path = anns, frames, images_path
        self.image_size, self.feature_size = tuple(image_size), tuple(feature_size)
        self.num_boxes, self.num_frames = num_boxes, num_frames
        self.is_training, self.is_finetune, self.uint8_images = is_training, is_finetune, uint8_images

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, index):
        return self.load_samples_sequence(self.get_frames(self.frames[index]))

    def get_frames(self, frame):
        sid, src = frame
        if self.is_finetune and self.is_training:
            return [(sid, src, random.randint(src, src + self.num_frames - 1))]
        return [(sid, src, fid) for fid in range(src, src + self.num_frames)]

    def load_samples_sequence(self, select_frames):
        oh, ow = self.feature_size
        T, N = len(select_frames), self.num_boxes
        images = np.stack([load_frame_u8(os.path.join(self.images_path, "seq%02d" % sid, "frame%04d.jpg" % fid), self.image_size)
                           for sid, _, fid in select_frames])
        boxes = np.zeros((T, N, 4), dtype=np.float64)
        actions = np.full((T, N), -1, dtype=np.int64)
        activities = np.empty(T, dtype=np.int64)
        count = np.empty(T, dtype=np.int32)
        for t, (sid, src, _) in enumerate(select_frames):
            ann = self.anns[sid][src]
            n = len(ann["bboxes"])
            if n > N:
                raise ValueError(f"seq{sid:02d} frame {src}: {n} people > num_boxes {N} (the reference loops forever here, collective.py:199)")
            if n:
                b = np.asarray(ann["bboxes"], dtype=np.float64)
                boxes[t, :n] = b[:, [1, 0, 3, 2]] * np.array([ow, oh, ow, oh], dtype=np.float64)
                actions[t, :n] = [Action6to5[a] for a in ann["actions"]]
            activities[t] = Activity5to4[ann["group_activity"]]
            count[t] = n
        img = torch.from_numpy(images)
        return (img if self.uint8_images else img.float(), torch.from_numpy(boxes.astype(np.float32)), torch.from_numpy(actions),
                torch.from_numpy(activities), torch.from_numpy(count))
