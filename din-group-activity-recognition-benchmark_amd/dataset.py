"""return_dataset(cfg): reference dataset.py:7-50 for the two datasets of the DIN stage-2 path (cfg.data_path, cfg.train_seqs,
cfg.test_seqs, cfg.image_size, cfg.out_size, cfg.num_before / num_after / num_frames, cfg.training_stage)."""
from __future__ import annotations

import os
import pickle

from .collective import CollectiveDataset, collective_all_frames, collective_read_dataset
from .volleyball import VolleyballDataset, volley_all_frames, volley_read_dataset


def return_dataset(cfg, uint8_images: bool = True):
    finetune = cfg.training_stage == 1
    if cfg.dataset_name == "volleyball":
        train_anns, test_anns = volley_read_dataset(cfg.data_path, cfg.train_seqs), volley_read_dataset(cfg.data_path, cfg.test_seqs)
        anns = {**train_anns, **test_anns}
        with open(os.path.join(cfg.data_path, "tracks_normalized.pkl"), "rb") as fh:
            tracks = pickle.load(fh)
        mk = lambda frames, training: VolleyballDataset(                                                    # noqa: E731
            anns, tracks, frames, cfg.data_path, cfg.image_size, cfg.out_size, cfg.inference_module_name, num_boxes=cfg.num_boxes,
            num_before=cfg.num_before, num_after=cfg.num_after, is_training=training, is_finetune=finetune, uint8_images=uint8_images)
        train_frames, test_frames = volley_all_frames(train_anns), volley_all_frames(test_anns)
    elif cfg.dataset_name == "collective":
        train_anns, test_anns = collective_read_dataset(cfg.data_path, cfg.train_seqs), collective_read_dataset(cfg.data_path, cfg.test_seqs)
        mk = lambda frames, training: CollectiveDataset(                                                    # noqa: E731
            train_anns if training else test_anns, frames, cfg.data_path, cfg.image_size, cfg.out_size,   # (num_boxes: the class default 13, as reference dataset.py:36-42 -- NOT cfg.num_boxes)
            num_frames=cfg.num_frames, is_training=training, is_finetune=finetune, uint8_images=uint8_images)
        train_frames, test_frames = collective_all_frames(train_anns), collective_all_frames(test_anns)
    else:
        raise AssertionError(cfg.dataset_name)
    print("Reading dataset finished...")
    print("%d train samples" % len(train_frames))
    print("%d test samples" % len(test_frames))
    return mk(train_frames, True), mk(test_frames, False)
