"""Config(dataset_name): attribute bag with the field names and defaults the DIN path reads from the reference's
config.py:10-104 (only the fields that exist on the hot path; dataset paths and the other methods' knobs are omitted).
Extra fields: backbone_dtype ('fp32' | 'bf16'); hier_dropout_p (the reference hard-codes F.dropout's defaults -- p = 0.5, always on --
at dynamic_infer_module.py:495; 0.5 keeps that, 0.0 switches it off)."""
from __future__ import annotations

import os
import time

_DEFAULTS = dict(
    image_size=(720, 1280), batch_size=32, test_batch_size=8, num_boxes=12,
    use_gpu=True, use_multi_gpu=True, device_list="0,1,2,3",
    backbone="res18", crop_size=(5, 5), train_backbone=False, out_size=(87, 157), emb_features=1056,
    num_actions=9, num_activities=8, actions_loss_weight=1.0, actions_weights=None,
    num_frames=3, num_before=5, num_after=4,
    num_features_boxes=1024, num_features_relation=256, num_graph=16, num_features_gcn=1024, gcn_layers=1,
    train_random_seed=0, train_learning_rate=1e-4, lr_plan={11: 3e-5, 21: 1e-5}, train_dropout_prob=0.3, weight_decay=0,
    max_epoch=30, test_interval_epoch=1,
    training_stage=1, stage1_model_path="", test_before_train=False, exp_note="Group-Activity-Recognition", exp_name=None,
    set_bn_eval=False, inference_module_name="dynamic_volleyball",
    stride=1, ST_kernel_size=3, dynamic_sampling=True, sampling_ratio=[1, 3], group=1, scale_factor=True, beta_factor=True,
    load_backbone_stage2=False, parallel_inference=False, hierarchical_inference=False, lite_dim=None, num_DIM=1,
    load_stage2model=False, stage2model=None,
    backbone_dtype="fp32", hier_dropout_p=0.5,
)


class Config(object):
    def __init__(self, dataset_name):
        assert dataset_name in ("volleyball", "collective")
        self.dataset_name = dataset_name
        for k, v in _DEFAULTS.items():
            setattr(self, k, list(v) if isinstance(v, list) else (dict(v) if isinstance(v, dict) else v))
        self.log_path = None
        # dataset trees (reference config.py:24-33): used when the directory exists, else the trainer falls back to synthetic clips
        if dataset_name == "volleyball":
            self.data_path = "data/volleyball/videos"
            self.test_seqs = [4, 5, 9, 11, 14, 20, 21, 25, 29, 34, 35, 37, 43, 44, 45, 47]
            self.train_seqs = [1, 3, 6, 7, 10, 13, 15, 16, 18, 22, 23, 31, 32, 36, 38, 39, 40, 41, 42, 48, 50, 52, 53, 54,
                               0, 2, 8, 12, 17, 19, 24, 26, 27, 28, 30, 33, 46, 49, 51]      # (the reference's order: it fixes the sample order)
        else:
            self.data_path = "data/collective"
            self.test_seqs = [5, 6, 7, 8, 9, 10, 11, 15, 16, 25, 28, 29]
            self.train_seqs = [s for s in range(1, 45) if s not in self.test_seqs]

    def init_config(self, need_new_folder=True):
        if self.exp_name is None:
            stamp = time.strftime("%Y-%m-%d_%H-%M-%S", time.localtime())
            self.exp_name = "[%s_stage%d]<%s>" % (self.exp_note, self.training_stage, stamp)
        self.result_path = "result/%s" % self.exp_name
        self.log_path = "result/%s/log.txt" % self.exp_name
        if need_new_folder:
            os.makedirs(self.result_path, exist_ok=True)
