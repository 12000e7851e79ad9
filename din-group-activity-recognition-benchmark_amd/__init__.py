"""din_amd -- MI355X-native DIN stage-2 hot path (package directory: din-group-activity-recognition-benchmark_amd/).

Import as `din_amd` (repo-root alias package) -- the hyphenated directory name is not a Python identifier.
Submodules mirror the reference layout: infer_model, infer_module.dynamic_infer_module, backbone.backbone,
roi_align.roi_align, utils, config, train_net_dynamic (+ parallel, optim, ops, nhwc, _lib).
"""
from . import _lib  # noqa: F401  (ctypes binding; loading the .so is deferred to first use so CPU-only hosts can import)

__all__ = ["_lib"]
__version__ = "0.1.0"
