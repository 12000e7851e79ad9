"""Zero-edit drop-in: a top-level module with the reference's name that re-exports the MI355X implementation.

Put this directory FIRST on the module search path (`PYTHONPATH=<repo>/dropin python scripts/train_volleyball_stage2_dynamic.py` from the
reference tree: the reference's launchers do `sys.path.append(".")`, which comes AFTER PYTHONPATH) and the reference's own import lines --
`from train_net_dynamic import *` (third-party roi_align.roi_align (reference infer_model.py:3)) -- resolve here; nothing of the reference is copied or edited.
"""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
_ROOT = _os.path.dirname(_ROOT)
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
from din_amd.roi_align.roi_align import *     # noqa: E402,F401,F403
from din_amd.roi_align.roi_align import RoIAlign   # noqa: E402,F401
