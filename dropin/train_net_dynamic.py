"""Zero-edit drop-in: a top-level module with the reference's name that re-exports the MI355X implementation.

Put this directory FIRST on the module search path (`PYTHONPATH=<repo>/dropin python scripts/train_volleyball_stage2_dynamic.py` from the
reference tree: the reference's launchers do `sys.path.append(".")`, which comes AFTER PYTHONPATH) and the reference's own import lines --
`from train_net_dynamic import *` (reference scripts/train_volleyball_stage2_dynamic.py:1-5, train_net_dynamic.py:9-15) -- resolve here; nothing of the reference is copied or edited.
"""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)
# the reference module star-imports its siblings (train_net_dynamic.py:9-15), so `from train_net_dynamic import *` also brings Config etc.
from din_amd.config import *            # noqa: E402,F401,F403
from din_amd.volleyball import *        # noqa: E402,F401,F403
from din_amd.collective import *        # noqa: E402,F401,F403
from din_amd.dataset import *           # noqa: E402,F401,F403
from din_amd.infer_model import *       # noqa: E402,F401,F403
from din_amd.utils import *             # noqa: E402,F401,F403
from din_amd.train_net_dynamic import *  # noqa: E402,F401,F403
from din_amd.train_net_dynamic import train_net, train_volleyball, test_volleyball, train_collective, test_collective, set_bn_eval, adjust_lr   # noqa: E402,F401
