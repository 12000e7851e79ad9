import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def monkeypatch(monkeypatch):
    """Kernel-selection switches go to the library THROUGH THE C ABI: `monkeypatch.setenv("DIN_...", v)` in a GPU test also calls
    din_set_option("DIN_...", v) (include/din_hip.h) and the option is restored when the test ends.  libdin_hip.so itself never reads the
    environment; the environment half of the call remains for the few switches the Python host layer owns
    (DIN_POOL_COMMUTE, DIN_ROI_COMPOSE, DIN_FUSE_*)."""
    from din_amd import _lib
    setenv, touched = monkeypatch.setenv, {}

    def setenv_and_option(name, value, prepend=None):
        setenv(name, value, prepend)
        if name.startswith("DIN_") and os.path.exists(_lib.LIB_PATH):
            if name not in touched:
                touched[name] = _lib.get_option(name)
            _lib.set_option(name, value)

    monkeypatch.setenv = setenv_and_option
    yield monkeypatch
    for name, old in touched.items():
        _lib.set_option(name, old)


# ---- tolerance margins ---------------------------------------------------------------------------------------------------------------
# Every error figure / cosine the GPU tests compare against a tolerance is a Measured: comparing it records (test, line, value, bound),
# and the session writes the worst margin per assert to gpurun_out/test_margins.txt (copied to profiles/rNN_test_margins.txt).  An
# assert that passes with less than 2x margin is a flake waiting for a different box: VERDICT r3, item 1(c).
_MARGINS = {}


class Measured(float):
    """a float that remembers what it was compared with.  kind 'err': bounds are upper bounds (margin = bound / value);
    kind 'cos': bounds are lower bounds on a cosine (margin = (1 - bound) / (1 - value))."""

    def __new__(cls, value, kind="err"):
        self = super().__new__(cls, value)
        self.kind = kind
        return self

    def _note(self, bound):
        if isinstance(bound, Measured):                       # ranking two measurements against each other is not a tolerance
            return
        try:
            bound = float(bound)
        except (TypeError, ValueError):
            return
        if bound == 0.0:                                      # max(0.0, measured): a running maximum, not a tolerance
            return
        f = sys._getframe(2)
        where = f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}"
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::", 1)[-1]
        v = float(self)
        if self.kind == "cos":
            margin = (1.0 - bound) / max(1.0 - v, 1e-300)
        else:
            margin = bound / max(v, 1e-300)
        key = (test, where)
        old = _MARGINS.get(key)
        if old is None or margin < old[0]:
            _MARGINS[key] = (margin, v, bound, self.kind)

    def __le__(self, o):
        self._note(o)
        return float(self) <= (float(o) if isinstance(o, Measured) else o)

    def __lt__(self, o):
        self._note(o)
        return float(self) < (float(o) if isinstance(o, Measured) else o)

    def __ge__(self, o):
        self._note(o)
        return float(self) >= (float(o) if isinstance(o, Measured) else o)

    def __gt__(self, o):
        self._note(o)
        return float(self) > (float(o) if isinstance(o, Measured) else o)

    __hash__ = float.__hash__


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rows = sorted(_MARGINS.items(), key=lambda kv: kv[1][0])
    tight = [r for r in rows if r[1][0] < 2.0]
    with open(os.path.join(out, "test_margins.txt"), "w") as fh:
        fh.write(f"# worst margin per tolerance assert of this session ({len(rows)} asserts; {len(tight)} below 2x).  margin = allowed / worst "
                 "(error bounds) or (1 - allowed) / (1 - worst) (cosine floors); exit status %d\n" % int(exitstatus))
        fh.write(f"# {'margin':>10} {'worst':>12} {'allowed':>12} kind  where  test\n")
        for (test, where), (m, v, b, kind) in rows:
            fh.write(f"{m:12.3g} {v:12.4e} {b:12.4e} {kind:4s}  {where}  {test}\n")


# ---- order of the GPU suite ----------------------------------------------------------------------------------------------------------
# The driver runs `pytest -m gpu -x`: whatever fails first hides everything after it (round 3: test 27 of 229).  So the deterministic
# evidence goes first -- bit-exact integer / index tests, then the kernel-level tests, then whole-model goldens in eval mode -- and the
# tests that cross dozens of order-sensitive fp32 layers in train mode (batch statistics, trainer steps) go last.  Stable within a rank.
_FIRST = ("bit_exact", "boxes_frame_index", "boxes_idx", "prep_images", "exports_every")
_LAST = ("batch_statistics", "train_net", "trainer", "captured_step", "stress")


def _rank(item):
    path, name = str(item.fspath), item.name
    if "test_gpu_kernels" in path:
        f = 0
    elif "test_gpu_din_model" in path:
        f = 2
    else:
        f = 1
    if any(k in name for k in _FIRST):
        w = 0
    elif any(k in name for k in _LAST):
        w = 2
    else:
        w = 1
    if w == 2:
        f = 3                                                 # train-mode whole-model tests after everything else
    return (f, w)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_rank)
