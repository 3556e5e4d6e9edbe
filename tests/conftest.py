import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# glibc reports heap corruption ("free(): invalid pointer", ...) on the controlling terminal unless told otherwise: keep such a
# message in the captured log next to the abort it explains
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def synth_audio(utt, n, fs, S=10):
    """SURVEY.md 8(d) synthetic-audio recipe (same as tests/golden/make_golden.py)."""
    rng = np.random.default_rng(1234 + utt)
    s = utt % S
    f0 = 90 + 3 * s
    t = np.arange(n) / fs
    x = 0.3 * sum(np.sin(2 * np.pi * h * f0 * t) / h for h in range(1, 6)) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t))
    x = x + 0.05 * rng.standard_normal(n)
    return np.clip(x, -1, 1).astype(np.float32)


@pytest.fixture(autouse=True)
def _poison_lds(request):
    """Before every GPU test the LDS of every CU is filled with NaNs (ssp_debug_poison_lds): a kernel that reads LDS it never wrote then
    fails its parity test every time, instead of only when the previous kernel happened to leave a NaN or an infinity behind."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    try:
        from speech_signal_processing_amd import api, _lib
        ctx = api.default_context(torch_stream=False)
        _lib.check(_lib.load().ssp_debug_poison_lds(ctx._h, 0x7FC00000))
    except Exception as e:  # pragma: no cover - no GPU / library: the test itself reports that
        print("LDS poison skipped:", repr(e))
    yield
