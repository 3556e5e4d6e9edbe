import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
from conftest import synth_audio
from test_gpu_parity import _run_plan
fs = 8000
tables = pkg.preset_librosa(fs, 13)
sigs = [synth_audio(u, n, fs) for u, n in enumerate([24000, 16037, 2049, 3000, 4801, 100003, 1025 + 7, 40000, 2048 * 40])]
runs = {}
for name, v in (("v4a", 4), ("v4b", 4), ("v0", 0), ("v1", 1)):
    runs[name], fseg = _run_plan(api, tables, sigs, variant=v)
for u in range(len(sigs)):
    d = {k: np.abs(runs[k][u] - runs["v1"][u]).max(axis=1) for k in ("v4a", "v4b", "v0")}
    bad = {k: np.nonzero(v > 1e-3)[0][:12].tolist() for k, v in d.items()}
    print(u, runs["v1"][u].shape, {k: float(v.max()) for k, v in d.items()}, bad, "v4a==v4b", np.array_equal(runs["v4a"][u], runs["v4b"][u]))
