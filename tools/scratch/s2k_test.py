import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
from test_gpu_parity import _run_plan, synth_audio
for name, tables, fs in (("librosa 8k", pkg.preset_librosa(8000, 13), 8000), ("librosa 16k", pkg.preset_librosa(16000, 13), 16000), ("inrepo 2048/512", pkg.preset_inrepo(16000, 2048, 512), 16000)):
    for tag, lens in (("mixed", [24000, 16037, 2049, 3000, 4801, 100003, 1025 + 7, 40000]), ("single-chunk", [24000, 16037, 2049, 3000, 4801, 1025 + 7, 40000, 65000])):
        sigs = [synth_audio(u, n, fs) for u, n in enumerate(lens)]
        g4, fseg = _run_plan(api, tables, sigs, variant=4)
        g1, _ = _run_plan(api, tables, sigs, variant=1)
        worst = max(float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) for a, b in zip(g4, g1))
        print(name, tag, "max frames", int(max(np.diff(fseg.offsets))), "worst", worst, "finite", all(bool(np.isfinite(a).all()) for a in g4))
