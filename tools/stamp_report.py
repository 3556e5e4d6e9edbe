"""Diagnostic (SSP_STAMP build only): per-phase cycle shares of the fused MFCC quad loop."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api, _lib
lib = _lib.load()
ctx = api.Context.for_torch(0)
n_utt, n = 20000, 48000
audio = (0.1 * torch.randn(n_utt * n, device="cuda")).float()
plan = api.MfccPlan(ctx, pkg.preset_sidekit(delta_order=2))
seg = api.Segments.from_lengths(ctx, [n] * n_utt)
fseg = plan.frame_segments(seg)
out = torch.empty((fseg.total, 39), device="cuda")
plan.run(audio, seg, fseg, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
lib.ssp_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.ssp_debug_stamps(buf, 1)
_, ms = plan.run(audio, seg, fseg, out=out, timing=True)
torch.cuda.synchronize()
lib.ssp_debug_stamps(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = ["preemph+window", "FFT1+twiddle", "transpose w+r", "FFT2", "exchange+split+P", "filterbank+log", "DCT+ceps", "loop exit", "barrier wait", "tail", "wait DMA", "stage reads", "DMA issue"]
waves = v[15] * 4
tot = v[:13].sum()
print("kernel ms %.3f, workgroups %d, cycles per wave %.0f" % (ms, v[15], tot / waves))
for i, nm in enumerate(names):
    print("%-28s %6.2f %%   %8.0f cycles/wave   %7.1f cycles/quad" % (nm, 100 * v[i] / tot, v[i] / waves, v[i] / waves / 18.75))
