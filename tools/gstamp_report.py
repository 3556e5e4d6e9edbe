"""Diagnostic (SSP_GSTAMP build only): per-phase cycle shares of the generic MFCC kernel."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api, _lib
lib = _lib.load()
ctx = api.Context.for_torch(0)
lib.ssp_debug_gstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["stage: compute+LDS", "FFT", "split+P", "filterbank+log", "DCT", "prologue", "wg barrier", "tail", "stage: loads land", "-"]
for name, tables, n_utt, n in (("PLP front (Bark 21 x 257, identity DCT)", pkg.preset_sidekit_plp(), 20000, 48000),
                               ("sidekit MFCC 13-d on the generic kernel", pkg.preset_sidekit(), 20000, 48000),
                               ("librosa 8k", pkg.preset_librosa(8000, 13), 20000, 24000),
                               ("in-repo 16k 1024/512", pkg.preset_inrepo(16000, 1024, 512), 20000, 48000)):
    audio = (0.1 * torch.randn(n_utt * n, device="cuda")).float()
    plan = api.MfccPlan(ctx, tables)
    seg = api.Segments.from_lengths(ctx, [n] * n_utt)
    fseg = plan.frame_segments(seg)
    out = torch.empty((fseg.total, plan.d_out), device="cuda")
    plan.run(audio, seg, fseg, out=out, variant=1)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.ssp_debug_gstamps(buf, 1)
    _, ms = plan.run(audio, seg, fseg, out=out, timing=True, variant=1)
    torch.cuda.synchronize()
    lib.ssp_debug_gstamps(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    waves = v[15]
    tot = v[:10].sum()
    fpw = fseg.total / waves
    print("%s: kernel ms %.3f, waves %d, frames/wave %.1f, cycles per wave %.0f (100 MHz ticks)" % (name, ms, waves, fpw, tot / waves))
    for i, nm in enumerate(names):
        print("  %-18s %6.2f %%   %8.0f ticks/wave   %7.1f ticks/frame" % (nm, 100 * v[i] / tot, v[i] / waves, v[i] / waves / fpw))
