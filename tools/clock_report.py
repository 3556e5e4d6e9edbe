"""Diagnostic (SSP_S_CLOCK build): shader clock the chip holds while the wave-stream MFCC kernel runs."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api, _lib
lib = _lib.load()
ctx = api.Context.for_torch(0)
n_utt, n = 100000, 48000
audio = (0.1 * torch.randn(n_utt * n, device="cuda")).float()
plan = api.MfccPlan(ctx, pkg.preset_sidekit(delta_order=2))
seg = api.Segments.from_lengths(ctx, [n] * n_utt)
fseg = plan.frame_segments(seg)
out = torch.empty((fseg.total, 39), device="cuda")
for _ in range(20):
    plan.run(audio, seg, fseg, out=out, variant=3)
_, ms = plan.run(audio, seg, fseg, out=out, variant=3, timing=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 2)()
lib.ssp_debug_clock.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.ssp_debug_clock(plan._h, buf)
print("kernel %.3f ms; shader cycles / realtime ticks = %.3f -> %.3f GHz; mean workgroup life %.3f ms" % (ms, buf[0] / buf[1], buf[0] / buf[1] * 0.1, buf[1] / 768 / 1e5))
