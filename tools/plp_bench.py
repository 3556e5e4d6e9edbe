"""Throughput of the PLP path (Bark front end through the MFCC pass + ssp_plp_post) on device-resident audio."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n = 48000
ctx = api.Context.for_torch(0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
audio = (0.1 * torch.randn(n_utt * n, generator=g, device='cuda')).float()
plan = api.MfccPlan(ctx, pkg.preset_sidekit_plp())
seg = api.Segments.from_lengths(ctx, np.full(n_utt, n, dtype=np.int64))
fseg = plan.frame_segments(seg)
logspec = torch.empty((fseg.total, plan.d_out), device='cuda')
for variant in (0, 1):
    try:
        plan.run(audio, seg, fseg, out=logspec, variant=variant)
        ms_f = min(plan.run(audio, seg, fseg, out=logspec, variant=variant, timing=True)[1] for _ in range(3))
        print("front end variant %d: %.2f ms" % (variant, ms_f))
    except Exception as e:
        print("front end variant %d: %s" % (variant, e))
plan.run(audio, seg, fseg, out=logspec)
ms_f = min(plan.run(audio, seg, fseg, out=logspec, timing=True)[1] for _ in range(3))
api.plp_post(ctx, logspec, fseg, 8000.0)
ms_b = min(api.plp_post(ctx, logspec, fseg, 8000.0, timing=True)[1] for _ in range(3))
ms_nr = min(api.plp_post(ctx, logspec, fseg, 8000.0, rasta=False, timing=True)[1] for _ in range(3))
F = fseg.total
print("PLP %d utt x 3 s: front (Bark log spectrum) %.2f ms + back %.2f ms (without RASTA %.2f) = %.2f ms -> %.3g frames/s"
      % (n_utt, ms_f, ms_b, ms_nr, ms_f + ms_b, F / (ms_f + ms_b) * 1e3))
