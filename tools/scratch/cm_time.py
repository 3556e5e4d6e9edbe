import sys, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
U, n = 100000, 48000
flat = (0.1 * torch.randn(U * n, device='cuda')).contiguous()
seg = api.Segments.from_lengths(ctx, np.full(U, n, dtype=np.int64))
for order in (1, 2):
    for cm in (0, 1):
        plan = api.MfccPlan(ctx, pkg.preset_sidekit(delta_order=order, cmvn=cm))
        fseg = plan.frame_segments(seg)
        out = torch.empty((fseg.total, plan.d_out), device='cuda')
        for v in (3, 2):
            plan.run(flat, seg, fseg, out=out, variant=v)
            ms = [plan.run(flat, seg, fseg, out=out, variant=v, timing=True)[1] for _ in range(4)]
            print("order", order, "cmvn", cm, "variant", v, "ms %.3f" % float(np.median(ms)))
