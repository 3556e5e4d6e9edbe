"""Soak: many launches of the persistent MFCC kernel on ragged batches; every launch must reproduce the first bit for bit."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
rng = np.random.default_rng(0)
for tag, lens in (("uniform 3 s", np.full(20000, 48000)), ("ragged 0.05-20 s", rng.integers(800, 320000, 6000)), ("tiny", rng.integers(1, 2000, 50000))):
    lens = np.asarray(lens, dtype=np.int64)
    audio = (0.1 * torch.randn(int(lens.sum()), device='cuda')).float()
    plan = api.MfccPlan(ctx, pkg.preset_sidekit(delta_order=2))
    seg = api.Segments.from_lengths(ctx, lens)
    fseg = plan.frame_segments(seg)
    ref = plan.run(audio, seg, fseg).clone()
    bad = 0
    for it in range(150):
        out = plan.run(audio, seg, fseg)
        if not bool((out == ref).all() | (out.isnan() & ref.isnan()).all()):
            bad += int(((out != ref) & ~(out.isnan() & ref.isnan())).any())
    torch.cuda.synchronize()
    print("%-18s %d utterances, %d frames: 150 launches, %d differing" % (tag, len(lens), fseg.total, bad))
    assert bad == 0
print("soak OK")
