"""Soak: many launches of the persistent MFCC kernels on ragged batches; every launch must reproduce the first bit for bit.
   python tools/soak.py ["sidekit 39-d"] ["sidekit 26-d + scaling"] ["librosa 2048 / 512"] ["in-repo 512 / 256"]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
rng = np.random.default_rng(0)
presets = {"sidekit 39-d": lambda: pkg.preset_sidekit(delta_order=2), "sidekit 26-d + scaling": lambda: pkg.preset_sidekit(delta_order=1, cmvn=1),
           "librosa 2048 / 512": lambda: pkg.preset_librosa(8000, 13), "in-repo 512 / 256": lambda: pkg.preset_inrepo(16000, 512, 256)}
which = sys.argv[1:] or ["sidekit 39-d"]
for tag, lens, name in [(t, l, nm) for nm in which for t, l in (("uniform 3 s", np.full(20000, 48000)), ("ragged 0.05-20 s", rng.integers(800, 320000, 6000)), ("tiny", rng.integers(1, 2000, 50000)))]:
    if name.startswith("librosa"):  # (centred frames with reflect padding need more than half a window of samples)
        lens = rng.integers(1100, 6000, 20000) if tag == "tiny" else np.maximum(lens, 1100)
    tag = name + ", " + tag
    lens = np.asarray(lens, dtype=np.int64)
    audio = (0.1 * torch.randn(int(lens.sum()), device='cuda')).float()
    plan = api.MfccPlan(ctx, presets[name]())
    seg = api.Segments.from_lengths(ctx, lens)
    fseg = plan.frame_segments(seg)
    ref = plan.run(audio, seg, fseg).clone()
    bad = 0
    for it in range(150):
        out = plan.run(audio, seg, fseg)
        if not bool((out == ref).all() | (out.isnan() & ref.isnan()).all()):
            bad += int(((out != ref) & ~(out.isnan() & ref.isnan())).any())
    torch.cuda.synchronize()
    print("%-44s %d utterances, %d frames: 150 launches, %d differing" % (tag, len(lens), fseg.total, bad))
    assert bad == 0
print("soak OK")
