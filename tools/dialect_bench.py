"""Throughput of the other MFCC dialects on the same 100k-utterance batch shape (device-resident audio)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
for name, tables, n_utt, n in (("in-repo 8 kHz 512/256 (utils.processing.MFCC, 13-d)", pkg.preset_inrepo(8000, 512, 256), 100000, 24000),
                               ("in-repo 16 kHz 512/256", pkg.preset_inrepo(16000, 512, 256), 100000, 48000),
                               ("sidekit 26-d + CMVN (GMM_UBM.extract_feature)", pkg.preset_sidekit(delta_order=1, cmvn=1), 100000, 48000),
                               ("in-repo 16 kHz 1024/512 (generic kernel)", pkg.preset_inrepo(16000, 1024, 512), 20000, 48000),
                               ("in-repo 8 kHz 256/128 (generic kernel)", pkg.preset_inrepo(8000, 256, 128), 20000, 24000),
                               ("librosa 8 kHz 2048/512 (MFCC_lib, generic kernel)", pkg.preset_librosa(8000, 13), 20000, 24000),
                               ("librosa 16 kHz 2048/512, 10 s utterances (two-pass top_db)", pkg.preset_librosa(16000, 13), 4000, 160000)):
    audio = (0.1 * torch.randn(n_utt * n, generator=g, device='cuda')).float()
    plan = api.MfccPlan(ctx, tables)
    seg = api.Segments.from_lengths(ctx, np.full(n_utt, n, dtype=np.int64))
    fseg = plan.frame_segments(seg)
    out = torch.empty((fseg.total, plan.d_out), device='cuda')
    plan.run(audio, seg, fseg, out=out)
    ms = min(plan.run(audio, seg, fseg, out=out, timing=True)[1] for _ in range(3))
    gb = (audio.numel() * 4 + out.numel() * 4) / 1e9
    print("%-58s %8.2f ms  %.3g frames/s  %.0f GB/s" % (name, ms, fseg.total / ms * 1e3, gb / ms * 1e3))
    del audio, out
