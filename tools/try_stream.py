"""Scratch check of the wave-stream MFCC kernel (variant 3) against the workgroup kernel (variant 2) and the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
from conftest import synth_audio
from oracle import ref_cpu as O

def run(tables, sigs, variant):
    ctx = api.default_context(torch_stream=False)
    plan = api.MfccPlan(ctx, tables)
    seg = api.Segments.from_lengths(ctx, [len(s) for s in sigs])
    fseg = plan.frame_segments(seg)
    flat = np.concatenate(sigs).astype(np.float32)
    out = plan.run(flat, seg, fseg, variant=variant)
    return [np.asarray(out[fseg.offsets[i]:fseg.offsets[i + 1]]) for i in range(len(sigs))]

lens = [48000, 16000, 400, 560, 720, 1040, 1044, 3000, 4800, 8000, 100004, 20000, 404, 880, 2960, 5200]
sigs = [synth_audio(i, n, 16000) for i, n in enumerate(lens)]
bad = 0
for do in (0, 1, 2):
    tables = pkg.preset_sidekit(delta_order=do)
    cfg, w, fb, dct = O.sidekit_tables(delta_order=do, cmvn=0)
    g3 = run(tables, sigs, 3)
    g2 = run(tables, sigs, 2)
    for u, s in enumerate(sigs):
        ref = O.mfcc_pipeline(s, cfg, w, fb, dct)
        if ref.size == 0:
            continue
        e3 = np.abs(g3[u] - ref).max() / max(1.0, np.abs(ref).max())
        e2 = np.abs(g2[u] - ref).max() / max(1.0, np.abs(ref).max())
        flag = "" if e3 < 1e-4 else "  <-- FAIL"
        if flag:
            bad += 1
            d = np.abs(g3[u] - ref)
            rows = np.where(d.max(axis=1) > 1e-4 * max(1.0, np.abs(ref).max()))[0]
            cols = np.where(d.max(axis=0) > 1e-4 * max(1.0, np.abs(ref).max()))[0]
            flag += " rows %s cols %s" % (rows[:12], cols[:12])
        print("delta %d utt %2d len %6d T %4d  err v3 %.2e  v2 %.2e%s" % (do, u, len(s), ref.shape[0], e3, e2, flag))
print("FAILS", bad)
