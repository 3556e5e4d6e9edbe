"""Soak of the 2048-point wave-stream kernel over ragged batch shapes (single-chunk / multi-chunk utterances mixed, hops 512 / 1024 / 300,
clean and with a NaN sample): every call must RETURN (run under `timeout`) and agree with the generic kernel's finite pattern.
   timeout -k 10 300 python tools/soak_s2k.py [seed] [cases]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
ctx = api.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time()
for case in range(n_cases):
    kind = rng.choice(["librosa16", "librosa8", "inrepo512", "inrepo1024", "inrepo300"])
    tables = {"librosa16": lambda: pkg.preset_librosa(16000, 13), "librosa8": lambda: pkg.preset_librosa(8000, 13),
              "inrepo512": lambda: pkg.preset_inrepo(16000, 2048, 512), "inrepo1024": lambda: pkg.preset_inrepo(16000, 2048, 1024),
              "inrepo300": lambda: pkg.preset_inrepo(16000, 2048, 300)}[kind]()
    n_utt = int(rng.integers(1, 40))
    lens = [int(x) for x in rng.choice([1025, 1026, 2048, 3000, 16000, 24000, 48123, 66000, 200000, 300001], n_utt)]
    sigs = [(0.3 * rng.standard_normal(l)).astype(np.float32) for l in lens]
    if rng.random() < 0.4:
        x = sigs[int(rng.integers(0, n_utt))]
        x[int(rng.integers(0, len(x)))] = np.nan
    plan = api.MfccPlan(ctx, tables)
    if rng.random() < 0.3:
        plan.set_reproducible(True)
    seg = api.Segments.from_lengths(ctx, lens)
    fseg = plan.frame_segments(seg)
    flat = np.concatenate(sigs)
    gen = np.asarray(plan.run(flat, seg, fseg, variant=1))
    s2k = np.asarray(plan.run(flat, seg, fseg, variant=4))
    assert (np.isfinite(gen) == np.isfinite(s2k)).all(), (case, kind, lens)
    fin = np.isfinite(gen)
    if fin.any():
        assert np.abs(gen[fin] - s2k[fin]).max() <= 2e-4 * max(1.0, float(np.abs(gen[fin]).max())), (case, kind, lens)
    print(case, kind, n_utt, "ok", flush=True)
print("soak_s2k OK: %d cases, %.1f s" % (n_cases, time.time() - t0))
