"""Randomised cross-check of the fused MFCC kernel against the generic kernel and the float64 oracle: random hops, window lengths,
filter counts, delta orders, CMVN, ragged utterance lengths (including shorter than a frame).  Run on the GPU box."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api, frontend as F
from oracle import ref_cpu as O

ctx = api.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
t_start = time.time()
for case in range(n_cases):
    dialect = rng.choice(["sidekit", "inrepo"])
    order = int(rng.integers(0, 3))
    cmvn = int(rng.integers(0, 2))
    if dialect == "sidekit":
        tables = pkg.preset_sidekit(fs=16000, delta_order=order, cmvn=cmvn)
        cfg, w, fb, dct = O.sidekit_tables(delta_order=order, cmvn=cmvn)
        fs = 16000
    else:
        fs = int(rng.choice([8000, 16000]))
        step = int(rng.choice([100, 128, 160, 256]))
        tables = pkg.preset_inrepo(fs, 512, step, delta_order=order, cmvn=cmvn)
        cfg, w, fb, dct = O.inrepo_tables(fs, 512, step)
        cfg["delta_order"], cfg["cmvn"] = order, cmvn
    n_utt = int(rng.integers(1, 12))
    lens = [int(x) for x in rng.choice([1, 7, 159, 400, 401, 512, 513, 3000, 16000, 48123, 200000], n_utt)]
    if cmvn:
        lens = [max(l, 2000) for l in lens]   # a 1-frame utterance has std 0 in every column: covered by the unit tests
    sigs = [(0.3 * rng.standard_normal(l)).astype(np.float32) for l in lens]
    if '-v' in sys.argv:
        print(case, dialect, order, cmvn, lens, flush=True)
    plan = api.MfccPlan(ctx, tables)
    seg = api.Segments.from_lengths(ctx, lens)
    fseg = plan.frame_segments(seg)
    flat = np.concatenate(sigs)
    fast = np.asarray(plan.run(flat, seg, fseg, variant=2))
    gen = np.asarray(plan.run(flat, seg, fseg, variant=1))
    for u, s in enumerate(sigs):
        ref = O.mfcc_pipeline(s, cfg, w, fb, dct)
        a, b = fast[fseg.offsets[u]:fseg.offsets[u + 1]], gen[fseg.offsets[u]:fseg.offsets[u + 1]]
        assert a.shape == ref.shape, (case, u, a.shape, ref.shape)
        if ref.size == 0:
            continue
        fin = np.isfinite(ref)
        assert (np.isfinite(a) == fin).all() and (np.isfinite(b) == fin).all(), (case, u, "finite pattern")
        scale = max(1.0, np.abs(ref[fin]).max()) if fin.any() else 1.0
        for nm, g in (("fast", a), ("generic", b)):
            err = np.abs(g[fin] - ref[fin]).max() / scale if fin.any() else 0.0
            worst = max(worst, err)
            assert err <= 1e-4, (case, dialect, order, cmvn, u, lens[u], nm, err)
print("fuzz OK: %d cases, worst relative error %.2e, %.1f s" % (n_cases, worst, time.time() - t_start))
