"""Randomised cross-check of the fused MFCC kernels (workgroup kernel, and whatever auto mode picks: the wave-stream kernels) against
the generic kernel and the float64 oracle: random dialects (sidekit, in-repo, PLP front end, librosa, in-repo with frameSize 2048), hops,
delta orders, CMVN, ragged utterance lengths (including shorter than a frame).  Run on the GPU box."""
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
    dialect = rng.choice(["sidekit", "inrepo", "plp", "librosa", "inrepo2048"])
    order = int(rng.integers(0, 3))
    cmvn = int(rng.integers(0, 2))
    has_fast = True
    min_len = 1
    if dialect == "sidekit":
        tables = pkg.preset_sidekit(fs=16000, delta_order=order, cmvn=cmvn)
        cfg, w, fb, dct = O.sidekit_tables(delta_order=order, cmvn=cmvn)
        fs = 16000
    elif dialect == "inrepo":
        fs = int(rng.choice([8000, 16000]))
        step = int(rng.choice([100, 128, 160, 256, 90]))
        tables = pkg.preset_inrepo(fs, 512, step, delta_order=order, cmvn=cmvn)
        cfg, w, fb, dct = O.inrepo_tables(fs, 512, step)
        cfg["delta_order"], cfg["cmvn"] = order, cmvn
    elif dialect == "plp":
        order = cmvn = 0
        tables = pkg.preset_sidekit_plp()
        cfg, w, fb, dct = O.sidekit_plp_tables()
        has_fast = False
    elif dialect == "librosa":
        order = cmvn = 0
        fs = int(rng.choice([8000, 16000]))
        tables = pkg.preset_librosa(fs, 13)
        cfg, w, fb, dct = O.librosa_tables(fs, 13)
        has_fast = False
        min_len = 1025
    else:
        order = cmvn = 0
        step = int(rng.choice([512, 300, 1024]))
        tables = pkg.preset_inrepo(16000, 2048, step)
        cfg, w, fb, dct = O.inrepo_tables(16000, 2048, step)
        has_fast = False
    n_utt = int(rng.integers(1, 12))
    lens = [int(x) for x in rng.choice([1, 7, 159, 400, 401, 512, 513, 3000, 16000, 48123, 200000], n_utt)]
    lens = [max(l, min_len) for l in lens]
    if cmvn:
        lens = [max(l, 2000) for l in lens]   # a 1-frame utterance has std 0 in every column: covered by the unit tests
    sigs = [(0.3 * rng.standard_normal(l)).astype(np.float32) for l in lens]
    # digitally silent stretches (sidekit has no log floor: ln 0 = -inf, NaN cepstra, the deltas spread them +-2 / +-4 frames; the
    # scaling leaves them out of its statistics): every kernel must reproduce the oracle's non-finite pattern exactly
    if dialect == "sidekit" and rng.random() < 0.5:
        for x in sigs:
            for _ in range(int(rng.integers(0, 3))):
                if len(x) > 800:
                    a0 = int(rng.integers(0, len(x) - 400))
                    x[a0:a0 + int(rng.integers(400, 4000))] = 0.0
    # a NaN sample (a corrupt recording): every frame that holds it — and, through the deltas, its +-2 / +-4 neighbours — is NaN in the
    # reference's arithmetic, whatever the log floor; nothing else may be touched.  (Not with scaling and an inf: the library raises there.)
    if rng.random() < 0.3:
        x = sigs[int(rng.integers(0, len(sigs)))]
        if len(x) > 0:
            x[int(rng.integers(0, len(x)))] = np.nan
    if '-v' in sys.argv:
        print(case, dialect, order, cmvn, lens, flush=True)
    plan = api.MfccPlan(ctx, tables)
    seg = api.Segments.from_lengths(ctx, lens)
    fseg = plan.frame_segments(seg)
    flat = np.concatenate(sigs)
    gen = np.asarray(plan.run(flat, seg, fseg, variant=1))
    fast = np.asarray(plan.run(flat, seg, fseg, variant=2)) if has_fast else gen
    auto = np.asarray(plan.run(flat, seg, fseg, variant=0))
    for u, s in enumerate(sigs):
        with np.errstate(all='ignore'):
            ref = O.mfcc_pipeline(s, cfg, w, fb, dct)
        a, b, c = (v[fseg.offsets[u]:fseg.offsets[u + 1]] for v in (fast, gen, auto))
        assert a.shape == ref.shape, (case, u, a.shape, ref.shape)
        if ref.size == 0:
            continue
        fin = np.isfinite(ref)
        for nm, g in (("fast", a), ("generic", b), ("auto", c)):
            if not (np.isfinite(g) == fin).all():
                badrows = np.unique(np.nonzero(np.isfinite(g) != fin)[0])
                raise AssertionError((case, dialect, order, cmvn, u, lens[u], nm, "finite pattern", "rows", badrows[:10].tolist(),
                                      "ref non-finite rows", np.unique(np.nonzero(~fin)[0])[:12].tolist(),
                                      "got", np.unique(np.nonzero(~np.isfinite(g))[0])[:12].tolist(),
                                      "nan samples at", np.nonzero(np.isnan(s))[0][:4].tolist(), "zeros", int((s == 0).sum())))
        # (scaling over a handful of finite rows — a short utterance most of whose frames the injected silence / NaN took — divides by
        #  a standard deviation of two or three nearly equal numbers: the float32 cepstra's own rounding is amplified without bound.
        #  Those utterances are checked for their finite pattern only)
        if cmvn and fin.any() and fin.sum(axis=0).min() < 8:
            continue
        scale = max(1.0, np.abs(ref[fin]).max()) if fin.any() else 1.0
        for nm, g in (("fast", a), ("generic", b), ("auto", c)):
            err = np.abs(g[fin] - ref[fin]).max() / scale if fin.any() else 0.0
            worst = max(worst, err)
            assert err <= 1e-4, (case, dialect, order, cmvn, u, lens[u], nm, err)
print("fuzz OK: %d cases, worst relative error %.2e, %.1f s" % (n_cases, worst, time.time() - t_start))
