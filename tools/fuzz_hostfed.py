"""Randomised ragged batches through the HOST-FED paths of the MFCC pass (round 6) against the device-pointer path on the same samples:
the sliced copy-in / compute / copy-back pipeline of ssp_mfcc_run(SSP_HOST) with random slice sizes (SSP_HOST_SLICE_MB is read per
call), float32 and int16 input (ssp_mfcc_run_i16: device-side widening), pinned and pageable memory, every dialect and every kernel
variant the plan has.  The sliced path launches the batch's work table in utterance ranges with data pointers biased by each slice's
first offsets: any kernel that addressed samples or rows other than through the batch's absolute offsets would differ here.  Bits must
be EQUAL (the work table is the same on both sides).  Every case also feeds the GMM scorer (precision 0 / 2) and the cosine scorer from
host arrays (feed_rows: rows through the same ring, ahead of the kernels) and compares with the device-pointer path, bit for bit.
Run on the GPU box:
    python tools/fuzz_hostfed.py [seed] [cases]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(seed)
ctx = api.default_context()                      # the library's own stream: host pointers
tctx = api.default_context(torch_stream=True)    # torch's stream: device pointers
t_start = time.time()
n_sliced = 0
for case in range(n_cases):
    dialect = str(rng.choice(["sidekit", "sidekit", "sidekit_cmvn", "inrepo", "plp", "librosa", "sidekit13"]))
    min_len = 1
    if dialect == "sidekit":
        tables = pkg.preset_sidekit(delta_order=int(rng.integers(1, 3)))
    elif dialect == "sidekit13":
        tables = pkg.preset_sidekit(delta_order=0)
    elif dialect == "sidekit_cmvn":
        tables = pkg.preset_sidekit(delta_order=int(rng.integers(1, 3)), cmvn=1)
    elif dialect == "inrepo":
        tables = pkg.preset_inrepo(int(rng.choice([8000, 16000])), 512, int(rng.choice([160, 256])))
    elif dialect == "plp":
        tables = pkg.preset_sidekit_plp()
    else:
        tables = pkg.preset_librosa(int(rng.choice([8000, 16000])), 13)   # (two-pass top_db: the host path stages this one whole)
        min_len = 1025
    n_utt = int(rng.integers(40, 1500))
    kind = rng.random(n_utt)
    lens = np.where(kind < 0.05, rng.integers(min_len, min_len + 900, n_utt),
           np.where(kind < 0.97, rng.integers(4000, 60000, n_utt), rng.integers(200000, 1500000, n_utt)))
    if rng.random() < 0.25:
        lens[:] = int(rng.integers(8000, 60000))
    lens = [int(v) for v in lens]
    total = int(np.sum(lens))
    x16 = (rng.standard_normal(total) * float(10.0 ** rng.uniform(1.0, 3.8))).clip(-32768, 32767).astype(np.int16)
    if rng.random() < 0.3:   # a silent stretch (ln 0 = -inf: the walk kernel inside a slice)
        a = int(rng.integers(0, max(1, total - 5000)))
        x16[a:a + int(rng.integers(400, 5000))] = 0
    x32 = x16.astype(np.float32)
    slice_mb = int(rng.choice([1, 1, 2, 3, 5]))
    os.environ["SSP_HOST_SLICE_MB"] = str(slice_mb)
    sliced = total * 4 >= 2 * (slice_mb << 20) and n_utt >= 2 and tables.cfg.n_fft != 2048
    n_sliced += int(sliced)
    plan, tplan = api.MfccPlan(ctx, tables), api.MfccPlan(tctx, tables)
    seg, tseg = api.Segments.from_lengths(ctx, lens), api.Segments.from_lengths(tctx, lens)
    fseg, tfseg = plan.frame_segments(seg), tplan.frame_segments(tseg)
    xd = torch.from_numpy(x32).cuda()
    variants = [0] + ([int(rng.choice([1, 2, 3]))] if tables.cfg.n_fft == 512 and dialect != "plp" else [])
    for variant in variants:
        try:
            want = tplan.run(xd, tseg, tfseg, variant=variant).cpu().numpy()
        except NotImplementedError:
            continue   # (the plan has no instance of that kernel)
        pinned = rng.random() < 0.5
        for name, src in (("f32", x32), ("i16", x16)):
            if pinned:
                buf = api.pinned_empty(src.shape, src.dtype)
                buf[:] = src
                src = buf
            got = plan.run(src, seg, fseg, variant=variant)
            assert np.array_equal(got, want, equal_nan=True), (case, dialect, variant, name, "pinned" if pinned else "pageable", slice_mb, n_utt,
                                                               int(np.argmax((got != want) & ~(np.isnan(got) & np.isnan(want))) // got.shape[1]))
        # int16 through device pointers (widened slice by slice on the device)
        got = tplan.run(torch.from_numpy(x16).cuda(), tseg, tfseg, variant=variant).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True), (case, dialect, variant, "i16 device")
    # ---- the scorers fed from the host (feed_rows: rows copied in ahead of the kernels that score them) against the device-pointer path
    K, D, S = int(rng.choice([8, 32, 64])), int(rng.choice([13, 26, 39])), int(rng.integers(2, 20))
    w, mu, cov = rng.dirichlet(5 * np.ones(K)), rng.standard_normal((K, D)), rng.uniform(0.5, 2.0, (K, D))
    mus = np.stack([mu] + [mu + 0.3 * rng.standard_normal((K, D)) for _ in range(S)])
    n_g = int(rng.integers(500, 6000))
    glens = [int(v) for v in np.where(rng.random(n_g) < 0.03, 0, rng.integers(1, 400, n_g))]   # (3 % empty utterances)
    if rng.random() < 0.3 and glens:
        glens[int(rng.integers(0, len(glens)))] = int(rng.integers(20000, 60000))      # an utterance longer than a slice
    F = int(np.sum(glens))
    Xg = rng.standard_normal((F, D)).astype(np.float32)
    hsc = api.GmmScorer(ctx, np.stack([w] * (S + 1)), mus, np.stack([cov] * (S + 1)), has_ubm=True)
    tsc = api.GmmScorer(tctx, np.stack([w] * (S + 1)), mus, np.stack([cov] * (S + 1)), has_ubm=True)
    hs, ts = api.Segments.from_lengths(ctx, glens), api.Segments.from_lengths(tctx, glens)
    nz = np.asarray(glens) > 0
    Xgd = torch.from_numpy(Xg).cuda()
    for prec in (0, 2):
        want, got = tsc.score(Xgd, ts, precision=prec), hsc.score(Xg, hs, precision=prec)
        assert np.array_equal(got["scores"][nz], want["scores"].cpu().numpy()[nz], equal_nan=True), (case, "gmm host-fed scores", prec, K, D, S, len(glens))
        assert np.array_equal(got["argmax"][nz], want["argmax"].cpu().numpy()[nz]), (case, "gmm host-fed arg-max", prec)
    dd, Sc = int(rng.choice([64, 128, 200, 256])), int(rng.integers(2, 1500))
    Nc = int(rng.integers(20000, 120000))
    Cn = rng.standard_normal((Sc, dd)).astype(np.float32)
    Xc = (Cn[rng.integers(0, Sc, Nc)] + float(10.0 ** rng.uniform(-1, 1)) * rng.standard_normal((Nc, dd))).astype(np.float32)
    want, got = api.cosine_identify(tctx, torch.from_numpy(Xc).cuda(), torch.from_numpy(Cn).cuda()), api.cosine_identify(ctx, Xc, Cn)
    assert np.array_equal(got["argmin"], want["argmin"].cpu().numpy()) and np.array_equal(got["min"], want["min"].cpu().numpy(), equal_nan=True), (case, "cosine host-fed", Nc, Sc, dd)
    print("case %d: %s, %d utterances / %.1f MB, slice %d MiB (%s), variants %s; gmm %d utt / %.1f MB K %d D %d S %d; cosine %d x %d x %d (%.1f MB)" % (
        case, dialect, n_utt, total * 4 / 1e6, slice_mb, "sliced" if sliced else "whole", variants, len(glens), Xg.nbytes / 1e6, K, D, S, Nc, Sc, dd, Xc.nbytes / 1e6), flush=True)
print("fuzz_hostfed OK: %d cases (%d through the sliced pipeline), %.1f s" % (n_cases, n_sliced, time.time() - t_start))
