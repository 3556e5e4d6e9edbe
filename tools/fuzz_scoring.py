"""Randomised shapes for the scoring / training / matching kernels against the float64 oracle.  Run on the GPU box."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
from oracle import ref_cpu as O

ctx = api.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
verbose = '-v' in sys.argv
t_start = time.time()
for case in range(n_cases):
    # ---- GMM scoring
    K = int(rng.choice([1, 2, 5, 16, 31, 32, 33, 64, 100, 257]))
    D = int(rng.choice([1, 2, 7, 13, 26, 39, 40, 64, 65, 100]))
    M = int(rng.integers(1, 6))
    has_ubm = bool(rng.integers(0, 2)) and M >= 2
    lens = [int(x) for x in rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, 1000], int(rng.integers(1, 7)))]
    if verbose:
        print(case, "gmm", K, D, M, has_ubm, lens, flush=True)
    w = rng.dirichlet(5 * np.ones(K), M)
    mu = rng.standard_normal((M, K, D))
    cov = rng.uniform(0.5, 2.0, (M, K, D))
    X = rng.standard_normal((sum(lens), D)).astype(np.float32)
    seg = api.Segments.from_lengths(ctx, lens)
    sc = api.GmmScorer(ctx, w, mu, cov, has_ubm=has_ubm)
    for prec in ((0, 1) if D <= 64 else (0,)):
        r = sc.score(X, seg, loglik=True, scores=True, argmax=True, precision=prec)
        ll = np.asarray(r["loglik"])
        for m in range(M):
            ref = O.gmm_score_samples(w[m], mu[m], cov[m], X) if len(X) else np.zeros(0)
            assert np.allclose(ll[m], ref, rtol=2e-4, atol=2e-4), (case, "loglik", K, D, m, prec, np.abs(ll[m] - ref).max())
        off = np.concatenate([[0], np.cumsum(lens)])
        for u, L in enumerate(lens):
            if L == 0:
                continue
            refs = np.array([O.gmm_score(w[m], mu[m], cov[m], X[off[u]:off[u + 1]]) for m in range(M)])
            assert np.allclose(np.asarray(r["scores"])[u], refs, rtol=2e-4, atol=2e-4), (case, "score", u, prec)
    # ---- the split-precision modes keep the fp32 path's arg-max (proven band: precision 1; calibrated band: 3), speakers a hair apart
    if D <= 64 and M - int(has_ubm) >= 2 and sum(lens) > 0:
        mu2 = mu.copy()
        for m in range(int(has_ubm) + 1, M):
            mu2[m] = mu2[int(has_ubm)] * (1.0 + 10.0 ** rng.uniform(-7, -2) * rng.standard_normal((K, D)))
        sc2 = api.GmmScorer(ctx, np.repeat(w[:1], M, 0), mu2, np.repeat(cov[:1], M, 0), has_ubm=has_ubm)
        a0 = np.asarray(sc2.score(X, seg, precision=0)["argmax"])
        for prec in (1, 3):
            a1 = np.asarray(sc2.score(X, seg, precision=prec)["argmax"])
            nz = np.array(lens) > 0
            assert np.array_equal(a0[nz], a1[nz]), (case, "gmm split-precision arg-max", prec, K, D, M)
    # ---- cosine
    N, S, d = int(rng.choice([1, 31, 32, 33, 500])), int(rng.choice([1, 2, 31, 32, 33, 129, 300])), int(rng.choice([1, 3, 64, 128, 255, 256, 257, 512]))
    if verbose:
        print(case, "cos", N, S, d, flush=True)
    Cn = rng.standard_normal((S, d)).astype(np.float32)
    Xc = rng.standard_normal((N, d)).astype(np.float32)
    rc = api.cosine_identify(ctx, Xc, Cn, dist=True)
    refd = O.cosine_matrix(Xc, Cn)
    assert np.abs(np.asarray(rc["dist"]) - refd).max() < 2e-5, (case, "cos", N, S, d)
    # arg-min: exact unless the two best are within float noise
    am = np.asarray(rc["argmin"])
    srt = np.sort(refd, axis=1)
    clear = (srt[:, 1] - srt[:, 0] > 1e-5) if S > 1 else np.ones(N, bool)
    assert (am[clear] == refd.argmin(1)[clear]).all(), (case, "argmin")
    if d <= 256:  # split precision (bf16x3; the cascade with a bf16 sweep in front): the fp32 path's arg-min on every row, close calls included
        Cc = Cn.copy()
        if S >= 2:
            Cc[1::2] = Cc[0::2][: len(Cc[1::2])] * (1.0 + 10.0 ** rng.uniform(-8, -2) * rng.standard_normal(Cc[1::2].shape).astype(np.float32))
        a0 = np.asarray(api.cosine_identify(ctx, Xc, Cc)["argmin"])
        for prec in (1, 2):
            assert np.array_equal(a0, np.asarray(api.cosine_identify(ctx, Xc, Cc, precision=prec)["argmin"])), (case, "cosine split-precision arg-min", prec, N, S, d)
    # ---- dense
    Nn, di, un = int(rng.choice([1, 5, 127, 128, 129, 300])), int(rng.choice([1, 2, 31, 32, 33, 100, 1274])), int(rng.choice([1, 3, 127, 128, 129, 256]))
    if verbose:
        print(case, "dense", Nn, di, un, flush=True)
    Xd = rng.standard_normal((Nn, di)).astype(np.float32)
    Wd = (rng.standard_normal((di, un)) / np.sqrt(di)).astype(np.float32)
    bd = rng.standard_normal(un).astype(np.float32)
    yd = api.dense_forward(ctx, Xd, np.ascontiguousarray(Wd.T), bd, relu=bool(rng.integers(0, 2)) and False)
    assert np.abs(yd - O.dense_net_forward(Xd, [(Wd, bd, 'linear')])).max() < 1e-4 * max(1.0, np.abs(yd).max()), (case, "dense")
    # ---- EM statistics
    Ke, De, ne = int(rng.choice([1, 3, 64, 65, 130])), int(rng.choice([1, 5, 13, 39, 47, 48, 64])), int(rng.choice([1, 63, 64, 65, 1000, 5000]))
    if verbose:
        print(case, "em", Ke, De, ne, flush=True)
    we = rng.dirichlet(5 * np.ones(Ke)); mue = rng.standard_normal((Ke, De)); cve = rng.uniform(0.5, 2.0, (Ke, De))
    Xe = rng.standard_normal((ne, De)).astype(np.float32)
    st = api.gmm_em_stats(ctx, we, mue, cve, Xe)
    nk, sx, sxx, ll = O.gmm_em_stats(we, mue, cve, Xe)
    assert abs(st["loglik_sum"] - ll) <= 2e-5 * max(1.0, abs(ll)), (case, "em ll", Ke, De, ne)
    assert np.allclose(st["nk"], nk, rtol=2e-4, atol=2e-4 * max(1.0, nk.max())), (case, "em nk")
    assert np.allclose(st["sx"], sx, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(sx).max())), (case, "em sx")
    assert np.allclose(st["sxx"], sxx, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(sxx).max())), (case, "em sxx")
    # ---- DTW
    dim = int(rng.choice([1, 1, 2, 13]))
    ql = [int(x) for x in rng.choice([1, 2, 63, 64, 65, 200, 700], int(rng.integers(1, 4)))]
    tl = [int(x) for x in rng.choice([1, 3, 255, 256, 257, 300, 1100, 2100], int(rng.integers(1, 4)))]
    if verbose:
        print(case, "dtw", dim, ql, tl, flush=True)
    mk = lambda L: (rng.standard_normal((L, dim)) if dim > 1 else rng.standard_normal(L)).astype(np.float32)
    Q, T = [mk(L) for L in ql], [mk(L) for L in tl]
    got = api.dtw_distances(ctx, Q, T)
    ref = np.array([[O.dtw_distance(q, t) for t in T] for q in Q])
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-4), (case, "dtw", dim, ql, tl)
print("fuzz_scoring OK: %d cases, %.1f s" % (n_cases, time.time() - t_start))
