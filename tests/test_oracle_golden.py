"""The oracle (oracle/ref_cpu.py) pinned against outputs of the reference's own code
(tests/golden/*.npz, made by tests/golden/make_golden.py which imports /root/reference)."""
import numpy as np
import pytest

from oracle import ref_cpu as O

SIGNALS = ["noise", "tone", "silence", "ragged", "short", "int16", "utt3s16k", "one_step"]
GEOMS = [(8000, 512, 256), (16000, 512, 256), (16000, 256, 128), (8000, 1024, 512)]


@pytest.mark.parametrize("name", SIGNALS)
def test_mfcc_inrepo_matches_reference(golden, name):
    g = golden("mfcc_inrepo")
    x = g[f"x_{name}"]
    for fs, L, st in GEOMS:
        ref = g[f"mfcc_{name}_{fs}_{L}_{st}"]
        got = O.MFCC(x, fs=fs, frameSize=L, step=st)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11)
    np.testing.assert_allclose(O.MFCC_flat(x), g[f"flat_{name}"], rtol=0, atol=1e-11)


@pytest.mark.parametrize("name", SIGNALS)
def test_enframe_matches_reference(golden, name):
    g = golden("mfcc_inrepo")
    x = g[f"x_{name}"].astype(np.float64)
    for L, st in [(400, 160), (512, 256)]:
        ref = g[f"enframe_{name}_{L}_{st}"]
        got = O.enframe(x, L, st)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-14, atol=1e-13)


def test_filterbank_and_stmfcc(golden):
    g = golden("mfcc_inrepo")
    for fs, L, _ in GEOMS:
        fb, fr = O.mfccInitFilterBanks(fs, L)
        np.testing.assert_allclose(fb, g[f"fbank_{fs}_{L}"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(fr, g[f"freqs_{fs}_{L}"], rtol=0, atol=1e-12)
    fb, _ = O.mfccInitFilterBanks(8000, 512)
    np.testing.assert_allclose(O.stMFCC(g["stmfcc_X"], fb, 13), g["stmfcc_out"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", SIGNALS)
def test_table_pipeline_equals_inrepo(golden, name):
    """rfft + folded filterbank + DCT matrix == the reference's full-FFT arithmetic."""
    g = golden("mfcc_inrepo")
    x = g[f"x_{name}"]
    for fs, L, st in GEOMS:
        cfg, w, fb, dct = O.inrepo_tables(fs, L, st)
        got = O.mfcc_pipeline(x, cfg, w, fb, dct)
        np.testing.assert_allclose(got, g[f"mfcc_{name}_{fs}_{L}_{st}"], rtol=0, atol=1e-10)


def test_delta_matches_reference(golden):
    g = golden("delta_scale")
    for T in (1, 2, 5, 298):
        for D in (13, 26):
            f = g[f"feat_{T}_{D}"]
            np.testing.assert_allclose(O.delta(f), g[f"delta_{T}_{D}"], rtol=0, atol=1e-14)
            np.testing.assert_allclose(O.delta(f, N=3), g[f"delta3_{T}_{D}"], rtol=0, atol=1e-14)
            np.testing.assert_allclose(O.delta(O.delta(f)), g[f"ddelta_{T}_{D}"], rtol=0, atol=1e-14)
    d32 = O.delta(g["feat_f32"])
    assert d32.dtype == np.float32
    np.testing.assert_allclose(d32, g["delta_f32"], rtol=0, atol=1e-6)
    with pytest.raises(ValueError):
        O.delta(g["feat_f32"], N=0)


def test_scale_matches_sklearn(golden):
    g = golden("delta_scale")
    np.testing.assert_allclose(O.scale(g["scale_in"]), g["scale_out"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.scale(g["scale_one_in"]), g["scale_one_out"], rtol=0, atol=1e-12)


def test_scale_constant_columns_match_the_installed_sklearn():
    """A constant column whose float64 mean is an ulp off the constant (an all-silent utterance of a dialect with a log floor: every
    frame the same; c0 = -50.59...): the library's SECOND re-centring (sk:preprocessing/_data.py:279-292) makes it zeros, a restatement
    without it returns -1 / +1 — tools/fuzz_mfcc_batch.py found the kernels (zeros) and the oracle apart there.  scikit-learn is part
    of the image on both sides, so the restatement is compared with the library itself, on that case and on seeded random ones."""
    import warnings
    preprocessing = pytest.importorskip("sklearn.preprocessing")
    rng = np.random.default_rng(5)
    c0 = float(-np.sqrt(40.0) * 8.0)   # DCT-II ortho term 0 of forty log10(1e-8)
    cases = [np.tile(np.array([[c0, 0.0, 1e-3, -7.25]]), (13, 1)),
             np.tile(rng.standard_normal((1, 39)) * 40.0, (29, 1)),
             np.hstack([np.full((17, 1), 1.0 / 3.0), rng.standard_normal((17, 3)), np.full((17, 1), -50.59644256269407)]),
             rng.standard_normal((298, 26)) * 5.0 + 3.0,
             (rng.standard_normal((50, 13)) * 1e-9) + 1e3]
    hit = False
    for X in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = preprocessing.scale(X.copy())
        got = O.scale(X)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
        const = (X == X[:1]).all(axis=0)
        if const.any():
            assert np.abs(ref[:, const]).max() == 0.0
            hit = hit or bool(np.nanstd(X, axis=0)[const].max() > 0)
    assert hit   # (at least one constant column whose computed standard deviation is NOT zero: the case the second re-centring is for)


def test_scale_nan_rows_match_sklearn(golden):
    """NaN rows (digital silence through a dialect without a log floor, GMM_UBM.py:89-93): statistics over the other entries, NaN kept;
    +-inf raises like the library (tests/golden/make_golden_nan.py)."""
    g = golden("scale_nan")
    for k in ("nanrows", "nancols"):
        got, ref = O.scale(g[k + "_in"]), g[k + "_out"]
        assert (np.isnan(got) == np.isnan(ref)).all()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12, equal_nan=True)
    assert int(g["inf_raises"]) == 1
    bad = g["nanrows_in"].copy()
    bad[3, 2] = -np.inf
    with pytest.raises(ValueError):
        O.scale(bad)


@pytest.mark.parametrize("K,D", [(1, 13), (16, 26), (64, 39), (5, 7), (40, 39)])
def test_gmm_score_samples_matches_sklearn(golden, K, D):
    g = golden("gmm")
    w, mu, cov, X = g[f"w_{K}_{D}"], g[f"mu_{K}_{D}"], g[f"cov_{K}_{D}"], g[f"X_{K}_{D}"]
    np.testing.assert_allclose(O.gmm_score_samples(w, mu, cov, X), g[f"ss_{K}_{D}"], rtol=1e-12, atol=1e-11)
    assert abs(O.gmm_score(w, mu, cov, X) - float(g[f"score_{K}_{D}"])) < 1e-11


def test_score_matrix_matches_reference_loop(golden):
    g = golden("gmm")
    ubm = (g["sm_ubm_w"], g["sm_ubm_mu"], g["sm_ubm_cov"])
    models = [(ubm[0], mu, ubm[2]) for mu in g["sm_spk_mu"]]
    offs = np.concatenate([[0], np.cumsum(g["sm_lens"])])
    feats = [g["sm_feats"][offs[j]:offs[j + 1]] for j in range(len(g["sm_lens"]))]
    pred, am = O.score_matrix(models, ubm, feats)
    np.testing.assert_allclose(pred, g["sm_pred"], rtol=0, atol=1e-11)
    assert (am == g["sm_argmax"]).all()


@pytest.mark.parametrize("d", [128, 256, 512, "tie"])
def test_cosine_matches_scipy(golden, d):
    g = golden("cosine")
    X, C = g[f"X_{d}"], g[f"C_{d}"]
    # scipy computes u.u in float32 for float32 u (sp:spatial/distance.py:682-685) => ~1e-7 differences
    np.testing.assert_allclose(O.cosine_matrix(X, C), g[f"dist_{d}"], rtol=0, atol=5e-7)
    assert (O.identify(X, C) == g[f"argmin_{d}"]).all()


def test_eval_rule():
    C = np.eye(4)
    assert O.eval_rule(np.array([0.0, 1.0, 0.1, 0.0]), ["a", "b", "c", "d"], C) == "b"
    # every distance >= 1 -> None (d_vector.py:352-357)
    assert O.eval_rule(np.array([-1.0, -1.0, -1.0, -1.0]), ["a", "b", "c", "d"], C) is None


def test_unpinned_presets_shapes():
    """report/final.pdf IV-B-2: 1 s @ 16 kHz -> 98 x 13 (the only pinned fact for sidekit)."""
    x = np.random.default_rng(0).standard_normal(16000)
    out = O.sidekit_mfcc(x, fs=16000)
    assert out[0].shape == (98, 13) and out[1].shape == (98,) and out[2] is None and out[3] is None
    assert O.sidekit_mfcc(np.random.default_rng(1).standard_normal(48000))[0].shape == (298, 13)
    cfg, w, fb, dct = O.sidekit_tables()
    assert fb.shape == (24, 257) and int((fb != 0).sum()) == 454
    assert O.librosa_mfcc_flat(np.random.default_rng(2).standard_normal(48000)).shape == (94 * 13,)
    assert O.extract_feature_one(x).shape == (98, 26)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_gmm_em_matches_sklearn_fit(golden, tag):
    """oracle EM loop vs sklearn GaussianMixture(...).fit from the same initial parameters (tests/golden/make_golden_em.py)."""
    g = golden("gmm_em")
    K, D, n, max_iter, tol = g[tag + "_cfg"]
    w, mu, cov, n_iter, lb, conv = O.gmm_fit(g[tag + "_X"], g[tag + "_w0"], g[tag + "_mu0"], g[tag + "_cov0"],
                                             max_iter=int(max_iter), tol=float(tol))
    assert n_iter == int(g[tag + "_niter"]) and conv == bool(g[tag + "_conv"])
    assert abs(lb - float(g[tag + "_lb"])) < 1e-11
    for got, ref in ((w, g[tag + "_w"]), (mu, g[tag + "_mu"]), (cov, g[tag + "_cov"])):
        assert np.abs(got - ref).max() < 1e-11


def test_dtw_oracle_matches_plain_recurrence():
    """the vectorised anti-diagonal oracle against the package's plain double loop (accelerated_dtw, warp = 1) and its
    elementary properties: d(x, x) = 0, symmetry, a known hand value"""
    rng = np.random.default_rng(0)
    for dim in (1, 4):
        x = rng.standard_normal((17, dim))
        y = rng.standard_normal((23, dim))
        r, c = len(x), len(y)
        D0 = np.zeros((r + 1, c + 1))
        D0[0, 1:] = np.inf
        D0[1:, 0] = np.inf
        D1 = D0[1:, 1:]
        D0[1:, 1:] = np.sqrt(((x[:, None] - y[None]) ** 2).sum(-1))
        for i in range(r):
            for j in range(c):
                D1[i, j] += min(D0[i, j], D0[i + 1, j], D0[i, j + 1])
        assert abs(O.dtw_distance(x, y) - D1[-1, -1]) < 1e-12
        assert abs(O.dtw_distance(x, y, True) - D1[-1, -1] / (r + c)) < 1e-12
        assert O.dtw_distance(x, x) == 0.0
        assert abs(O.dtw_distance(x, y) - O.dtw_distance(y, x)) < 1e-12
    assert O.dtw_distance([0.0, 1.0, 2.0], [0.0, 2.0]) == 1.0


def test_dtw_path_oracle():
    """the oracle's path obeys the package's invariants: starts at (0, 0), ends at (r-1, c-1), unit steps, and its cost is d"""
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(31), rng.standard_normal(45)
    d, p, q = O.dtw_path(x, y)
    assert (p[0], q[0]) == (0, 0) and (p[-1], q[-1]) == (30, 44)
    dp, dq = np.diff(p), np.diff(q)
    assert ((dp >= 0) & (dq >= 0) & (dp + dq >= 1) & (dp <= 1) & (dq <= 1)).all()
    assert abs(np.abs(x[p] - y[q]).sum() - d) < 1e-9 and abs(d - O.dtw_distance(x, y)) < 1e-12
    t = O.generate_template([x, y, rng.standard_normal(20)])
    assert t.shape == y.shape


def test_plp_oracle_building_blocks():
    """PLP restatement (parity unpinned: sidekit absent) — each block against an independent formulation."""
    from scipy.linalg import solve_toeplitz
    from scipy.signal import lfilter
    rng = np.random.default_rng(3)
    # Levinson-Durbin = the Toeplitz normal equations; the error is r0 + a . r[1:]
    x = rng.standard_normal(500)
    r = np.correlate(x, x, "full")[499:499 + 13]
    a, e = O.levinson(r, 12)
    np.testing.assert_allclose(a, solve_toeplitz(r[:12], -r[1:13]), atol=1e-10)
    assert abs(e - (r[0] + a @ r[1:13])) < 1e-9 * r[0]
    # LPC -> cepstrum recursion = cepstrum of the all-pole spectrum 1 / |A(w)|^2 (c_n of ln(1/A), n >= 1)
    poly = np.concatenate(([1.0], 0.5 * a))       # a stable polynomial
    c = O.lpc2cep(poly[None, :] / 2.0, 13)[0]     # gain 2: c0 = ln 2
    assert abs(c[0] - np.log(2.0)) < 1e-12
    A = np.fft.fft(poly, 4096)
    ceps = np.fft.ifft(-np.log(A)).real
    np.testing.assert_allclose(c[1:], ceps[1:13], atol=1e-9)
    # RASTA: four zero outputs, then the IIR started from the FIR-only state
    xs = rng.standard_normal((50, 3)) + 5.0
    y = O.rasta_filt(xs)
    assert np.all(y[:4] == 0.0)
    numer = np.array([0.2, 0.1, 0.0, -0.1, -0.2])
    for b in range(3):
        full = lfilter(numer, [1.0], xs[:, b])    # FIR part over everything
        ref = np.zeros(50)
        for t in range(4, 50):
            ref[t] = full[t] + 0.94 * ref[t - 1]
        np.testing.assert_allclose(y[:, b], ref, atol=1e-12)
    # Bark bank: 21 bands at 16 kHz, 17 at 8 kHz, peak weight 1 inside every band, equal-loudness zero at 0 Hz
    assert O.plp_num_bands(16000) == 21 and O.plp_num_bands(8000) == 17
    w = O.fft2barkmx(512, 16000, 21)
    assert w.shape == (21, 257) and np.allclose(w.max(axis=1), 1.0) and (w > 0).all()
    eql = O.plp_equal_loudness(21, 8000.0)
    assert eql[0] == 0.0 and np.all(np.diff(eql[:15]) > 0)
    # autocorrelation in dolpc = real IDFT of the symmetric extension: lag 0 is the mean of the extended spectrum
    spec = rng.uniform(0.5, 2.0, (4, 21))
    lp = O.dolpc(spec, 12)
    assert lp.shape == (4, 13) and np.isfinite(lp).all()


def test_plp_oracle_shape_and_rasta_head():
    """1 s at 16 kHz -> 98 x 13 (the only pinned fact, report/final.pdf IV-B-2); the first four frames carry the flat-spectrum
    cepstrum because RASTA zeroes them in the log domain (rastamat behaviour)"""
    rng = np.random.default_rng(4)
    x = 0.3 * rng.standard_normal(16000)
    c = O.sidekit_plp(x)[0]
    assert c.shape == (98, 13) and np.isfinite(c).all()
    assert np.allclose(c[:4], c[0]) and not np.allclose(c[4], c[0])
    c2 = O.sidekit_plp(x, rasta=False)[0]
    assert not np.allclose(c2[0], c2[1])
    assert O.extract_feature_plp_one(x).shape == (98, 26)
    assert O.sidekit_plp(x[:300])[0].shape == (0, 13)
