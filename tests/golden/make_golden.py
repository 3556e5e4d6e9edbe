#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

Imports /root/reference (read-only) with empty stubs for the audio/GUI packages
that are not installed, calls the reference's own functions
(utils.processing.enframe / mfccInitFilterBanks / stMFCC / MFCC, MFCC_DTW._MFCC,
GMM_UBM.delta) and the very libraries it calls for scoring
(sklearn.mixture.GaussianMixture, sklearn.preprocessing.scale,
scipy.spatial.distance.cosine) on seeded inputs, and stores inputs + outputs.

The fixtures are DATA (inputs and expected outputs).  Nothing from the reference
travels to the GPU box except these arrays.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    if not hasattr(np, "int"):
        np.int = int  # utils/processing.py:79,83 use the removed alias
    _stub("pyaudio", PyAudio=object, paInt16=8)
    _stub("simpleaudio")
    _stub("python_speech_features")
    _stub("sidekit")
    _stub("sidekit.frontend")
    _stub("sidekit.frontend.features", mfcc=None, plp=None)
    _stub("librosa")
    _stub("dtw", dtw=None, accelerated_dtw=None)
    _stub("fastdtw", fastdtw=None)
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import utils.processing as proc  # noqa
    import MFCC_DTW  # noqa
    import GMM_UBM  # noqa
    return proc, MFCC_DTW, GMM_UBM


def synth_audio(utt, n, fs, S=10):
    """SURVEY.md 8(d) synthetic-audio recipe (host, parity sets)."""
    rng = np.random.default_rng(1234 + utt)
    s = utt % S
    f0 = 90 + 3 * s
    t = np.arange(n) / fs
    x = 0.3 * sum(np.sin(2 * np.pi * h * f0 * t) / h for h in range(1, 6)) * (0.6 + 0.4 * np.sin(2 * np.pi * 3 * t))
    x = x + 0.05 * rng.standard_normal(n)
    return np.clip(x, -1, 1).astype(np.float32)


def main():
    proc, MFCC_DTW, GMM_UBM = import_reference()
    from sklearn.mixture import GaussianMixture
    from sklearn import preprocessing
    from scipy.spatial.distance import cosine

    # ------------------------------------------------------------------ MFCC-A
    out = {}
    rng = np.random.default_rng(0)
    signals = {
        "noise": (0.1 * rng.standard_normal(6000)).astype(np.float32),
        "tone": synth_audio(3, 8000, 8000),
        "silence": np.zeros(2048, dtype=np.float32),
        "ragged": synth_audio(5, 5000 + 77, 8000),       # N not a multiple of step
        "short": synth_audio(7, 300, 8000),              # N < frameSize
        "int16": (synth_audio(9, 4096, 8000) * 20000).astype(np.int16),
        "utt3s16k": synth_audio(11, 48000, 16000),
        "one_step": synth_audio(13, 256, 8000),
    }
    geoms = [(8000, 512, 256), (16000, 512, 256), (16000, 256, 128), (8000, 1024, 512)]
    for name, x in signals.items():
        out[f"x_{name}"] = x
        for (fs, L, st) in geoms:
            out[f"mfcc_{name}_{fs}_{L}_{st}"] = proc.MFCC(x, fs=fs, frameSize=L, step=st)
        out[f"flat_{name}"] = MFCC_DTW._MFCC(x)
        out[f"enframe_{name}_400_160"] = proc.enframe(x.astype(np.float64), 400, 160)
        out[f"enframe_{name}_512_256"] = proc.enframe(x.astype(np.float64), 512, 256)
    for (fs, L, st) in geoms:
        fb, fr = proc.mfccInitFilterBanks(fs, L)
        out[f"fbank_{fs}_{L}"] = fb
        out[f"freqs_{fs}_{L}"] = fr
    X = np.abs(rng.standard_normal(512))
    fb, _ = proc.mfccInitFilterBanks(8000, 512)
    out["stmfcc_X"] = X
    out["stmfcc_out"] = proc.stMFCC(X, fb, 13)
    np.savez_compressed(os.path.join(HERE, "mfcc_inrepo.npz"), **out)

    # ------------------------------------------------------------------ delta / scale
    out = {}
    for T in (1, 2, 5, 298):
        for D in (13, 26):
            f = rng.standard_normal((T, D))
            out[f"feat_{T}_{D}"] = f
            out[f"delta_{T}_{D}"] = GMM_UBM.delta(f)
            out[f"delta3_{T}_{D}"] = GMM_UBM.delta(f, N=3)
            out[f"ddelta_{T}_{D}"] = GMM_UBM.delta(GMM_UBM.delta(f))
    f32 = rng.standard_normal((50, 13)).astype(np.float32)
    out["feat_f32"] = f32
    out["delta_f32"] = GMM_UBM.delta(f32)
    sc_in = rng.standard_normal((298, 26)) * rng.uniform(0.1, 30, 26) + rng.uniform(-5, 5, 26)
    sc_in[:, 7] = 3.25  # constant column -> std 0 -> 1
    out["scale_in"] = sc_in
    out["scale_out"] = preprocessing.scale(sc_in)
    one = rng.standard_normal((1, 26))
    out["scale_one_in"] = one
    out["scale_one_out"] = preprocessing.scale(one)
    np.savez_compressed(os.path.join(HERE, "delta_scale.npz"), **out)

    # ------------------------------------------------------------------ GMM scoring
    out = {}

    def make_gmm(K, D, seed, mu_base=None):
        r = np.random.default_rng(seed)
        w = r.dirichlet(5 * np.ones(K))
        mu = r.standard_normal((K, D)) if mu_base is None else mu_base + 0.3 * r.standard_normal((K, D))
        cov = r.uniform(0.5, 2.0, (K, D))
        g = GaussianMixture(n_components=K, covariance_type="diag")
        g.weights_, g.means_, g.covariances_ = w, mu, cov
        g.precisions_cholesky_ = 1.0 / np.sqrt(cov)
        return g

    for (K, D) in [(1, 13), (16, 26), (64, 39), (5, 7), (40, 39)]:
        g = make_gmm(K, D, 100 + K + D)
        Xs = np.random.default_rng(K * D).standard_normal((123, D)) * 1.3
        out[f"w_{K}_{D}"], out[f"mu_{K}_{D}"], out[f"cov_{K}_{D}"] = g.weights_, g.means_, g.covariances_
        out[f"X_{K}_{D}"] = Xs
        out[f"ss_{K}_{D}"] = g.score_samples(Xs)
        out[f"score_{K}_{D}"] = np.float64(g.score(Xs))
    # (U,S) score-difference matrix, reference loop GMM_UBM.py:181-197, U=100, S=10, K=16, D=26
    K, D, S, U = 16, 26, 10, 100
    ubm = make_gmm(K, D, 7)
    spk = []
    for s in range(S):
        g = make_gmm(K, D, 700 + s, mu_base=ubm.means_)
        g.weights_, g.covariances_ = ubm.weights_, ubm.covariances_
        g.precisions_cholesky_ = ubm.precisions_cholesky_
        spk.append(g)
    r = np.random.default_rng(99)
    lens = r.integers(20, 120, U)
    feats = []
    for j in range(U):
        s = j % S
        comp = r.choice(K, size=lens[j], p=ubm.weights_)
        feats.append(spk[s].means_[comp] + np.sqrt(ubm.covariances_[comp]) * r.standard_normal((lens[j], D)))
    pred = np.zeros((U, S))
    for i in range(S):
        for j in range(U):
            pred[j, i] = spk[i].score(feats[j]) - ubm.score(feats[j])
    out["sm_ubm_w"], out["sm_ubm_mu"], out["sm_ubm_cov"] = ubm.weights_, ubm.means_, ubm.covariances_
    out["sm_spk_mu"] = np.stack([g.means_ for g in spk])
    out["sm_lens"] = lens
    out["sm_feats"] = np.vstack(feats)
    out["sm_pred"] = pred
    out["sm_argmax"] = pred.argmax(axis=1)
    out["sm_ubm_score"] = np.array([ubm.score(f) for f in feats])
    np.savez_compressed(os.path.join(HERE, "gmm.npz"), **out)

    # ------------------------------------------------------------------ cosine
    out = {}
    for d in (128, 256, 512):
        r = np.random.default_rng(11 + d)
        C = r.standard_normal((50, d))
        lab = r.integers(0, 50, 200)
        Xe = (C[lab] + 0.7 * r.standard_normal((200, d))).astype(np.float32)
        if d == 256:
            Xe[5] = 0.5 * (C[3] / np.linalg.norm(C[3]) + C[9] / np.linalg.norm(C[9])).astype(np.float32)  # near tie
            Xe[6] = -C[4].astype(np.float32)  # distance ~2 to centroid 4
        dist = np.zeros((200, 50))
        for i in range(200):
            for j in range(50):
                dist[i, j] = cosine(Xe[i], C[j])   # d_vector.py:315-318
        out[f"X_{d}"], out[f"C_{d}"], out[f"dist_{d}"], out[f"argmin_{d}"] = Xe, C, dist, dist.argmin(axis=1)
    # exact tie: two identical centroids -> first index wins
    r = np.random.default_rng(5)
    C = r.standard_normal((8, 64))
    C[5] = C[2]
    Xe = r.standard_normal((4, 64)).astype(np.float32)
    Xe[0] = C[2].astype(np.float32)
    dist = np.array([[cosine(Xe[i], C[j]) for j in range(8)] for i in range(4)])
    out["X_tie"], out["C_tie"], out["dist_tie"], out["argmin_tie"] = Xe, C, dist, dist.argmin(axis=1)
    np.savez_compressed(os.path.join(HERE, "cosine.npz"), **out)
    print("golden fixtures written to", HERE)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(" ", fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
