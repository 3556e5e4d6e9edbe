"""GPU-backed mirror of the hot-path functions of the reference's ``GMM_UBM.py``.

* ``delta``            GMM_UBM.py:53-69
* ``extract_feature``  GMM_UBM.py:72-118  (sidekit mfcc -> [c, delta c] -> per-utterance scale; one fused kernel; 'PLP': plp back end)
* ``score_matrix``     the scoring loops GMM_UBM.py:181-197 as a function that returns what the reference prints
* ``GMM``              GMM_UBM.py:134-199: trains one GMM per speaker + the UBM (EM on the GPU, gmm_train.GaussianMixture)
                       or takes pre-trained models, then scores
"""
from __future__ import annotations

import functools
import os
import pickle as pkl

import numpy as np

from . import api, frontend
from .gmm_train import GaussianMixture
from .sidekit_features import mfcc, plp, plp_batch  # noqa: F401  (GMM_UBM.py:20 imports both names)


def delta(feat, N=2):
    """GMM_UBM.py:53-69 — regression delta over +-N frames with edge padding; returns an array like ``feat``."""
    if N < 1:
        raise ValueError('N must be an integer >= 1')
    feat = np.asarray(feat)
    if feat.ndim != 2:
        raise ValueError("feat must be (NUMFRAMES, features)")
    ctx = api.default_context()
    seg = api.Segments.from_lengths(ctx, [feat.shape[0]])
    out = api.delta_features(ctx, feat, seg, N)
    return out.astype(feat.dtype if feat.dtype.kind == "f" else np.float64)


@functools.lru_cache(maxsize=8)
def _feature_plan(feature_type, fs, delta_order):
    if feature_type != 'MFCC':
        raise NameError  # GMM_UBM.py:100-101
    return api.MfccPlan(api.default_context(), frontend.preset_sidekit(fs=fs, delta_order=delta_order, cmvn=1))


def extract_feature(x, y, is_train=False, feature_type='MFCC', fs=16000, delta_order=1):
    """GMM_UBM.py:72-118.  x: list of 1-D audio arrays, y: list of labels.
    Returns (feature, y) or (train_data, feature, y); every feature is (T_i, 26) float64 = scale([c, delta c]).
    ``delta_order=2`` appends delta-delta (39-d) — an extension used by the benchmark configs."""
    if feature_type == 'PLP':  # GMM_UBM.py:94-99: plp -> hstack(c, delta c) -> scale
        feature = _extract_plp(x, int(fs), int(delta_order))
        if not is_train:
            return feature, y
        train_data = {}
        for f, lab in zip(feature, y):
            train_data[lab] = np.vstack((train_data[lab], f)) if lab in train_data else f
        return train_data, feature, y
    plan = _feature_plan(feature_type, int(fs), int(delta_order))
    flat, lens = api.flatten_signals(x)   # (int16 PCM — what load_data reads, GMM_UBM.py:24-50 — goes to the device as int16)
    seg = api.Segments.from_lengths(plan.ctx, lens)
    fseg = plan.frame_segments(seg)
    feats = np.asarray(plan.run(flat, seg, fseg), dtype=np.float64)
    feature = [feats[fseg.offsets[i]:fseg.offsets[i + 1]] for i in range(len(lens))]
    if not is_train:
        return feature, y
    train_data = {}
    for f, lab in zip(feature, y):
        train_data[lab] = np.vstack((train_data[lab], f)) if lab in train_data else f
    return train_data, feature, y


def _extract_plp(x, fs, delta_order):
    """[c, delta c (, delta delta c)] of the PLP cepstra, per-utterance scaled; every stage on the GPU."""
    ctx = api.default_context()
    c, fseg = plp_batch(x, fs=fs)
    blocks = [c]
    for _ in range(delta_order):
        blocks.append(api.delta_features(ctx, blocks[-1], fseg, 2))
    feats = np.asarray(api.cmvn_features(ctx, np.ascontiguousarray(np.hstack(blocks)), fseg), dtype=np.float64)
    return [feats[fseg.offsets[i]:fseg.offsets[i + 1]] for i in range(len(x))]


def score_matrix(models, ubm, feats):
    """GMM_UBM.py:181-187 as a function: pred[j, i] = models[i].score(feats[j]) - ubm.score(feats[j]).

    models / ubm: fitted sklearn GaussianMixture(covariance_type='diag') objects (or anything with weights_,
    means_, covariances_).  feats: list of (T_j, D) arrays.  Returns (pred (U, S) float64, argmax (U,) int64)."""
    ctx = api.default_context()
    scorer = api.GmmScorer.from_sklearn(ctx, models, ubm)
    fseg = api.Segments.from_lengths(ctx, [len(f) for f in feats])
    flat = np.ascontiguousarray(np.vstack(feats), dtype=np.float32) if len(feats) else np.zeros((0, scorer.D), np.float32)
    r = scorer.score(flat, fseg, scores=True, argmax=True)
    sc = np.asarray(r["scores"], dtype=np.float64)
    return sc[:, 1:] - sc[:, :1], np.asarray(r["argmax"]).astype(np.int64)


def identify_with_confidence(models, ubm, feature):
    """The GUI's read-out of one utterance (UI/GMM_UBM_GUI.py:102-113): prob[0, i] = models[i].score(feature) - ubm.score(feature),
    res = argmax, then the softmax of the score differences, exp(prob) / sum exp(prob).  Returns (index, softmax probability of that
    index, the (1, S) softmax row)."""
    pred, am = score_matrix(models, ubm, [feature])
    e = np.exp(pred)
    prob = e / e.sum(axis=1)
    return int(am[0]), float(prob[0, am[0]]), prob


def save_models(gmms, ubm, model_dir="Model"):
    """The two pickles the reference writes after training (GMM_UBM.py:173-179): Model/GMM_MFCC_model.pkl (list of the
    per-speaker models) and Model/UBM_MFCC_model.pkl."""
    if not os.path.exists(model_dir):
        os.mkdir(model_dir)
    with open(os.path.join(model_dir, "GMM_MFCC_model.pkl"), "wb") as f:
        pkl.dump(gmms, f)
    with open(os.path.join(model_dir, "UBM_MFCC_model.pkl"), "wb") as f:
        pkl.dump(ubm, f)


def load_models(model_dir="Model"):
    """GMM_UBM.py:141-146: the pickled speaker models and UBM (sklearn GaussianMixture objects written by the reference,
    or gmm_train.GaussianMixture objects written by save_models)."""
    with open(os.path.join(model_dir, "GMM_MFCC_model.pkl"), "rb") as f:
        gmms = pkl.load(f)
    with open(os.path.join(model_dir, "UBM_MFCC_model.pkl"), "rb") as f:
        ubm = pkl.load(f)
    return gmms, ubm


def GMM(train, x_train, y_train, x_test, y_test, n_components=16, model=None, random_state=None, model_dir=None):
    """GMM_UBM.py:134-199.  ``model`` falsy (the reference's default): one ``GaussianMixture(n_components, 'diag')`` per
    speaker is fitted on ``train[speaker]`` (speakers in ascending label order, like label_encoder.values()) and the UBM
    on the stacked training data (GMM_UBM.py:154-170), EM on the GPU; with ``model_dir`` (the reference always uses
    "Model") the two pickles of GMM_UBM.py:173-179 are written.  ``model=True``: load those pickles from ``model_dir``
    (default "Model", GMM_UBM.py:141-146).  ``model`` = (list_of_speaker_GMMs, UBM): use them as given.
    Prints and returns the train/test accuracies the reference prints; the models are left in ``GMM.last_model``."""
    if model is True:
        model = load_models(model_dir or "Model")
    elif not model:
        speakers = sorted(train.keys())
        gmms = [GaussianMixture(n_components=n_components, covariance_type='diag', random_state=random_state).fit(train[s])
                for s in speakers]
        ubm_train = np.vstack([train[s] for s in speakers])
        ubm = GaussianMixture(n_components=n_components, covariance_type='diag', random_state=random_state).fit(ubm_train)
        model = (gmms, ubm)
        if model_dir:
            save_models(gmms, ubm, model_dir)
    gmms, ubm = model
    GMM.last_model = model
    valid = score_matrix(gmms, ubm, x_train)[1]
    acc_train = (valid == np.array(y_train)).sum() / len(x_train)
    pred = score_matrix(gmms, ubm, x_test)[1]
    acc = (pred == np.array(y_test)).sum() / len(x_test)
    print("train acc {:.2%}, test acc {:.2%}".format(acc_train, acc))
    return acc_train, acc
