"""GPU-backed mirror of the scoring half of the reference's ``d_vector.py`` (nn_model.test / enroll / eval,
d_vector.py:296-361) and of Data_gen's feature front end (d_vector.py:80-98).  The Keras networks are out of scope:
embeddings are inputs here (BASELINE.json config 5: "precomputed 256-d embeddings")."""
from __future__ import annotations

import functools

import numpy as np

from . import api, frontend
from .GMM_UBM import delta as _delta


def cosine_scores(X, centroids):
    """(N, S) float64 matrix of scipy.spatial.distance.cosine(X[i], centroids[j]) — d_vector.py:315-318."""
    r = api.cosine_identify(api.default_context(), X, np.asarray(centroids, dtype=np.float32), dist=True, argmin=False, minval=False)
    return np.asarray(r["dist"], dtype=np.float64)


def identify(X, centroids):
    """argmin_j cosine(X[i], centroids[j]) (first index on ties) — d_vector.py:319."""
    r = api.cosine_identify(api.default_context(), X, np.asarray(centroids, dtype=np.float32), dist=False, argmin=True, minval=False)
    return np.asarray(r["argmin"]).astype(np.int64)


class Data_gen:
    """Feature front end of d_vector.Data_gen: 1-second chunks -> sidekit mfcc(x, fs)[0] -> (98, 13) (d_vector.py:80-98)."""

    def __init__(self, sample_rate=16000):
        self.sample_rate = sample_rate

    @staticmethod
    def delta(feat, N=2):
        """d_vector.py:143-160 (identical to GMM_UBM.delta)."""
        return _delta(feat, N)

    @functools.lru_cache(maxsize=4)
    def _plan(self, feature_type):
        if feature_type != 'MFCC':
            raise NameError  # d_vector.py:94-95
        return api.MfccPlan(api.default_context(), frontend.preset_sidekit(fs=self.sample_rate))

    def extract_feature(self, x, y, feature_type='MFCC'):
        """x: list of audio arrays, y: labels.  Cuts each audio into 1 s chunks (d_vector.py:80-83), extracts
        MFCC per chunk, drops chunks whose features contain NaN (d_vector.py:97-98).  Returns (feature, label)."""
        plan = self._plan(feature_type)
        sr = self.sample_rate
        chunks, labels = [], []
        for xi, yi in zip(x, y):
            xi = np.asarray(xi, dtype=np.float32).reshape(-1)
            for j in range(xi.shape[0] // sr):
                chunks.append(xi[j * sr:(j + 1) * sr])
                labels.append(yi)
        if not chunks:
            return [], []
        seg = api.Segments.from_lengths(plan.ctx, [sr] * len(chunks))
        fseg = plan.frame_segments(seg)
        feats = np.asarray(plan.run(np.concatenate(chunks), seg, fseg), dtype=np.float64)
        feature, label = [], []
        for i, lab in enumerate(labels):
            f = feats[fseg.offsets[i]:fseg.offsets[i + 1]]
            if np.isnan(f).sum() > 0:
                continue
            feature.append(f)
            label.append(lab)
        return feature, label


class nn_model:
    """Scoring half of d_vector.nn_model.  ``X_*`` are embeddings (outputs of the speaker network)."""

    def __init__(self):
        self.d_vector = {}  # name -> mean embedding, the dict the reference pickles (d_vector.py:333-344)

    def test(self, X_train, Y_train, X_val, Y_val):
        """d_vector.py:296-320: per-speaker centroids of X_train (one-hot Y_train), cosine distance of every
        X_val row to every centroid, accuracy of the arg-min."""
        num = Y_train.shape[1]
        lab = np.argmax(Y_train, axis=1)  # decoding the one-hot labels is index bookkeeping, not arithmetic
        avg = np.asarray(api.centroids(api.default_context(), np.asarray(X_train, dtype=np.float32), lab, num))
        pred = identify(np.asarray(X_val, dtype=np.float32), avg)
        return (np.argmax(Y_val, axis=1) == pred).sum() / X_val.shape[0]

    def enroll(self, X_train, name):
        """d_vector.py:322-344: store the mean embedding under ``name`` (overwrites, like the reference)."""
        if name in self.d_vector:
            print("sample already exists")
        X = np.asarray(X_train, dtype=np.float32)
        self.d_vector[name] = np.asarray(api.centroids(api.default_context(), X, np.zeros(len(X), np.int32), 1))[0]

    def eval(self, target):
        """d_vector.py:346-361: linear scan in dict order; the minimum is kept only while < 1; returns the name or None."""
        if not self.d_vector:
            return None
        names = list(self.d_vector.keys())
        C = np.stack([self.d_vector[n] for n in names]).astype(np.float32)
        r = api.cosine_identify(api.default_context(), np.asarray(target, dtype=np.float32).reshape(1, -1), C,
                                dist=False, argmin=True, minval=True)
        mn = float(np.asarray(r["min"])[0])
        return names[int(np.asarray(r["argmin"])[0])] if mn < 1 else None
