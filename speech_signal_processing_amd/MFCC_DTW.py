"""GPU-backed mirror of the MFCC wrappers of the reference's ``MFCC_DTW.py`` (lines 28-54).

``load_train`` / ``load_test`` of the reference take ``mfcc_extract=`` (MFCC_DTW.py:122,155): pass ``_MFCC`` or
``MFCC_lib`` from this module there.  The matcher (distance_dtw / distance_train / distance_test, MFCC_DTW.py:57-108, and the
arg-min classification of test(), MFCC_DTW.py:187-217) runs as one all-pairs DTW kernel (api.dtw_distances)."""
from __future__ import annotations

import functools

import numpy as np

from . import api, frontend
from .utils.processing import MFCC


@functools.lru_cache(maxsize=8)
def _librosa_plan(n_mfcc):
    return api.MfccPlan(api.default_context(), frontend.preset_librosa(8000, n_mfcc))


def MFCC_lib(raw_signal, n_mfcc=13):
    """MFCC_DTW.py:28-31 — librosa.feature.mfcc(y.astype('float32'), n_mfcc=n_mfcc, sr=8000).T.flatten()."""
    x = np.ascontiguousarray(np.asarray(raw_signal).astype("float32").reshape(-1))
    plan = _librosa_plan(int(n_mfcc))
    seg = api.Segments.from_lengths(plan.ctx, [x.shape[0]])
    return np.asarray(plan.run(x, seg)).flatten()


def _MFCC(raw_signal):
    """MFCC_DTW.py:33-54 — MFCC(raw_signal, fs=8000, frameSize=512, step=256).flatten()."""
    return MFCC(raw_signal, fs=8000, frameSize=512, step=256).flatten()


def distance_dtw(sample_x, sample_y, show=False, dtw_method=1, dist=None, normalize=False):
    """MFCC_DTW.py:57-76 — dtw_method=1: d of dtw.accelerated_dtw(sample_x, sample_y, dist='euclidean') (``normalize=True`` divides by
    len(x) + len(y), dtw <= 1.3.3); dtw_method=2: d of fastdtw.fastdtw(sample_x, sample_y, dist=euclidean) (radius 1; the reference's
    flattened 1-D sequences)."""
    if dtw_method == 2:
        return float(api.fastdtw_distances(api.default_context(), [sample_x], [sample_y])[0, 0])
    if dtw_method != 1:
        raise ValueError("dtw_method must be 1 (accelerated_dtw) or 2 (fastdtw)")
    return float(api.dtw_distances(api.default_context(), [sample_x], [sample_y], normalize=normalize)[0, 0])


def distance_train(data, normalize=False):
    """MFCC_DTW.py:79-95 — symmetric matrix of pairwise DTW distances with a zero diagonal."""
    d = np.asarray(api.dtw_distances(api.default_context(), data, data, normalize=normalize), dtype=np.float64)
    d = np.triu(d, 1)
    return d + d.T


def distance_test(x_test, x_train, show=False, normalize=False):
    """MFCC_DTW.py:98-108 — (1, len(x_train)) distances of one test sample to every training sample:
    distance[0, k] = distance_dtw(x_train[k], x_test)."""
    return np.asarray(api.dtw_distances(api.default_context(), x_train, [x_test], normalize=normalize), dtype=np.float64).T.copy()


def classify(x_test, x_train, y_train, normalize=False):
    """The matching loop of test() (MFCC_DTW.py:196-203) for a whole test set in ONE launch: distances (n_test, n_train)
    with distances[i, k] = distance_dtw(x_train[k], x_test[i]) and y_pred[i] = y_train[argmin_k]."""
    d = np.asarray(api.dtw_distances(api.default_context(), x_train, x_test, normalize=normalize), dtype=np.float64).T
    return d, [y_train[k] for k in d.argmin(axis=1)]


def generate_template(x):
    """MFCC_DTW.py:187-217 — one template per speaker: start from the longest sample; for every other sample warp it onto
    the template (accelerated_dtw path), average the aligned values and keep the first entry of every template index, so the
    template keeps the longest sample's length.  The DTW + traceback run on the GPU in float64 (api.dtw_path)."""
    ctx = api.default_context()
    max_length_index = int(np.argmax([np.asarray(_x).shape[0] for _x in x]))  # first of the longest, like the reference's scan
    template = np.asarray(x[max_length_index], dtype=np.float64)
    for index, _x in enumerate(x):
        if index == max_length_index:
            continue
        _x = np.asarray(_x, dtype=np.float64)
        _, p0, p1 = api.dtw_path(ctx, _x, template)
        template = (_x[p0] + template[p1]) / 2
        keep = np.empty(len(p1), dtype=bool)
        keep[0] = True
        keep[1:] = p1[1:] != p1[:-1]
        template = template[keep]
    return template


def vote(label):
    """MFCC_DTW.py:220-229 — the most frequent label; among equally frequent ones the one met first (a stable sort by count over a dict
    in insertion order, as the reference's)."""
    counts = {}
    for l in np.array(label):
        counts[l] = counts.get(l, 0) + 1
    return sorted(counts.items(), key=lambda kv: kv[1], reverse=True)[0][0]
