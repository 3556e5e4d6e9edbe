"""GPU-backed mirror of the MFCC wrappers of the reference's ``MFCC_DTW.py`` (lines 28-54).

``load_train`` / ``load_test`` (MFCC_DTW.py:122-184) walk ``<path>/<speaker>/*.wav`` like the reference's, with its ``mfcc_extract=`` plug
point: with this module's own ``_MFCC`` / ``MFCC_lib`` / ``MFCC`` every file of the directory tree goes through ONE batched kernel launch,
any other callable is applied per file as the reference applies it.  The matcher (distance_dtw / distance_train / distance_test, MFCC_DTW.py:57-108, and the
arg-min classification of test(), MFCC_DTW.py:187-217) runs as one all-pairs DTW kernel (api.dtw_distances)."""
from __future__ import annotations

import functools
import os
import random

import numpy as np

from . import api, frontend
from .utils.processing import MFCC, MFCC_batch
from .utils.tools import get_time, read


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


def MFCC_lib_batch(signals, n_mfcc=13):
    """MFCC_lib over a list of signals in one kernel launch."""
    flat, lens = api.flatten_signals(signals)   # (int16 wav data stays int16 up to the device; the float32 of MFCC_DTW.py:29 is taken there)
    plan = _librosa_plan(int(n_mfcc))
    seg = api.Segments.from_lengths(plan.ctx, lens)
    fseg = plan.frame_segments(seg)
    feats = np.asarray(plan.run(flat, seg, fseg))
    return [feats[fseg.offsets[i]:fseg.offsets[i + 1]].flatten() for i in range(len(lens))]


def _extract_all(signals, mfcc_extract):
    """``mfcc_extract`` over every signal: this module's own extractors run batched (one launch), anything else per signal."""
    if mfcc_extract is _MFCC:
        return [f.flatten() for f in MFCC_batch(signals, fs=8000, frameSize=512, step=256)]
    if mfcc_extract is MFCC:
        return MFCC_batch(signals)  # (the reference's default arguments: fs=8000, frameSize=512, step=256)
    if mfcc_extract is MFCC_lib:
        return MFCC_lib_batch(signals)
    return [mfcc_extract(x) for x in signals]


def _read_8k(path):
    """MFCC_DTW.py:137-145 / 170-178: the file's first channel (some recordings have two), every second sample (16 kHz -> 8 kHz)."""
    _, data = read(path)
    data = np.asarray(data)
    if data.ndim > 1:
        data = data[:, 0]
    return data[range(0, data.shape[0], 2)]


def sample(x, y, sample_num=2, whole_num=8):
    """MFCC_DTW.py:110-118 — ``sample_num`` of every speaker's ``whole_num`` consecutive utterances (four speakers), the same random
    positions for each."""
    index = random.sample(range(whole_num), sample_num)
    sample_x, sample_y = [], []
    for i in range(4):
        for _index in index:
            sample_x.append(x[_index + whole_num * i])
            sample_y.append(y[_index + whole_num * i])
    return sample_x, sample_y


def load_train(path='dataset/ASR/train', mfcc_extract=_MFCC):
    """MFCC_DTW.py:122-152 — one template per speaker directory (in os.listdir order): every wav of the directory read, down-sampled
    to 8 kHz, ``mfcc_extract``-ed, then generate_template over the speaker's feature sequences.  Returns (templates, labels)."""
    start_time = get_time()
    wav_dir = os.listdir(path)
    signals, owner = [], []
    print("Generate template according to train set.")
    for k, _dir in enumerate(wav_dir):
        for _path in os.listdir(os.path.join(path, _dir)):
            signals.append(_read_8k(os.path.join(path, _dir, _path)))
            owner.append(k)
    feats = _extract_all(signals, mfcc_extract)
    x, y_label = [], []
    for k, _dir in enumerate(wav_dir):
        x.append(generate_template([f for f, o in zip(feats, owner) if o == k]))
        y_label.append(_dir)
    print('Loading train data, extract mfcc feature and generate template spend {}s'.format(get_time(start_time)))
    return x, y_label


def load_test(path='dataset/ASR/test', mfcc_extract=MFCC, template=False):
    """MFCC_DTW.py:155-184 — every wav under ``<path>/<speaker>/`` read, down-sampled to 8 kHz and ``mfcc_extract``-ed; the label of a
    file is its directory.  (``template`` is accepted and ignored, as in the reference.)  Returns (features, labels)."""
    start_time = get_time()
    signals, y_label = [], []
    for _dir in os.listdir(path):
        for _path in os.listdir(os.path.join(path, _dir)):
            signals.append(_read_8k(os.path.join(path, _dir, _path)))
            y_label.append(_dir)
    x = _extract_all(signals, mfcc_extract)
    print('Loading test data and extract mfcc feature spend {}s'.format(get_time(start_time)))
    return x, y_label


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
