"""GPU-backed mirror of the MFCC wrappers of the reference's ``MFCC_DTW.py`` (lines 28-54).

``load_train`` / ``load_test`` of the reference take ``mfcc_extract=`` (MFCC_DTW.py:122,155): pass ``_MFCC`` or
``MFCC_lib`` from this module there.  DTW matching itself is out of scope."""
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
