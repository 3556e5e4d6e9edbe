"""GPU-backed mirror of the reference's ``utils/processing.py`` call surface.

Same names, argument meaning and return shapes/dtypes as the reference (file:line cited per function); the work is
done by the fused HIP MFCC kernel through the C-ABI (include/ssp.h).  No CPU fallback.
"""
from __future__ import annotations

import functools

import numpy as np

from .. import api, frontend

eps = 1e-8  # utils/processing.py:17


@functools.lru_cache(maxsize=32)
def _plan(fs, frameSize, step, n_ceps, frames_only):
    ctx = api.default_context()
    tables = frontend.preset_inrepo(fs, frameSize, step, n_ceps=n_ceps)
    return api.MfccPlan(ctx, tables)


def mfccInitFilterBanks(fs, nfft):
    """utils/processing.py:42-88 — host-side table: (fbank (40, nfft), freqs (42,)) float64."""
    return frontend.mfccInitFilterBanks(fs, nfft)


def enframe(wavData, frameSize=400, step=160):
    """utils/processing.py:19-38 — (frameSize, ceil(N/step)) float64, zero padded tail, symmetric Hamming applied
    (pre-emphasis is a commented-out line in the reference and is not applied).  Runs on the GPU (ssp_enframe)."""
    x = np.ascontiguousarray(np.asarray(wavData).reshape(-1), dtype=np.float32)
    frames = api.enframe(api.default_context(), x, int(frameSize), int(step), np.hamming(frameSize))
    return np.asarray(frames, dtype=np.float64)


def stMFCC(X, fbank, n_mfcc_feats):
    """utils/processing.py:91-107 — ceps = DCT-II_ortho(log10(X . fbank^T + 1e-8))[:n_mfcc_feats] for one spectrum X
    (or a batch on the leading axis).  Runs on the GPU (ssp_cepstrum)."""
    X = np.asarray(X)
    fbank = np.asarray(fbank)
    rows = np.ascontiguousarray(X.reshape(-1, X.shape[-1]), dtype=np.float32)
    dct = frontend.dct2_ortho(fbank.shape[0], 0, int(n_mfcc_feats))
    out = api.cepstrum(api.default_context(), rows, fbank, dct, frontend.LOG_LOG10, frontend.FLOOR_ADD_EPS, eps)
    return np.asarray(out, dtype=np.float64).reshape(X.shape[:-1] + (int(n_mfcc_feats),))


def MFCC(raw_signal, fs=8000, frameSize=512, step=256):
    """utils/processing.py:110-144 — (frames, 13) float64 MFCC matrix of one utterance, computed on the GPU."""
    x = np.ascontiguousarray(np.asarray(raw_signal).reshape(-1), dtype=np.float32)
    plan = _plan(int(fs), int(frameSize), int(step), 13, False)
    seg = api.Segments.from_lengths(plan.ctx, [x.shape[0]])
    feats = plan.run(x, seg)
    return np.asarray(feats, dtype=np.float64)


def MFCC_batch(signals, fs=8000, frameSize=512, step=256):
    """Batched form of MFCC(): list of 1-D signals -> list of (frames_i, 13) float64 (one kernel launch)."""
    sig = [np.asarray(s, dtype=np.float32).reshape(-1) for s in signals]
    plan = _plan(int(fs), int(frameSize), int(step), 13, False)
    seg = api.Segments.from_lengths(plan.ctx, [s.shape[0] for s in sig])
    fseg = plan.frame_segments(seg)
    flat = np.concatenate(sig) if sig else np.zeros(0, dtype=np.float32)
    feats = np.asarray(plan.run(flat, seg, fseg), dtype=np.float64)
    return [feats[fseg.offsets[i]:fseg.offsets[i + 1]] for i in range(len(sig))]
