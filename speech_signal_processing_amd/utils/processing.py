"""GPU-backed mirror of the reference's ``utils/processing.py`` call surface.

Same names, argument meaning and return shapes/dtypes as the reference (file:line cited per function); the work is
done by the fused HIP MFCC kernel through the C-ABI (include/ssp.h).  No CPU fallback.
"""
from __future__ import annotations

import functools
import math

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
    """utils/processing.py:19-38 — (frameSize, ceil(N/step)) float64, zero padded tail, symmetric Hamming applied.

    Framing is pure data movement (no arithmetic beyond the window multiply); it is the first stage of the fused
    kernel and is exposed here as a strided host view for API completeness."""
    x = np.asarray(wavData, dtype=np.float64)
    wlen = x.shape[0]
    n_frames = math.ceil(wlen / step)
    padded = np.zeros((n_frames - 1) * step + frameSize if n_frames else 0)
    padded[: min(wlen, padded.shape[0])] = x[: padded.shape[0]]
    view = np.lib.stride_tricks.as_strided(padded, shape=(frameSize, n_frames),
                                           strides=(padded.strides[0], padded.strides[0] * step), writeable=False)
    return view * np.hamming(frameSize)[:, None]


def stMFCC(X, fbank, n_mfcc_feats):
    """utils/processing.py:91-107 — cepstrum of ONE magnitude spectrum (host; a 40x512 mat-vec + 40-point DCT).
    Kept for signature parity; bulk extraction goes through MFCC()."""
    mspec = np.log10(np.dot(X, np.asarray(fbank).T) + eps)
    return np.dot(mspec, frontend.dct2_ortho(mspec.shape[-1], 0, n_mfcc_feats).T)


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
