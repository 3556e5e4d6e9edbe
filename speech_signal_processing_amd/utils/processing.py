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


@functools.lru_cache(maxsize=8)
def _dft_tables(fs, L):
    """Frame sizes that are not powers of two: the length-L real DFT as a matrix ([cos | -sin] rows, float64 -> float32) for
    the MFMA dense kernel, and the reference's filterbank over L bins folded onto the L // 2 + 1 one-sided bins."""
    nb = L // 2 + 1
    k = np.arange(nb)[:, None]
    n = np.arange(L)[None, :]
    ang = 2.0 * np.pi * ((k * n) % L) / L
    Wt = np.concatenate([np.cos(ang), -np.sin(ang)]).astype(np.float32)      # (2 nb, L)
    bank, _ = frontend.mfccInitFilterBanks(fs, L)
    folded = np.array(bank[:, :nb])
    m = (L - 1) // 2                                                         # bins 1..m have a mirror image L - k
    folded[:, 1:m + 1] += bank[:, :L - m - 1:-1]
    return Wt, folded.astype(np.float32)


def _mfcc_any_size(x, fs, L, step):
    """utils/processing.py:110-144 for any frame size: enframe (GPU) -> DFT as one MFMA matrix product -> |.| / L ->
    filterbank + log10 + DCT (ssp_cepstrum)."""
    ctx = api.default_context()
    Wt, folded = _dft_tables(int(fs), int(L))
    frames = np.asarray(api.enframe(ctx, x, L, step, np.hamming(L)))          # (L, n_frames), the reference's layout
    rows = np.ascontiguousarray(frames.T)                                    # (n_frames, L): a re-layout, no arithmetic
    reim = api.dense_forward(ctx, rows, Wt)
    mag = api.spectrum_abs(ctx, reim, L // 2 + 1, scale=1.0 / L, power=1)
    dct = frontend.dct2_ortho(40, 0, 13)
    return np.asarray(api.cepstrum(ctx, mag, folded, dct, frontend.LOG_LOG10, frontend.FLOOR_ADD_EPS, eps), dtype=np.float64)


def MFCC(raw_signal, fs=8000, frameSize=512, step=256):
    """utils/processing.py:110-144 — (frames, 13) float64 MFCC matrix of one utterance, computed on the GPU (the fused
    kernel for power-of-two frame sizes, a DFT-matrix product on the matrix cores for any other size)."""
    L = int(frameSize)
    if L < 64 or L > 2048 or (L & (L - 1)):  # (the fused plans cover power-of-two frames up to 2048, ssp_mfcc_plan_create)
        return _mfcc_any_size(np.ascontiguousarray(np.asarray(raw_signal).reshape(-1), dtype=np.float32), int(fs), L, int(step))
    plan = _plan(int(fs), int(frameSize), int(step), 13, False)
    x, _ = api.flatten_signals([raw_signal])   # (int16 PCM — utils/tools.py:45-47 — is widened on the device)
    seg = api.Segments.from_lengths(plan.ctx, [x.shape[0]])
    feats = plan.run(x, seg)
    return np.asarray(feats, dtype=np.float64)


def MFCC_batch(signals, fs=8000, frameSize=512, step=256):
    """Batched form of MFCC(): list of 1-D signals -> list of (frames_i, 13) float64 (one kernel launch)."""
    sig = [np.asarray(s, dtype=np.float32).reshape(-1) for s in signals]
    L = int(frameSize)
    if L < 64 or L > 2048 or (L & (L - 1)):  # (the fused plans cover power-of-two frames up to 2048, ssp_mfcc_plan_create)
        return [_mfcc_any_size(s, int(fs), L, int(step)) for s in sig]
    plan = _plan(int(fs), int(frameSize), int(step), 13, False)
    flat, lens = api.flatten_signals(signals)   # (int16 PCM goes to the device as it is)
    seg = api.Segments.from_lengths(plan.ctx, lens)
    fseg = plan.frame_segments(seg)
    feats = np.asarray(plan.run(flat, seg, fseg), dtype=np.float64)
    return [feats[fseg.offsets[i]:fseg.offsets[i + 1]] for i in range(len(sig))]
