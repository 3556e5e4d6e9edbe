"""Host-side MFCC dialect tables (window, filterbank, DCT) and the cfg block of include/ssp.h.

Three presets mirror the three MFCC variants the reference uses:

* ``preset_inrepo``  — utils/processing.py:19-144 (Hamming, |FFT|/L, 40 talkbox triangles over all nfft bins —
  folded here onto the L/2+1 real-FFT bins —, log10(.+1e-8), DCT-II ortho, c0..c12, ceil framing with zero pad).
* ``preset_sidekit`` — sidekit.frontend.features.mfcc as called at GMM_UBM.py:89, d_vector.py:91,
  UI/GMM_UBM_GUI.py:91 (25 ms / 10 ms, per-frame pre-emphasis 0.97, 512-point power spectrum, 24 HTK-mel
  triangles 100..8000 Hz, ln, DCT-II ortho, c1..c13).  sidekit is absent from this image: restated from its
  published algorithm — parity unpinned.
* ``preset_librosa`` — librosa.feature.mfcc(y, n_mfcc=13, sr=8000) as called at MFCC_DTW.py:29 (n_fft 2048,
  hop 512, periodic Hann, centred reflect padding, 128 Slaney mel bands, power_to_db(top_db=80), DCT-II ortho).
  Parity unpinned for the same reason.

Only table construction happens here (numpy, float64 -> float32); every per-sample operation runs on the GPU.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

FRAME_FLOOR, FRAME_CEIL_ZEROPAD, FRAME_CENTER_REFLECT = 0, 1, 2
LOG_LN, LOG_LOG10, LOG_10LOG10 = 0, 1, 2
FLOOR_NONE, FLOOR_ADD_EPS, FLOOR_MAX_EPS = 0, 1, 2


@dataclasses.dataclass
class MfccConfig:
    sample_rate: int
    win_len: int
    hop: int
    n_fft: int
    n_filt: int
    n_ceps: int
    frame_mode: int = FRAME_FLOOR
    preemph_mode: int = 0
    preemph: float = 0.0
    spec_power: int = 2
    spec_scale: float = 1.0
    log_mode: int = LOG_LN
    floor_mode: int = FLOOR_NONE
    eps: float = 0.0
    top_db: float = -1.0
    delta_order: int = 0
    delta_N: int = 2
    cmvn: int = 0

    @property
    def d_out(self) -> int:
        return self.n_ceps * (1 + self.delta_order)

    def num_frames(self, n_samples: int) -> int:
        """Same rule as ssp_mfcc_num_frames (csrc/mfcc_plan.hip:frames_for)."""
        if self.frame_mode == FRAME_FLOOR:
            return 0 if n_samples < self.win_len else (n_samples - self.win_len) // self.hop + 1
        if self.frame_mode == FRAME_CEIL_ZEROPAD:
            return -(-n_samples // self.hop)
        return 0 if n_samples <= 0 else 1 + n_samples // self.hop

    def as_dict(self) -> dict:
        return dataclasses.asdict(self)


@dataclasses.dataclass
class MfccTables:
    cfg: MfccConfig
    window: np.ndarray   # (win_len,) float32
    fbank: np.ndarray    # (n_filt, n_fft/2+1) float32
    dct: np.ndarray      # (n_ceps, n_filt) float32


def dct2_ortho(n_in: int, first: int, count: int) -> np.ndarray:
    """Rows first..first+count-1 of the orthonormal DCT-II matrix (scipy.fftpack.dct type=2 norm='ortho')."""
    q = np.arange(first, first + count, dtype=np.float64).reshape(-1, 1)
    j = np.arange(n_in, dtype=np.float64).reshape(1, -1)
    mat = np.cos(np.pi * q * (2.0 * j + 1.0) / (2.0 * n_in)) * math.sqrt(2.0 / n_in)
    mat[q[:, 0] == 0] *= math.sqrt(0.5)
    return mat


def _triangle_bank(edges_hz: np.ndarray, n_bins_total: int, n_fft: int, fs: float, drop_last_falling: bool) -> np.ndarray:
    """talkbox-style triangles used by both utils/processing.py:64-86 and sidekit's trfbank: bin index ranges are
    floor(f * nfft / fs) + 1 on each side, heights 2 / (f[i+2] - f[i])."""
    n_filt = len(edges_hz) - 2
    bank = np.zeros((n_filt, n_bins_total), dtype=np.float64)
    bin_hz = np.arange(n_fft, dtype=np.float64) * (fs / n_fft)
    pos = np.floor(edges_hz * n_fft / fs).astype(np.int64) + 1
    for i in range(n_filt):
        lo, ce, hi = edges_hz[i], edges_hz[i + 1], edges_hz[i + 2]
        height = 2.0 / (hi - lo)
        rise = np.arange(pos[i], pos[i + 1])
        fall_end = min(pos[i + 2], n_fft) if drop_last_falling else pos[i + 2]
        fall = np.arange(pos[i + 1], fall_end)
        if drop_last_falling:
            fall = fall[:-1]
        bank[i, rise] = height / (ce - lo) * (bin_hz[rise] - lo)
        bank[i, fall] = height / (hi - ce) * (hi - bin_hz[fall])
    return bank


def mfccInitFilterBanks(fs, nfft):
    """Same return as utils/processing.py:42-88: (fbank (40, nfft), freqs (42,)) — 13 linear + 27 log-spaced
    triangles laid over ALL nfft bins, i.e. including the mirrored upper half of the spectrum."""
    freqs = np.empty(42)
    freqs[:13] = 133.33 + (200.0 / 3.0) * np.arange(13)
    freqs[13:] = freqs[12] * 1.0711703 ** np.arange(1, 30)
    return _triangle_bank(freqs, nfft, nfft, float(fs), drop_last_falling=False), freqs


def fold_to_rfft_bins(bank_full: np.ndarray) -> np.ndarray:
    """|X[k]| = |X[L-k]| for a real frame, so a bank over k = 0..L-1 equals a bank over k = 0..L/2 with the
    mirrored weights added in (bins 0 and L/2 have no mirror image)."""
    L = bank_full.shape[1]
    half = L // 2
    folded = np.array(bank_full[:, : half + 1])
    folded[:, 1:half] += bank_full[:, :half:-1]
    return folded


def preset_inrepo(fs=8000, frameSize=512, step=256, n_ceps=13, delta_order=0, cmvn=0) -> MfccTables:
    L = int(frameSize)
    cfg = MfccConfig(sample_rate=int(fs), win_len=L, hop=int(step), n_fft=L, n_filt=40, n_ceps=n_ceps,
                     frame_mode=FRAME_CEIL_ZEROPAD, preemph_mode=0, preemph=0.0, spec_power=1, spec_scale=1.0 / L,
                     log_mode=LOG_LOG10, floor_mode=FLOOR_ADD_EPS, eps=1e-8, top_db=-1.0,
                     delta_order=delta_order, delta_N=2, cmvn=cmvn)
    window = np.hamming(L)  # == scipy.signal.windows.hamming(L) (symmetric), utils/processing.py:30
    bank, _ = mfccInitFilterBanks(fs, L)
    return MfccTables(cfg, window.astype(np.float32), fold_to_rfft_bins(bank).astype(np.float32),
                      dct2_ortho(40, 0, n_ceps).astype(np.float32))


def _htk_mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def _htk_mel_inv(m):
    return 700.0 * (np.power(10.0, np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def preset_sidekit(fs=16000, nwin=0.025, shift=0.01, nlogfilt=24, nceps=13, lowfreq=100.0, maxfreq=8000.0,
                   prefac=0.97, window="hanning", delta_order=0, cmvn=0) -> MfccTables:
    win_len = int(round(nwin * fs))
    hop = int(shift * fs)
    n_fft = 1 << int(math.ceil(math.log2(win_len)))
    maxfreq = min(float(maxfreq), fs / 2.0)
    mel_edges = np.linspace(_htk_mel(lowfreq), _htk_mel(maxfreq), nlogfilt + 2)
    bank = _triangle_bank(_htk_mel_inv(mel_edges), n_fft // 2 + 1, n_fft, float(fs), drop_last_falling=True)
    cfg = MfccConfig(sample_rate=int(fs), win_len=win_len, hop=hop, n_fft=n_fft, n_filt=nlogfilt, n_ceps=nceps,
                     frame_mode=FRAME_FLOOR, preemph_mode=1, preemph=float(prefac), spec_power=2, spec_scale=1.0,
                     log_mode=LOG_LN, floor_mode=FLOOR_NONE, eps=0.0, top_db=-1.0,
                     delta_order=delta_order, delta_N=2, cmvn=cmvn)
    if window not in ("hanning", "hamming"):
        raise ValueError("window must be 'hanning' or 'hamming'")
    w = np.hanning(win_len) if window == "hanning" else np.hamming(win_len)
    return MfccTables(cfg, w.astype(np.float32), bank.astype(np.float32),
                      dct2_ortho(nlogfilt, 1, nceps).astype(np.float32))  # c0 dropped


def hz2bark(f):
    return 6.0 * np.arcsinh(np.asarray(f, dtype=np.float64) / 600.0)


def plp_num_bands(fs) -> int:
    """sidekit / rastamat audspec default: ceil(hz2bark(fs / 2)) + 1 critical bands (21 at 16 kHz, 17 at 8 kHz)."""
    return int(math.ceil(float(hz2bark(fs / 2.0)))) + 1


def bark_filterbank(n_fft: int, fs: float, nfilts: int, width: float = 1.0, minfreq: float = 0.0, maxfreq: float = 8000.0) -> np.ndarray:
    """rastamat fft2barkmx over the n_fft/2+1 rfft bins: 10^(min(0, min(hif, -2.5 lof) / width)) around Bark-spaced centres."""
    maxfreq = min(float(maxfreq), fs / 2.0)
    min_bark = float(hz2bark(minfreq))
    nyqbark = float(hz2bark(maxfreq)) - min_bark
    step = nyqbark / (nfilts - 1)
    binbarks = hz2bark(np.arange(n_fft // 2 + 1) * fs / float(n_fft))
    wts = np.zeros((nfilts, n_fft // 2 + 1))
    for i in range(nfilts):
        mid = min_bark + i * step
        wts[i] = 10.0 ** (np.minimum(0.0, np.minimum(binbarks - mid + 0.5, -2.5 * (binbarks - mid - 0.5)) / width))
    return wts


def preset_sidekit_plp(fs=16000, nwin=0.025, shift=0.01, prefac=0.97) -> MfccTables:
    """Front end of sidekit's plp up to ln(audspec(power_spectrum)): the sidekit framing / per-frame pre-emphasis / hanning /
    power spectrum of ``preset_sidekit`` into Bark bands, natural log, identity "DCT" (n_ceps = n_filt = bands).  The back end
    (RASTA ... lifter) is ``api.plp_post``.  sidekit's source is absent: restated from the rastamat algorithm it ports."""
    win_len = int(round(nwin * fs))
    hop = int(shift * fs)
    n_fft = 1 << int(math.ceil(math.log2(win_len)))
    nb = plp_num_bands(fs)
    cfg = MfccConfig(sample_rate=int(fs), win_len=win_len, hop=hop, n_fft=n_fft, n_filt=nb, n_ceps=nb,
                     frame_mode=FRAME_FLOOR, preemph_mode=1, preemph=float(prefac), spec_power=2, spec_scale=1.0,
                     log_mode=LOG_LN, floor_mode=FLOOR_NONE, eps=0.0, top_db=-1.0, delta_order=0, delta_N=2, cmvn=0)
    return MfccTables(cfg, np.hanning(win_len).astype(np.float32), bark_filterbank(n_fft, float(fs), nb).astype(np.float32),
                      np.eye(nb, dtype=np.float32))


def _slaney_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * (3.0 / 200.0)
    log_part = 15.0 + np.log(np.maximum(f, 1.0) / 1000.0) * (27.0 / math.log(6.4))
    return np.where(f >= 1000.0, log_part, lin)


def _slaney_mel_inv(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * (math.log(6.4) / 27.0)), m * (200.0 / 3.0))


def preset_librosa(sr=8000, n_mfcc=13, n_fft=2048, hop=512, n_mels=128, top_db=80.0) -> MfccTables:
    bins_hz = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = _slaney_mel_inv(np.linspace(_slaney_mel(0.0), _slaney_mel(sr / 2.0), n_mels + 2))
    up = (bins_hz[None, :] - edges[:-2, None]) / (edges[1:-1] - edges[:-2])[:, None]
    down = (edges[2:, None] - bins_hz[None, :]) / (edges[2:] - edges[1:-1])[:, None]
    bank = np.clip(np.minimum(up, down), 0.0, None) * (2.0 / (edges[2:] - edges[:-2]))[:, None]
    cfg = MfccConfig(sample_rate=int(sr), win_len=n_fft, hop=hop, n_fft=n_fft, n_filt=n_mels, n_ceps=n_mfcc,
                     frame_mode=FRAME_CENTER_REFLECT, preemph_mode=0, preemph=0.0, spec_power=2, spec_scale=1.0,
                     log_mode=LOG_10LOG10, floor_mode=FLOOR_MAX_EPS, eps=1e-10, top_db=float(top_db),
                     delta_order=0, delta_N=2, cmvn=0)
    n = np.arange(n_fft, dtype=np.float64)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)  # periodic Hann
    return MfccTables(cfg, window.astype(np.float32), bank.astype(np.float32),
                      dct2_ortho(n_mels, 0, n_mfcc).astype(np.float32))
