"""Cross-check of the oracle's restatement of the dialects whose third-party source is NOT in this image (librosa: MFCC_DTW.py:28-31;
sidekit's mel edges: GMM_UBM.py:89) against an INDEPENDENT implementation that is: `transformers.audio_utils`, whose mel filter bank,
spectrogram and dB conversion are documented as librosa-equivalent.  This corroborates the restatement; it does not pin it to the
reference (the reference holds no fixture for these dialects and librosa / sidekit cannot be imported here): parity for rows a6 / a7 stays
"unpinned" (DESIGN.md 2).  CPU only."""
import numpy as np
import pytest

au = pytest.importorskip("transformers.audio_utils")

from oracle import ref_cpu as O  # noqa: E402


def _signal(seed, n, fs):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    x = 0.4 * np.sin(2 * np.pi * 220.0 * t) + 0.2 * np.sin(2 * np.pi * 1370.0 * t + 0.3) + 0.05 * rng.standard_normal(n)
    return (x * np.hanning(n) ** 0.25).astype(np.float32)


@pytest.mark.parametrize("sr,n_fft,n_mels", [(8000, 2048, 128), (16000, 2048, 128), (16000, 512, 40)])
def test_librosa_mel_bank_vs_transformers(sr, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels, htk=False, norm='slaney') as the oracle restates it == mel_filter_bank(norm='slaney',
    mel_scale='slaney') to rounding (both float64)."""
    ours = O.librosa_mel_filters(sr, n_fft, n_mels)
    theirs = au.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=sr / 2.0,
                                sampling_rate=sr, norm="slaney", mel_scale="slaney").T
    assert ours.shape == theirs.shape
    assert np.abs(ours - theirs).max() <= 1e-12 * max(1.0, np.abs(theirs).max())


def test_slaney_and_htk_mel_scales_vs_transformers():
    """the two mel scales the dialects use: Slaney (librosa default) and HTK (sidekit's hz2mel / mel2hz: 2595 log10(1 + f / 700))."""
    f = np.concatenate([np.linspace(0.0, 8000.0, 161), [999.9, 1000.0, 1000.1]])
    assert np.allclose(O.slaney_hz_to_mel(f), au.hertz_to_mel(f, mel_scale="slaney"), rtol=1e-13, atol=1e-12)
    m = O.slaney_hz_to_mel(f)
    assert np.allclose(O.slaney_mel_to_hz(m), au.mel_to_hertz(m, mel_scale="slaney"), rtol=1e-13, atol=1e-9)
    assert np.allclose(O.hz2mel(f), au.hertz_to_mel(f, mel_scale="htk"), rtol=1e-13, atol=1e-12)
    mh = O.hz2mel(f)
    assert np.allclose(O.mel2hz(mh), au.mel_to_hertz(mh, mel_scale="htk"), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("seed,n", [(1, 8000), (2, 12345), (3, 4097)])
def test_librosa_mfcc_vs_transformers(seed, n):
    """MFCC_DTW.MFCC_lib = librosa.feature.mfcc(y, sr=8000, n_mfcc=13): centred reflect-padded frames of 2048 at hop 512, periodic Hann,
    power spectrum, Slaney mel bank, 10 log10(max(1e-10, S)) clamped at max - 80 dB, DCT-II ortho.  The independent chain: spectrogram(
    center, reflect, power 2, mel_filters, log_mel='dB', db_range 80) + scipy's DCT."""
    from scipy.fft import dct
    sr, n_fft, hop, n_mels, n_mfcc = 8000, 2048, 512, 128, 13
    x = _signal(seed, n, sr)
    ours = O.librosa_mfcc_flat(x, n_mfcc).reshape(-1, n_mfcc)
    fb = au.mel_filter_bank(1 + n_fft // 2, n_mels, 0.0, sr / 2.0, sr, norm="slaney", mel_scale="slaney")
    win = au.window_function(n_fft, "hann", periodic=True)
    logmel = au.spectrogram(x.astype(np.float64), win, frame_length=n_fft, hop_length=hop, fft_length=n_fft, power=2.0, center=True,
                            pad_mode="reflect", mel_filters=fb, mel_floor=1e-10, log_mel="dB", reference=1.0, min_value=1e-10,
                            db_range=80.0, dtype=np.float64)
    theirs = dct(logmel.T, type=2, norm="ortho", axis=1)[:, :n_mfcc]
    assert ours.shape == theirs.shape
    scale = float(np.abs(theirs).max())
    assert np.abs(ours - theirs).max() <= 1e-4 * scale, (np.abs(ours - theirs).max(), scale)


@pytest.mark.parametrize("seed,n,nwin", [(4, 16000, 0.025), (5, 48000, 0.025), (6, 7777, 0.025), (7, 16000, 0.018)])
def test_sidekit_front_end_vs_transformers(seed, n, nwin):
    """The sidekit dialect's front end as the oracle restates it (GMM_UBM.py:89, d_vector.py:91: frames without padding, floor((N - L) /
    hop) + 1 of them; per-FRAME pre-emphasis y[0] = x[0] - a x[0], y[n] = x[n] - a x[n - 1]; numpy.hanning; 512-point power spectrum)
    against `transformers.audio_utils.spectrogram(center=False, preemphasis=0.97, power=2)` — an independent implementation of the same
    Kaldi-style frame processing — and, with the oracle's own filterbank handed to it, the ln + DCT-II(ortho)[1:14] back end against
    scipy's DCT.  Corroborates the restatement's frame rule, pre-emphasis, window and transform; the filterbank's integer-bin triangles
    (sidekit's trfbank) have no independent counterpart here, and none of this pins row a7 to the reference."""
    from scipy.fft import dct
    fs = 16000
    x = _signal(seed, n, fs).astype(np.float64) * 3000.0           # (int16-scale amplitudes: what utils.tools.read hands over)
    cfg, w, fb, dctm = O.sidekit_tables(fs=fs, nwin=nwin)
    L, hop, n_fft = cfg["win_len"], cfg["hop"], cfg["n_fft"]
    # the oracle's power spectrum, step by step as mfcc_pipeline forms it
    frames = O.frame_matrix(x, cfg)
    prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)
    spec = np.fft.rfft((frames - cfg["preemph"] * prev) * w[None, :], n=n_fft, axis=1)
    P = spec.real ** 2 + spec.imag ** 2
    theirs = au.spectrogram(x, np.hanning(L), frame_length=L, hop_length=hop, fft_length=n_fft, power=2.0, center=False, preemphasis=0.97,
                            dtype=np.float64).T
    assert P.shape == theirs.shape == (O.num_frames(n, cfg), n_fft // 2 + 1)
    assert np.abs(P - theirs).max() <= 2e-6 * np.abs(theirs).max()      # (their transform runs in complex64: measured 6e-8)
    # back end: ln(P . fbank^T) -> DCT-II ortho, c0 dropped
    logmel = au.spectrogram(x, np.hanning(L), frame_length=L, hop_length=hop, fft_length=n_fft, power=2.0, center=False, preemphasis=0.97,
                            mel_filters=np.asarray(fb, dtype=np.float64).T, mel_floor=1e-300, log_mel="log", dtype=np.float64)
    ceps_theirs = dct(logmel.T, type=2, norm="ortho", axis=1)[:, 1:14]
    ours = O.mfcc_pipeline(x, cfg, w, fb, dctm)
    assert ours.shape == ceps_theirs.shape
    assert np.abs(ours - ceps_theirs).max() <= 1e-4 * max(1.0, np.abs(ceps_theirs).max())      # (the bar of the librosa check above)


def test_dense_network_vs_torch():
    """Row f2 (d_vector.py:171-189: Dense(256, relu) x 3 + Dense(256) on the 1274-d input, `spkModel.predict`): the oracle's restatement of
    a Keras Dense stack (kernel (d_in, units), y = act(x @ kernel + bias)) against torch.nn.functional.linear in float64 — an independent
    implementation of the same layer (torch keeps the weight as (units, d_in): the kernel's transpose).  Keras itself is absent."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(12)
    dims = [1274, 256, 256, 256, 256]
    layers = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i]), 0.1 * rng.standard_normal(dims[i + 1]), "relu" if i < 3 else None) for i in range(4)]
    X = rng.standard_normal((64, dims[0]))
    ours = O.dense_net_forward(X, layers)
    h = torch.from_numpy(X)
    for W, b, act in layers:
        h = torch.nn.functional.linear(h, torch.from_numpy(W).T.contiguous(), torch.from_numpy(b))
        if act == "relu":
            h = torch.relu(h)
    assert np.abs(ours - h.numpy()).max() <= 1e-12 * max(1.0, float(h.abs().max()))


def test_dtw_recurrence_vs_plain_recursion():
    """Row f4 (MFCC_DTW.py:57-108: accelerated_dtw(x, y, 'euclidean') on flattened sequences): the oracle's anti-diagonal numpy sweep of
    the dtw package's recurrence D[i, j] = |x_i - y_j| + min(D[i-1, j-1], D[i-1, j], D[i, j-1]) against a memoised top-down recursion
    written from the textbook definition (a different evaluation order, no shared code); the package itself is absent."""
    import functools
    rng = np.random.default_rng(13)
    for n, m in ((1, 1), (1, 7), (9, 1), (13, 29), (40, 40)):
        x, y = rng.standard_normal(n), rng.standard_normal(m)

        @functools.lru_cache(maxsize=None)
        def D(i, j):
            c = abs(x[i] - y[j])
            if i == 0 and j == 0:
                return c
            best = float("inf")
            if i > 0 and j > 0:
                best = min(best, D(i - 1, j - 1))
            if i > 0:
                best = min(best, D(i - 1, j))
            if j > 0:
                best = min(best, D(i, j - 1))
            return c + best
        ref = D(n - 1, m - 1)
        assert abs(O.dtw_distance(x, y) - ref) <= 1e-12 * max(1.0, ref), (n, m)
