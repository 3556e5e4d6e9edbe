"""Object layer over the C-ABI (include/ssp.h): Context, Segments, MfccPlan, GmmScorer, cosine_identify.

Bulk arrays may be numpy arrays (host pointers, the library stages them) or torch CUDA tensors (device pointers,
results stay on the device).  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import contextlib
import os

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .frontend import MfccConfig, MfccTables


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _as_f32(x, name):
    """-> (array-like kept alive, raw address, where)"""
    if _is_torch(x):
        import torch
        if not x.is_cuda:
            raise ValueError("%s: torch tensors must live on the GPU (pass numpy arrays for host data)" % name)
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        return x, x.data_ptr(), _lib.DEVICE
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a, a.ctypes.data, _lib.HOST


def _as_samples(x, name):
    """-> (array kept alive, raw address, where, is_int16): int16 PCM (what utils.tools.read / scipy.io.wavfile return) stays int16 —
    half the bytes cross PCIe and the device widens (ssp_mfcc_run_i16); everything else goes as float32"""
    if _is_torch(x):
        import torch
        if x.is_cuda and x.dtype == torch.int16:
            x = x.contiguous()
            return x, x.data_ptr(), _lib.DEVICE, True
    elif isinstance(x, np.ndarray) and x.dtype == np.int16:
        a = np.ascontiguousarray(x)
        return a, a.ctypes.data, _lib.HOST, True
    return _as_f32(x, name) + (False,)


class Context:
    """One HIP device + one stream (ssp_ctx).  stream=None: the library owns a stream; an int is a borrowed
    hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; 0 = the HIP default stream)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.ssp_ctx_create(int(device), C.c_void_p(stream or 0), 0 if stream is None else 1, C.byref(h)))
        self._h = h
        self.device = int(device)
        self.stream = stream

    @classmethod
    def for_torch(cls, device: Optional[int] = None) -> "Context":
        import torch
        dev = torch.cuda.current_device() if device is None else int(device)
        return cls(dev, torch.cuda.current_stream(dev).cuda_stream)

    def sync(self):
        _lib.check(self._lib.ssp_ctx_sync(self._h))

    def calibrate(self, target_ms: float = 20.0) -> dict:
        """what this box sustains on two textbook loads (ssp_calibrate): a float4 copy (GB/s) and packed-fp32 FMA chains (TFLOP/s), each
        run for about target_ms on this context's stream — bench.py divides its headline by them to compare boxes"""
        v = [C.c_double() for _ in range(4)]
        _lib.check(self._lib.ssp_calibrate(self._h, float(target_ms), *[C.byref(x) for x in v]))
        return {"copy_gbs": v[0].value, "fma_tflops": v[1].value, "copy_ms": v[2].value, "fma_ms": v[3].value}

    @contextlib.contextmanager
    def _ordered(self, where):
        """Stream ordering around a call that takes device pointers.  A context that BORROWS torch's stream needs none (the kernels
        are queued behind the producers of their inputs and ahead of the consumers of their outputs).  A context that owns its
        stream is ordered against nothing torch does, so around a device-pointer call its stream first waits for torch's current
        stream and torch's current stream then waits for it — two events, no host wait (ssp_ctx_wait_stream / _signal_stream): calls
        on different owned-stream contexts overlap, and the host runs ahead as it does with torch's own kernels."""
        own = where == _lib.DEVICE and self.stream is None
        ts = None
        host_sync = own and bool(os.environ.get("SSP_ORDER_SYNC"))  # (diagnostic: host waits on both sides, as before round 3)
        if own:
            import torch
            if host_sync:
                torch.cuda.current_stream(self.device).synchronize()
            else:
                ts = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                _lib.check(self._lib.ssp_ctx_wait_stream(self._h, ts))
        try:
            yield
        finally:
            if own:
                if host_sync:
                    self.sync()
                else:
                    _lib.check(self._lib.ssp_ctx_signal_stream(self._h, ts))

    # ---- collectives (include/ssp.h: ssp_comm_*): one process and one Context per GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 opaque bytes made on rank 0; ship them to every rank (any transport) and pass them to comm_init."""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.load().ssp_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, nranks: int, unique_id: bytes):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _lib.COMM_ID_BYTES)
        _lib.check(self._lib.ssp_comm_init(self._h, int(rank), int(nranks), C.c_char_p(unique_id)))

    def comm_destroy(self):
        _lib.check(self._lib.ssp_comm_destroy(self._h))

    def comm_info(self):
        r, n = C.c_int(), C.c_int()
        _lib.check(self._lib.ssp_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def allgather(self, local):
        """All-gather of equally shaped device tensors: returns [nranks * local.shape[0], ...] in rank order (RCCL over xGMI)."""
        import torch
        local = local.contiguous()
        _, n = self.comm_info()
        out = torch.empty((n * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        with self._ordered(_lib.DEVICE):
            _lib.check(self._lib.ssp_allgather(self._h, C.c_void_p(local.data_ptr()), C.c_void_p(out.data_ptr()),
                                               C.c_size_t(local.numel() * local.element_size())))
        return out

    def allreduce_sum_(self, t):
        """In-place sum over ranks of a float32 / float64 device tensor."""
        import torch
        if t.dtype not in (torch.float32, torch.float64) or not t.is_contiguous():
            raise ValueError("allreduce_sum_ takes a contiguous float32 / float64 device tensor")
        with self._ordered(_lib.DEVICE):
            _lib.check(self._lib.ssp_allreduce_sum(self._h, C.c_void_p(t.data_ptr()), C.c_size_t(t.numel()), int(t.dtype == torch.float64)))
        return t

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ssp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _empty(self, shape, where, dtype="float32"):
        if where == _lib.DEVICE:
            import torch
            return torch.empty(shape, dtype=getattr(torch, dtype), device="cuda:%d" % self.device)
        return np.empty(shape, dtype=dtype)


_default_ctx = {}


def default_context(device: int = 0, torch_stream: bool = False) -> Context:
    key = (device, torch_stream)
    if key not in _default_ctx:
        _default_ctx[key] = Context.for_torch(device) if torch_stream else Context(device)
    return _default_ctx[key]


class Segments:
    """Per-utterance offsets (ssp_segments): int64[n+1], uploaded once."""

    def __init__(self, ctx: Context, offsets=None, _handle=None):
        self.ctx = ctx
        self._lib = ctx._lib
        if _handle is not None:
            self._h = _handle
        else:
            off = np.ascontiguousarray(offsets, dtype=np.int64)
            if off.ndim != 1 or off.size < 1:
                raise ValueError("offsets must be a 1-D int64 array of length n+1")
            h = C.c_void_p()
            _lib.check(self._lib.ssp_segments_create(ctx._h, off.ctypes.data_as(C.POINTER(C.c_int64)), off.size - 1, C.byref(h)))
            self._h = h
        n, tot = C.c_int64(), C.c_int64()
        _lib.check(self._lib.ssp_segments_count(self._h, C.byref(n), C.byref(tot)))
        self.n = n.value
        out = np.empty(self.n + 1, dtype=np.int64)
        _lib.check(self._lib.ssp_segments_read(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))))
        self.offsets = out

    @classmethod
    def from_lengths(cls, ctx: Context, lengths: Sequence[int]) -> "Segments":
        off = np.zeros(len(lengths) + 1, dtype=np.int64)
        np.cumsum(np.asarray(lengths, dtype=np.int64), out=off[1:])
        return cls(ctx, off)

    @property
    def total(self) -> int:
        return int(self.offsets[-1])

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ssp_segments_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _cfg_struct(cfg: MfccConfig) -> _lib.ssp_mfcc_cfg:
    s = _lib.ssp_mfcc_cfg()
    for k, v in cfg.as_dict().items():
        setattr(s, k, v)
    return s


class MfccPlan:
    """Fused MFCC (+delta, +CMVN) pass for one dialect (ssp_mfcc_plan)."""

    def __init__(self, ctx: Context, tables: MfccTables):
        self.ctx = ctx
        self._lib = ctx._lib
        self.cfg = tables.cfg
        nb = self.cfg.n_fft // 2 + 1
        w = np.ascontiguousarray(tables.window, dtype=np.float32)
        fb = np.ascontiguousarray(tables.fbank, dtype=np.float32)
        dct = np.ascontiguousarray(tables.dct, dtype=np.float32)
        if w.shape != (self.cfg.win_len,) or fb.shape != (self.cfg.n_filt, nb) or dct.shape != (self.cfg.n_ceps, self.cfg.n_filt):
            raise ValueError("table shapes do not match the cfg")
        self._cs = _cfg_struct(self.cfg)
        h = C.c_void_p()
        _lib.check(self._lib.ssp_mfcc_plan_create(ctx._h, C.byref(self._cs), w.ctypes.data, fb.ctypes.data, dct.ctypes.data, C.byref(h)))
        self._h = h

    @property
    def d_out(self) -> int:
        return self.cfg.d_out

    def set_reproducible(self, on: bool = True) -> "MfccPlan":
        """SSP_MFCC_REPRODUCIBLE (include/ssp.h): an utterance's float32 bits no longer depend on the batch or the machine."""
        _lib.check(self._lib.ssp_mfcc_plan_set_flags(self._h, 1 if on else 0))
        return self

    def num_frames(self, n_samples: int) -> int:
        out = C.c_int64()
        _lib.check(self._lib.ssp_mfcc_num_frames(C.byref(self._cs), int(n_samples), C.byref(out)))
        return out.value

    def frame_segments(self, sample_seg: Segments) -> Segments:
        h = C.c_void_p()
        _lib.check(self._lib.ssp_mfcc_frame_segments(self._h, sample_seg._h, C.byref(h)))
        return Segments(self.ctx, _handle=h)

    def run(self, samples, sample_seg: Segments, frame_seg: Optional[Segments] = None, out=None, variant: int = 0,
            timing: bool = False):
        """samples: float32[total samples] or int16 PCM (numpy -> host path, torch cuda -> device path).  Large host batches run as a
        copy / compute / copy-back pipeline inside the library (include/ssp.h, ssp_mfcc_run); pinned arrays (`pinned_empty`) get the
        full PCIe rate.  Returns feats (total_frames, d_out) [and kernel milliseconds when timing=True]."""
        if frame_seg is None:
            frame_seg = self.frame_segments(sample_seg)
        keep, ptr, where, is_i16 = _as_samples(samples, "samples")
        if keep.ndim != 1 and not (keep.ndim == 2 and keep.shape[0] * keep.shape[1] == sample_seg.total):
            raise ValueError("samples must be a flat array of all utterances' samples")
        if int(np.prod(keep.shape)) < sample_seg.total:
            raise ValueError("samples shorter than the segment table")
        if out is None:
            out = self.ctx._empty((frame_seg.total, self.d_out), where)
        okeep, optr, owhere = _as_f32(out, "out")
        if owhere != where or okeep is not out:
            raise ValueError("out must be a contiguous float32 array of the same kind as samples")
        ms = C.c_float(0.0)
        with self.ctx._ordered(where):
            fn = self._lib.ssp_mfcc_run_i16 if is_i16 else self._lib.ssp_mfcc_run
            _lib.check(fn(self._h, sample_seg._h, frame_seg._h, ptr, optr, where, int(variant), C.byref(ms) if timing else None))
        return (out, ms.value) if timing else out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ssp_mfcc_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flatten_signals(signals):
    """a list of 1-D utterances -> (one flat array, their lengths).  When every utterance is int16 PCM (utils.tools.read /
    scipy.io.wavfile, utils/tools.py:45-47) the flat array stays int16: no widening pass on the host, half the bytes over PCIe, the
    device widens (ssp_mfcc_run_i16).  Anything else is converted to float32 as before."""
    sig = [np.asarray(s).reshape(-1) for s in signals]
    if sig and all(s.dtype == np.int16 for s in sig):
        return (sig[0] if len(sig) == 1 else np.concatenate(sig)), [s.shape[0] for s in sig]
    sig = [np.asarray(s, dtype=np.float32) for s in sig]
    return (np.concatenate(sig) if sig else np.zeros(0, dtype=np.float32)), [s.shape[0] for s in sig]


def pinned_empty(shape, dtype=np.float32):
    """a numpy array over page-locked host memory (torch's pinned allocator = hipHostMalloc): host-fed calls copy from / to such arrays
    asynchronously and at the full PCIe rate; the array keeps its tensor alive"""
    import torch
    t = torch.empty(tuple(np.atleast_1d(shape)), dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True)
    return t.numpy()   # (the array's base keeps the pinned storage alive)


def enframe(ctx: Context, samples, frame_size: int, step: int, window):
    """(frame_size, ceil(n/step)) float32 frames, zero padded tail, times window — utils/processing.py:19-38."""
    keep, ptr, where = _as_f32(samples, "samples")
    if keep.ndim != 1:
        raise ValueError("samples must be 1-D")
    n = int(keep.shape[0])
    n_frames = -(-n // step)
    w = np.ascontiguousarray(window, dtype=np.float32)
    if w.shape != (frame_size,):
        raise ValueError("window must have frame_size taps")
    out = ctx._empty((frame_size, n_frames), where)
    optr = out.data_ptr() if where == _lib.DEVICE else out.ctypes.data
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_enframe(ctx._h, ptr, n, int(frame_size), int(step), w.ctypes.data, optr, where, None))
    return out


def cepstrum(ctx: Context, X, fbank, dct, log_mode: int, floor_mode: int, eps: float):
    """DCT(log(X . fbank^T)) per row of X — utils/processing.py:91-107 on a batch of spectra."""
    keep, ptr, where = _as_f32(X, "X")
    if keep.ndim != 2:
        raise ValueError("X must be (rows, bins)")
    fb = np.ascontiguousarray(fbank, dtype=np.float32)
    dc = np.ascontiguousarray(dct, dtype=np.float32)
    if fb.ndim != 2 or fb.shape[1] != keep.shape[1] or dc.shape != (dc.shape[0], fb.shape[0]):
        raise ValueError("fbank must be (n_filt, bins) and dct (n_ceps, n_filt)")
    out = ctx._empty((int(keep.shape[0]), dc.shape[0]), where)
    optr = out.data_ptr() if where == _lib.DEVICE else out.ctypes.data
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_cepstrum(ctx._h, ptr, int(keep.shape[0]), int(keep.shape[1]), fb.ctypes.data, fb.shape[0], dc.ctypes.data,
                                         dc.shape[0], int(log_mode), int(floor_mode), float(eps), optr, where, None))
    return out


def spectrum_abs(ctx: Context, reim, n_bins: int, scale: float = 1.0, power: int = 1):
    """rows of [re | im] (rows, 2 n_bins) -> scale * |.| (power 1) or scale * |.|^2 (power 2), (rows, n_bins)."""
    keep, ptr, where = _as_f32(reim, "reim")
    if keep.ndim != 2 or keep.shape[1] != 2 * n_bins:
        raise ValueError("reim must be (rows, 2 * n_bins)")
    out = ctx._empty((int(keep.shape[0]), int(n_bins)), where)
    optr = out.data_ptr() if where == _lib.DEVICE else out.ctypes.data
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_spectrum_abs(ctx._h, ptr, int(keep.shape[0]), int(n_bins), float(scale), int(power), optr, where, None))
    return out


def delta_features(ctx: Context, feats, frame_seg: Segments, N: int = 2, timing: bool = False):
    keep, ptr, where = _as_f32(feats, "feats")
    if keep.ndim != 2:
        raise ValueError("feats must be (frames, dim)")
    out = ctx._empty(tuple(keep.shape), where)
    ms = C.c_float(0.0)
    optr = out.data_ptr() if where == _lib.DEVICE else out.ctypes.data
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_delta(ctx._h, ptr, frame_seg._h, int(keep.shape[1]), int(N), optr, where, C.byref(ms) if timing else None))
    return (out, ms.value) if timing else out


def cmvn_features(ctx: Context, feats, frame_seg: Segments, timing: bool = False):
    keep, ptr, where = _as_f32(feats, "feats")
    if keep.ndim != 2:
        raise ValueError("feats must be (frames, dim)")
    out = ctx._empty(tuple(keep.shape), where)
    ms = C.c_float(0.0)
    optr = out.data_ptr() if where == _lib.DEVICE else out.ctypes.data
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_cmvn(ctx._h, ptr, frame_seg._h, int(keep.shape[1]), optr, where, C.byref(ms) if timing else None))
    return (out, ms.value) if timing else out


def plp_post(ctx: Context, logspec, frame_seg: Segments, fmax_hz: float, plp_order: int = 13, rasta: bool = True, lift: float = 0.6,
             timing: bool = False):
    """Back end of sidekit's plp on the GPU (ssp_plp_post): (frames, bands) ln critical-band energies -> (frames, plp_order)
    cepstra (RASTA along each utterance of ``frame_seg``, equal loudness, ^0.33, autocorrelation, Levinson, LPC->cepstrum, lifter)."""
    keep, ptr, where = _as_f32(logspec, "logspec")
    if keep.ndim != 2:
        raise ValueError("logspec must be (frames, bands)")
    out = ctx._empty((int(keep.shape[0]), int(plp_order)), where)
    ms = C.c_float(0.0)
    optr = out.data_ptr() if where == _lib.DEVICE else out.ctypes.data
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_plp_post(ctx._h, ptr, frame_seg._h, int(keep.shape[1]), float(fmax_hz), int(plp_order), int(bool(rasta)),
                                         float(lift), optr, where, C.byref(ms) if timing else None))
    return (out, ms.value) if timing else out


def gmm_em_stats(ctx: "Context", weights, means, covars, feats, timing: bool = False) -> dict:
    """E step + M-step sums of ONE EM iteration of a diagonal GMM on the GPU (ssp_gmm_em_stats).
    weights (K,), means (K,D), covars (K,D) float64; feats (n, D) float32 (numpy or device tensor).
    Returns nk (K,), sx (K,D), sxx (K,D), loglik_sum (float) as float64."""
    w = np.ascontiguousarray(weights, dtype=np.float64)
    mu = np.ascontiguousarray(means, dtype=np.float64)
    cv = np.ascontiguousarray(covars, dtype=np.float64)
    if w.ndim != 1 or mu.ndim != 2 or mu.shape != cv.shape or mu.shape[0] != w.shape[0]:
        raise ValueError("expected weights (K,), means (K,D), covars (K,D)")
    K, D = mu.shape
    keep, ptr, where = _as_f32(feats, "feats")
    if keep.ndim != 2 or keep.shape[1] != D:
        raise ValueError("feats must be (frames, %d)" % D)
    nk = np.empty(K, dtype=np.float64)
    sx = np.empty((K, D), dtype=np.float64)
    sxx = np.empty((K, D), dtype=np.float64)
    ll = C.c_double(0.0)
    ms = C.c_float(0.0)
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_gmm_em_stats(ctx._h, K, D, w.ctypes.data, mu.ctypes.data, cv.ctypes.data, ptr, int(keep.shape[0]),
                                              nk.ctypes.data, sx.ctypes.data, sxx.ctypes.data, C.byref(ll), where,
                                              C.byref(ms) if timing else None))
    res = {"nk": nk, "sx": sx, "sxx": sxx, "loglik_sum": ll.value}
    if timing:
        res["kernel_ms"] = ms.value
    return res


class GmmScorer:
    """Packed diagonal GMMs (ssp_gmm).  weights (M,K), means (M,K,D), covars (M,K,D) float64.
    has_ubm: model 0 is the UBM (GMM_UBM.py:169-170); scores/argmax are then taken against it."""

    def __init__(self, ctx: Context, weights, means, covars, has_ubm: bool = True):
        self.ctx = ctx
        self._lib = ctx._lib
        w = np.ascontiguousarray(weights, dtype=np.float64)
        mu = np.ascontiguousarray(means, dtype=np.float64)
        cv = np.ascontiguousarray(covars, dtype=np.float64)
        if w.ndim != 2 or mu.ndim != 3 or mu.shape != cv.shape or mu.shape[:2] != w.shape:
            raise ValueError("expected weights (M,K), means (M,K,D), covars (M,K,D)")
        self.n_models, self.K, self.D = mu.shape
        self.has_ubm = bool(has_ubm)
        h = C.c_void_p()
        _lib.check(self._lib.ssp_gmm_pack(ctx._h, self.n_models, self.K, self.D, w.ctypes.data, mu.ctypes.data, cv.ctypes.data,
                                           1 if has_ubm else 0, C.byref(h)))
        self._h = h

    @classmethod
    def from_sklearn(cls, ctx: Context, models: Sequence, ubm=None) -> "GmmScorer":
        """models / ubm: fitted sklearn GaussianMixture(covariance_type='diag') objects (duck-typed:
        weights_, means_, covariances_), all with the same K and D — the lists GMM_UBM.py:141-146 pickles."""
        allm = ([ubm] if ubm is not None else []) + list(models)
        for g in allm:
            if getattr(g, "covariance_type", "diag") != "diag":
                raise ValueError("only covariance_type='diag' models are supported")
        return cls(ctx, np.stack([g.weights_ for g in allm]), np.stack([g.means_ for g in allm]),
                   np.stack([g.covariances_ for g in allm]), has_ubm=ubm is not None)

    def score(self, feats, frame_seg: Segments, loglik: bool = False, scores: bool = True, argmax: bool = True,
              precision: int = 0, timing: bool = False) -> dict:
        """Returns a dict with the requested arrays:
        loglik (M, F) per-frame log-likelihood per model (= score_samples), scores (U, M) mean log-likelihood
        (= GaussianMixture.score), argmax (U,) int32 over speaker models of score - score_ubm.
        precision: 0 exact-fp32 MFMA | 1 bf16x3 split MFMA with the close calls (top-2 margin inside the split-precision error
        BOUND) scored again in fp32, so the arg-max equals precision 0's (``last_rescored`` = how many) | 2 bf16x3 alone | 3 as 1
        with the calibrated, heuristic band (about 100 times narrower than the bound: far fewer utterances scored twice) | 4 or
        "auto": precision 1's guarantee at the cost of the cheaper of 1 and 0 — a pilot on the first ~2 % of the utterances prices the
        re-scoring; ``last_auto`` says what it chose (include/ssp.h, ssp_gmm_score)."""
        if precision == "auto":
            precision = 4
        keep, ptr, where = _as_f32(feats, "feats")
        if keep.ndim != 2 or keep.shape[1] != self.D:
            raise ValueError("feats must be (frames, %d)" % self.D)
        if keep.shape[0] < frame_seg.total:
            raise ValueError("feats has fewer rows than the frame segments cover")
        F, U = frame_seg.total, frame_seg.n
        res = {}
        ll = self.ctx._empty((self.n_models, F), where) if loglik else None
        sc = self.ctx._empty((U, self.n_models), where) if scores else None
        am = self.ctx._empty((U,), where, "int32") if argmax else None

        def p(x):
            if x is None:
                return None
            return x.data_ptr() if where == _lib.DEVICE else x.ctypes.data
        ms = C.c_float(0.0)
        with self.ctx._ordered(where):
            _lib.check(self._lib.ssp_gmm_score(self._h, ptr, frame_seg._h, p(ll), p(sc), p(am), where, int(precision),
                                                C.byref(ms) if timing else None))
        if loglik:
            res["loglik"] = ll
        if scores:
            res["scores"] = sc
        if argmax:
            res["argmax"] = am
        if timing:
            res["kernel_ms"] = ms.value
        return res

    @property
    def last_rescored(self) -> int:
        n = C.c_int32(0)
        _lib.check(self._lib.ssp_gmm_last_rescored(self._h, C.byref(n)))
        return n.value

    @property
    def last_auto(self) -> dict:
        """what the last precision = "auto" call chose and what its pilot saw (precision_used -1: no such call yet)"""
        a, b, c, f = C.c_int32(-1), C.c_int32(0), C.c_int32(0), C.c_float(0.0)
        _lib.check(self._lib.ssp_gmm_last_auto(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(f)))
        return {"precision_used": a.value, "pilot_utterances": b.value, "pilot_listed": c.value, "predicted_cost_of_precision_1": f.value}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ssp_gmm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def centroids(ctx: Context, X, labels, num: int):
    """avg[s] = mean(X[labels == s]) (float64 accumulator, row order) — d_vector.py:310-313."""
    keep, ptr, where = _as_f32(X, "X")
    if keep.ndim != 2:
        raise ValueError("X must be (N, d)")
    if where == _lib.DEVICE:
        import torch
        lab = labels.to(torch.int32).contiguous()
        lptr = lab.data_ptr()
    else:
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        lptr = lab.ctypes.data
    if int(lab.shape[0]) != int(keep.shape[0]):
        raise ValueError("one label per row of X")
    out = ctx._empty((int(num), int(keep.shape[1])), where)
    optr = out.data_ptr() if where == _lib.DEVICE else out.ctypes.data
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_centroids(ctx._h, ptr, lptr, int(keep.shape[0]), int(keep.shape[1]), int(num), optr, where, None))
    return out


def dtw_distances(ctx: Context, queries, templates, normalize: bool = False, timing: bool = False):
    """All-pairs DTW distances (ssp_dtw_distances).  queries / templates: lists of arrays, each (L,) (the reference's
    flattened MFCCs) or (L, dim).  Returns the (n_q, n_t) float32 matrix (numpy)."""
    def pack(seqs):
        arrs = [np.asarray(s, dtype=np.float32) for s in seqs]
        arrs = [a.reshape(-1, 1) if a.ndim == 1 else a for a in arrs]
        dims = {a.shape[1] for a in arrs if a.shape[0] > 0}
        if len(dims) > 1:
            raise ValueError("all sequences must share the feature dimension")
        dim = dims.pop() if dims else 1
        flat = np.concatenate([a.reshape(-1, dim) for a in arrs]) if arrs else np.zeros((0, dim), np.float32)
        return np.ascontiguousarray(flat), Segments.from_lengths(ctx, [a.shape[0] for a in arrs]), dim
    q, qs, dq = pack(queries)
    t, ts, dt = pack(templates)
    if dq != dt:
        raise ValueError("queries and templates must share the feature dimension")
    out = np.empty((qs.n, ts.n), dtype=np.float32)
    ms = C.c_float(0.0)
    _lib.check(ctx._lib.ssp_dtw_distances(ctx._h, q.ctypes.data, qs._h, t.ctypes.data, ts._h, dq, 1 if normalize else 0,
                                           out.ctypes.data, _lib.HOST, C.byref(ms) if timing else None))
    return (out, ms.value) if timing else out


def fastdtw_distances(ctx: Context, queries, templates, radius: int = 1, timing: bool = False):
    """All-pairs FastDTW distances (ssp_fastdtw_distances): the reference's dtw_method = 2 (MFCC_DTW.py:69-70) on 1-D sequences,
    float64.  Returns the (n_q, n_t) float64 matrix."""
    def pack(seqs):
        arrs = [np.ascontiguousarray(np.asarray(s, dtype=np.float32).reshape(-1)) for s in seqs]
        flat = np.concatenate(arrs) if arrs else np.zeros(0, np.float32)
        return flat, Segments.from_lengths(ctx, [a.shape[0] for a in arrs])
    fq, sq = pack(queries)
    ft, st = pack(templates)
    out = np.empty((sq.n, st.n), dtype=np.float64)
    ms = C.c_float(0.0)
    _lib.check(ctx._lib.ssp_fastdtw_distances(ctx._h, fq.ctypes.data, sq._h, ft.ctypes.data, st._h, int(radius), out.ctypes.data,
                                               C.byref(ms) if timing else None))
    return (out, ms.value) if timing else out


def dtw_path(ctx: Context, x, y):
    """d and the warping path of dtw.accelerated_dtw(x, y, 'euclidean') (ssp_dtw_path, float64 on the GPU).
    x (r,) or (r, dim), y (c,) or (c, dim).  Returns (d, path_i int64[len], path_j int64[len])."""
    a = np.asarray(x, dtype=np.float32)
    b = np.asarray(y, dtype=np.float32)
    a = np.ascontiguousarray(a.reshape(-1, 1) if a.ndim == 1 else a)
    b = np.ascontiguousarray(b.reshape(-1, 1) if b.ndim == 1 else b)
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1] or a.shape[0] < 1 or b.shape[0] < 1:
        raise ValueError("x (r, dim) and y (c, dim) must be non-empty and share dim")
    r, c = a.shape[0], b.shape[0]
    pi = np.empty(r + c, dtype=np.int32)
    pj = np.empty(r + c, dtype=np.int32)
    d = C.c_double(0.0)
    n = C.c_int32(0)
    _lib.check(ctx._lib.ssp_dtw_path(ctx._h, a.ctypes.data, r, b.ctypes.data, c, a.shape[1], C.byref(d), pi.ctypes.data,
                                      pj.ctypes.data, C.byref(n)))
    return d.value, pi[:n.value].astype(np.int64), pj[:n.value].astype(np.int64)


def dense_forward(ctx: Context, X, Wt, bias=None, relu: bool = False, timing: bool = False):
    """Y = act(X @ Wt.T + bias) — one Keras Dense layer (ssp_dense_forward).  X (N, d_in); Wt (units, d_in) is the Keras
    kernel transposed; all arrays numpy (host) or all torch CUDA tensors.  Returns Y (N, units) of the same kind."""
    xk, xp, where = _as_f32(X, "X")
    wk, wp, wwhere = _as_f32(Wt, "Wt")
    if wwhere != where:
        raise ValueError("X and Wt must both be numpy arrays or both be torch CUDA tensors")
    if xk.ndim != 2 or wk.ndim != 2 or xk.shape[1] != wk.shape[1]:
        raise ValueError("X (N,d_in) and Wt (units,d_in) must share d_in")
    N, d_in, units = int(xk.shape[0]), int(xk.shape[1]), int(wk.shape[0])
    bk, bp = None, None
    if bias is not None:
        bk, bp, bwhere = _as_f32(bias, "bias")
        if bwhere != where or int(np.prod(bk.shape)) != units:
            raise ValueError("bias must have `units` entries and live where X lives")
    Y = ctx._empty((N, units), where)
    yp = Y.data_ptr() if where == _lib.DEVICE else Y.ctypes.data
    ms = C.c_float(0.0)
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_dense_forward(ctx._h, xp, N, d_in, wp, bp, units, 1 if relu else 0, yp, where,
                                               C.byref(ms) if timing else None))
    return (Y, ms.value) if timing else Y


class DnnForward:
    """A fully connected network packed once for the GPU forward pass (ssp_dnn): ``layers`` = list of (Wt (units, d_in) float32 — the
    Keras kernel transposed —, bias (units,) or None, relu bool).  ``forward(X)`` = predict: the layers whose widths are <= 256 run in
    one kernel with the activations kept in registers (the d-vector network's hidden and output layers), wider layers in front of
    them one GEMM launch each."""

    def __init__(self, ctx: Context, layers):
        self.ctx = ctx
        self._lib = ctx._lib
        n = len(layers)
        if n < 1:
            raise ValueError("at least one layer")
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w, _, _ in layers]
        bs = [None if b is None else np.ascontiguousarray(b, dtype=np.float32).reshape(-1) for _, b, _ in layers]
        dims = [int(ws[0].shape[1])] + [int(w.shape[0]) for w in ws]
        for i, w in enumerate(ws):
            if w.ndim != 2 or w.shape[1] != dims[i] or (bs[i] is not None and bs[i].shape[0] != dims[i + 1]):
                raise ValueError("layer %d: kernel (units, d_in) / bias (units,) do not chain" % i)
        self.dims = dims
        c_dims = (C.c_int32 * (n + 1))(*dims)
        c_w = (C.c_void_p * n)(*[w.ctypes.data for w in ws])
        c_b = (C.c_void_p * n)(*[None if b is None else b.ctypes.data for b in bs])
        c_r = (C.c_int32 * n)(*[1 if r else 0 for _, _, r in layers])
        h = C.c_void_p()
        _lib.check(self._lib.ssp_dnn_create(ctx._h, n, c_dims, c_w, c_b, c_r, C.byref(h)))
        self._h = h

    def forward(self, X, timing: bool = False):
        xk, xp, where = _as_f32(X, "X")
        if xk.ndim != 2 or int(xk.shape[1]) != self.dims[0]:
            raise ValueError("X must be (N, %d)" % self.dims[0])
        N = int(xk.shape[0])
        Y = self.ctx._empty((N, self.dims[-1]), where)
        yp = Y.data_ptr() if where == _lib.DEVICE else Y.ctypes.data
        ms = C.c_float(0.0)
        with self.ctx._ordered(where):
            _lib.check(self._lib.ssp_dnn_forward(self._h, xp, N, yp, where, C.byref(ms) if timing else None))
        return (Y, ms.value) if timing else Y

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ssp_dnn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cosine_identify(ctx: Context, X, Cn, dist: bool = False, argmin: bool = True, minval: bool = True,
                    timing: bool = False, precision: int = 0, counts: bool = True) -> dict:
    """dist[i,j] = clip(1 - cos(X[i], C[j]), 0, 2); argmin over j (first index on ties) — d_vector.py:315-319.
    precision 0: fp32 MFMA (parity path).  precision 1: bf16x3 MFMA sweep + fp32 re-scoring of the rows whose two best cosines are
    closer than a proven error bound — the arg-min of the fp32 path on every row, about 3x faster; no distance matrix (dist=False);
    the result then carries "rescored" (rows that went through fp32 again).  precision 2: a bf16 sweep in front (bound 4e-3), its close
    calls to the bf16x3 sweep: same arg-min guarantee, the minimum only within 4e-3 on rows the first sweep decided.
    precision 3 or "auto": the fp32 path's arg-min at the cost of the cheapest of 2, 1 and 0 — a pilot on the first ~2 % of the rows
    reads how many each sweep would hand on (one host wait); the result carries "auto" (what it chose and saw).
    ``counts=False`` skips the "rescored" / "split_rows" diagnostics: with CUDA tensors the call then returns without waiting for the GPU."""
    if precision == "auto":
        precision = 3
    xk, xp, where = _as_f32(X, "X")
    ck, cp, cwhere = _as_f32(Cn, "C")
    if cwhere != where:
        raise ValueError("X and C must both be numpy arrays or both be torch CUDA tensors")
    if xk.ndim != 2 or ck.ndim != 2 or xk.shape[1] != ck.shape[1]:
        raise ValueError("X (N,d) and C (S,d) must share d")
    N, d = int(xk.shape[0]), int(xk.shape[1])
    S = int(ck.shape[0])
    dm = ctx._empty((N, S), where) if dist else None
    am = ctx._empty((N,), where, "int32") if argmin else None
    mv = ctx._empty((N,), where) if minval else None

    def p(x):
        if x is None:
            return None
        return x.data_ptr() if where == _lib.DEVICE else x.ctypes.data
    ms = C.c_float(0.0)
    with ctx._ordered(where):
        _lib.check(ctx._lib.ssp_cosine_identify2(ctx._h, xp, N, d, cp, S, p(dm), p(am), p(mv), where, int(precision),
                                                 C.byref(ms) if timing else None))
    res = {}
    if precision == 3:
        a, b, c, e = C.c_int32(-1), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _lib.check(ctx._lib.ssp_cosine_last_auto(ctx._h, C.byref(a), C.byref(b), C.byref(c), C.byref(e)))
        res["auto"] = {"precision_used": a.value, "pilot_rows": b.value, "pilot_to_bf16x3": c.value, "pilot_to_fp32": e.value}
    if precision >= 1 and counts:
        n = C.c_int32(0)
        _lib.check(ctx._lib.ssp_cosine_last_rescored(ctx._h, C.byref(n)))
        res["rescored"] = n.value
        _lib.check(ctx._lib.ssp_cosine_last_split_rows(ctx._h, C.byref(n)))
        res["split_rows"] = n.value   # precision 2: rows the bf16 sweep handed to the bf16x3 sweep
    if dist:
        res["dist"] = dm
    if argmin:
        res["argmin"] = am
    if minval:
        res["min"] = mv
    if timing:
        res["kernel_ms"] = ms.value
    return res
