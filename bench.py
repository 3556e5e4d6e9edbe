#!/usr/bin/env python
"""Benchmark of the MFCC -> GMM-UBM / d-vector scoring hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N ranks itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline (`value`): MFCC frames/s on BASELINE.json configs[1] — 100k x 3 s synthetic 16 kHz utterances per GPU,
sidekit-dialect MFCC + delta + delta-delta = 39-d, inputs resident in HBM, one fused kernel launch per step.
A "step" = one pass of the fused MFCC kernel over the rank's whole batch.  Weak scaling: every rank owns its own
100k utterances (utterances shard with no data-path collective); the only collective on the path is the gather of
per-utterance speaker decisions after GMM scoring (stage "gmm").
Extra stages reported in the same JSON line (not part of `value`): GMM-UBM scoring on configs[2] and cosine
scoring on configs[4].
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
VALU_FP32_PEAK_TF = 157.3     # packed fp32 FMA on the vector ALU: 256 CUs x 4 SIMDs x 16 lanes x 2 (packed) x 2 flop x 2.4 GHz (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md; never the 2:1-sparsity figure)
MFMA_F32_PEAK_TF = 157.3     # v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md, Matrix cores)


def synth_audio_device(torch, n_utt, n_samp, fs, seed, device, out=None, chunk=2000, n_spk=50):
    """SURVEY.md 8(d) throughput recipe on the device: 5 harmonics of f0 = 90 + 3*spk Hz, 3 Hz AM, N(0, 0.05) noise."""
    if out is None:
        out = torch.empty((n_utt, n_samp), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    t = torch.arange(n_samp, device=device, dtype=torch.float32) / fs
    am = 0.6 + 0.4 * torch.sin(2 * np.pi * 3 * t)
    for lo in range(0, n_utt, chunk):
        hi = min(lo + chunk, n_utt)
        spk = (torch.arange(lo, hi, device=device) % n_spk).to(torch.float32)
        f0 = (90.0 + 3.0 * spk)[:, None]
        x = torch.zeros((hi - lo, n_samp), dtype=torch.float32, device=device)
        for h in range(1, 6):
            x += torch.sin((2 * np.pi * h) * f0 * t[None, :]) / h
        x *= 0.3 * am[None, :]
        x += 0.05 * torch.randn((hi - lo, n_samp), generator=g, device=device, dtype=torch.float32)
        out[lo:hi] = x.clamp_(-1, 1)
    return out


def cpu_baseline_mfcc(n_utt, n_samp, fs, budget_s=15.0):
    """The oracle (numpy float64 restatement of the reference path) timed on this host, one thread."""
    from oracle import ref_cpu as O
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    cfg, w, fb, dct = O.sidekit_tables(delta_order=2)
    rng = np.random.default_rng(0)
    x = np.clip(0.3 * np.sin(2 * np.pi * 120 * np.arange(n_samp) / fs) + 0.05 * rng.standard_normal(n_samp), -1, 1).astype(np.float32)

    def run():
        frames, utts, t0 = 0, 0, time.perf_counter()
        while utts < n_utt and time.perf_counter() - t0 < budget_s:
            frames += O.mfcc_pipeline(x, cfg, w, fb, dct).shape[0]
            utts += 1
        return frames, utts, time.perf_counter() - t0
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            frames, utts, dt = run()
    else:
        frames, utts, dt = run()
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d utterances x %d samples, oracle.ref_cpu.mfcc_pipeline (numpy float64 restatement of the "
                      "sidekit-dialect MFCC + delta + delta-delta path), 1 thread, %.1f s" % (utts, n_samp, dt),
            "host_cores": os.cpu_count(), "blas_threads": 1 if threadpool_limits is not None else None}


def cpu_baseline_mfcc_loop(n_samp, fs, budget_s=6.0):
    """the reference's own per-frame Python loop (utils/processing.py:129-143, restated as oracle.MFCC_loop) on the
    in-repo dialect, one thread."""
    from oracle import ref_cpu as O
    rng = np.random.default_rng(3)
    x = np.clip(0.3 * np.sin(2 * np.pi * 120 * np.arange(n_samp) / fs) + 0.05 * rng.standard_normal(n_samp), -1, 1)
    frames, utts, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        frames += O.MFCC_loop(x, fs=fs, frameSize=512, step=256).shape[0]
        utts += 1
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d utterances x %d samples, oracle.ref_cpu.MFCC_loop = the per-frame Python loop of utils/processing.py:129-143 "
                      "(in-repo dialect, %d Hz, 512/256, 13-d), 1 thread, %.1f s" % (utts, n_samp, fs, dt)}


def cpu_baseline_gmm_loop(D, K, S, budget_s=8.0):
    """the reference's scoring loop itself, GMM_UBM.py:190-193 — `GMM[i].score(x_j) - UBM.score(x_j)` per (speaker,
    utterance) with sklearn GaussianMixture objects (the library the reference calls), the UBM rescored once per speaker."""
    from sklearn.mixture import GaussianMixture
    rng = np.random.default_rng(4)

    def make(mu):
        g = GaussianMixture(n_components=K, covariance_type="diag")
        g.weights_, g.means_, g.covariances_ = w, mu, cov
        g.precisions_cholesky_ = 1.0 / np.sqrt(cov)
        return g
    w = rng.dirichlet(5 * np.ones(K))
    mu0 = rng.standard_normal((K, D))
    cov = rng.uniform(0.5, 2, (K, D))
    UBM = make(mu0)
    GMM = [make(mu0 + 0.3 * rng.standard_normal((K, D))) for _ in range(S)]
    X = [rng.standard_normal((298, D)) for _ in range(4)]
    n_utt, t0 = 0, time.perf_counter()
    pred = np.zeros((1, S))
    while time.perf_counter() - t0 < budget_s:
        x = X[n_utt % 4]
        for i in range(S):
            pred[0, i] = GMM[i].score(x) - UBM.score(x)
        n_utt += 1
    dt = time.perf_counter() - t0
    return {"value": n_utt * 298 * (S + 1) / dt, "unit": "frame-scores/s", "cores": 1, "kind": "reference",
            "sample": "%d utterances of 298 x %d frames through the loop of GMM_UBM.py:190-193 with sklearn GaussianMixture.score "
                      "(K=%d, %d speakers, the UBM rescored per speaker; counted as %d useful model scores per frame), "
                      "single process, BLAS threads = host default, %.1f s" % (n_utt, D, K, S, S + 1, dt)}


def cpu_baseline_gmm(D, K, n_models, budget_s=8.0):
    from oracle import ref_cpu as O
    rng = np.random.default_rng(1)
    w = rng.dirichlet(5 * np.ones(K))
    mu = rng.standard_normal((K, D))
    cov = rng.uniform(0.5, 2, (K, D))
    X = rng.standard_normal((298, D))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        O.gmm_score(w, mu, cov, X)  # one (speaker, utterance) score call, as the loop at GMM_UBM.py:183-185
        n += 298
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frame-scores/s", "cores": 1, "kind": "port",
            "sample": "per-(model, utterance) score() calls on 298x%d frames, K=%d, numpy float64, one process (BLAS threads = host "
                      "default), %.1f s" % (D, K, dt), "host_cores": os.cpu_count()}


def cpu_baseline_cosine(d, S, budget_s=5.0):
    from scipy.spatial.distance import cosine
    rng = np.random.default_rng(2)
    Cn = rng.standard_normal((S, d))
    x = rng.standard_normal(d).astype(np.float32)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for j in range(64):
            cosine(x, Cn[j])  # per-pair call as in d_vector.py:315-318
        n += 64
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "pair-scores/s", "cores": 1, "kind": "reference",
            "sample": "scipy.spatial.distance.cosine per pair (the reference's own call, d_vector.py:317), d=%d, %.1f s" % (d, dt)}


def _timed_loop(fn, budget_s):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        n += fn()
    return n, time.perf_counter() - t0


def cpu_baseline_em(D, K, budget_s=3.0):
    """one EM iteration's sufficient statistics (E step + M sums) on the oracle's float64 restatement of sklearn's diag EM"""
    from oracle import ref_cpu as O
    rng = np.random.default_rng(21)
    w, mu, cov = rng.dirichlet(5 * np.ones(K)), rng.standard_normal((K, D)), rng.uniform(0.5, 2, (K, D))
    X = rng.standard_normal((20000, D))
    n, dt = _timed_loop(lambda: (O.gmm_em_stats(w, mu, cov, X), len(X))[1], budget_s)
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "oracle.ref_cpu.gmm_em_stats (numpy float64 restatement of sklearn's diag-covariance E step + M sums, "
                      "sk:mixture/_gaussian_mixture.py), 20000 x %d frames per call, K=%d, one process (BLAS threads = host default), %.1f s" % (D, K, dt)}


def cpu_baseline_dnn(dims, budget_s=3.0):
    """the d-vector network forward (d_vector.py:171-189: Dense(256) x 4) as numpy float32 matmuls — what Keras' predict computes"""
    from oracle import ref_cpu as O
    rng = np.random.default_rng(22)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / dims[i] ** 0.5).astype(np.float32), None, "relu" if i < len(dims) - 2 else None)
              for i in range(len(dims) - 1)]
    X = rng.standard_normal((4096, dims[0])).astype(np.float32)
    n, dt = _timed_loop(lambda: (O.dense_net_forward(X, layers), len(X))[1], budget_s)
    return {"value": n / dt, "unit": "embeddings/s", "cores": 1, "kind": "port",
            "sample": "oracle.ref_cpu.dense_net_forward (numpy matmul + bias + ReLU per layer, %s), batches of 4096, one process "
                      "(BLAS threads = host default), %.1f s" % ("-".join(str(d) for d in dims), dt)}


def cpu_baseline_dtw(L, budget_s=4.0):
    """one DTW distance per call on 1222-element flattened MFCC sequences (MFCC_DTW.py:57-108: accelerated_dtw on (-1, 1) sequences)"""
    from oracle import ref_cpu as O
    rng = np.random.default_rng(23)
    a, b = rng.standard_normal(L), rng.standard_normal(L)
    n, dt = _timed_loop(lambda: (O.dtw_distance(a, b), 1)[1], budget_s)
    return {"value": n / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "oracle.ref_cpu.dtw_distance (numpy anti-diagonal sweep of the dtw package's recurrence; the package itself is absent), "
                      "%d x %d cells per pair, 1 thread, %.1f s" % (L, L, dt)}


def cpu_baseline_plp(n_samp, fs, budget_s=4.0):
    from oracle import ref_cpu as O
    rng = np.random.default_rng(24)
    x = np.clip(0.3 * np.sin(2 * np.pi * 120 * np.arange(n_samp) / fs) + 0.05 * rng.standard_normal(n_samp), -1, 1).astype(np.float32)
    n, dt = _timed_loop(lambda: O.sidekit_plp(x, fs)[0].shape[0], budget_s)
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "oracle.ref_cpu.sidekit_plp (numpy float64 restatement of sidekit's plp: Bark front end + RASTA + Levinson + cepstrum) on "
                      "%d-sample utterances, one process, %.1f s" % (n_samp, dt)}


def kernel_source_sha256(names=("mfcc_stream_kernel.hpp", "mfcc_stream.hip", "cplx.hpp", "mfcc.hpp", "common.hpp")):
    """identity of the headline kernel's source: the PMC files under profiles/ record the value they were taken on, and a counter
    reading is only quoted for the binary built from the same source"""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "speech_signal_processing_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


GMM_SOURCES = ("gmm.hip", "common.hpp")         # kernel source the MFMA-pipe counters of profiles/gmm_mfma_util.json are tied to
COSINE_SOURCES = ("cosine.hip", "common.hpp")    # ... profiles/cosine_mfma_util.json


def mfma_busy_from_profile(fname, kernel, sources):
    """profiles/<fname> (tools/pmc_mfma.sh + store_mfma_pmc.py): the MFMA-pipe busy fraction of `kernel`, quoted only when the counters
    were taken on the source this build was made from (sha256 of `sources`); -> (value or None, provenance dict)"""
    src = {"file": "profiles/" + fname, "matches_this_build": False}
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", fname)))
        src["taken_on"] = j.get("kernel_source_sha256")
        src["matches_this_build"] = j.get("kernel_source_sha256") == kernel_source_sha256(sources)
        src["workload"] = j.get("workload")
        if src["matches_this_build"]:
            return (j["kernels"][kernel]["derived"] or {}).get("mfma_busy_fraction"), src
    except Exception as e:
        src["error"] = repr(e)
    return None, src


# ---------------------------------------------------------------------------------------------- the line the driver parses
LINE_LIMIT = 4096            # the LAST stdout line stays under this; everything else goes to the detail file (round 5's 22.8 KB line did not parse)
LINE_TARGET = 3600           # extras are dropped from the back until the line is under this (a margin below the limit)
LIMITER_MFCC = "valu-issue at the power-capped clock (VALU busy 78 %, HBM traffic 1.01 x algorithmic); frac stays priced on HBM"


def _sig(x, n=6):
    """floats to n significant digits (bytes of the line); non-finite floats become null: the line is strict JSON"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (n, x))
    if isinstance(x, (int, np.integer)):
        return int(x)
    return x


def clean_json(o, n=None):
    """a JSON-safe copy: numpy scalars to Python's, NaN / inf to null (json.dumps(allow_nan=False) passes), floats rounded if n is given"""
    if isinstance(o, dict):
        return {str(k): clean_json(v, n) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [clean_json(v, n) for v in o]
    if isinstance(o, np.ndarray):
        return clean_json(o.tolist(), n)
    if isinstance(o, (float, np.floating)):
        return _sig(o, n if n else 17)
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    return o


def _pick(d, *keys):
    return {k: _sig(d[k]) for k in keys if isinstance(d, dict) and k in d}


def _stage_summary(s):
    """one small object per stage: value / unit / kernel_ms / frac (+ the stage's own few scalars); the full stage is in the detail file"""
    if not isinstance(s, dict):
        return None
    rf = s.get("roofline") or s.get("front_roofline") or {}
    out = {"value": _sig(s.get("value"), 5)}          # (units, bounds and kernel names: the detail file)
    ms = rf.get("kernel_ms", s.get("kernel_ms"))
    if ms is not None:
        out["kernel_ms"] = _sig(ms, 5)
    if "frac" in rf:
        out["frac"] = _sig(rf["frac"], 4)
    if "mfma_busy" in rf:
        out["mfma_busy"] = _sig(rf["mfma_busy"], 4)
    return out


def compact_line(res, detail_path=None):
    """The ONE line the driver parses (<= LINE_LIMIT bytes, strict JSON): the contract's keys, `roofline`, `cpu_baseline`, `value_normalised`,
    `build`, and one small summary per other stage.  `res` is the full result (what bench_detail.json holds)."""
    line = _pick(res, "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {"metric": "MFCC frames/s (one fused pass: framing .. DCT + delta + delta-delta, 39-d)", **line}
    c = res.get("config", {})
    line["config"] = _pick(c, "utterances_per_gpu", "frames_per_gpu", "d_out", "parallelism", "world_size_observed", "backend")
    line["config"] = {"workload": str(c.get("workload", ""))[:120], **line["config"]}
    if "kernel_ms_per_rank" in c:
        line["config"]["kernel_ms_per_rank"] = [_sig(v, 5) for v in c["kernel_ms_per_rank"]]
    r = res.get("roofline", {})
    line["roofline"] = _pick(r, "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes_per_launch", "bytes_per_frame")
    line["roofline"]["kernel"] = str(r.get("kernel", ""))[:100]
    if "limiter" in r:
        line["roofline"]["limiter"] = r["limiter"]
    f = res.get("roofline_flop") or {}
    if "frac" in f:
        line["roofline"]["valu_fp32_frac"] = _sig(f["frac"], 4)
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind", "host_cores", "blas_threads", "skipped")
        if "sample" in cb:
            line["cpu_baseline"]["sample"] = str(cb["sample"])[:110]
    cp = res.get("cpu_baseline_parallel")
    if isinstance(cp, dict):
        line["cpu_baseline_parallel"] = _pick(cp, "value", "unit", "cores", "kind", "host_cores", "usable_cores", "blas_threads_per_worker", "value_16_workers", "error")
    if "value_normalised" in res:
        line["value_normalised"] = _pick(res["value_normalised"], "value", "ms_per_step")
    env = res.get("env") or {}
    su = env.get("sustained_mfcc") or {}
    if su:
        line["env"] = {"sclk_mhz": _sig((su.get("sclk_mhz") or {}).get("mean"), 4), "power_w": _sig((su.get("power_w") or {}).get("mean"), 4)}
    if isinstance(res.get("build"), dict):
        line["build"] = _pick(res["build"], "build_mode", "lib_bytes")
    # ---- the other stages: a summary each (never part of `value`)
    for k in ("mfcc_ref26_cmvn", "mfcc_librosa", "mfcc_host_fed", "gmm", "gmm_bf16x3", "gmm_bf16x3_proven_band", "gmm_auto", "gmm_host_fed", "cosine",
              "cosine_bf16x3", "cosine_bf16_cascade", "cosine_auto", "cosine_host_fed", "gmm_em", "dvector_dnn", "dvector_pipeline", "dtw", "plp"):
        if k in res:
            line[k] = _stage_summary(res[k])
    if "gmm" in res:   # the one collective of the path: what the N-rank runs are checked on
        line["gmm"].update(_pick(res["gmm"], "gathered_rows", "record_bytes", "ms_per_step"))
    if "mfcc_host_fed" in res:
        line["mfcc_host_fed"].update(_pick(res["mfcc_host_fed"], "h2d_gbs", "frac_of_pcie_bound", "value_i16", "frac_of_pcie_bound_i16"))
    for k in ("gmm_auto", "cosine_auto"):
        if k in res:
            line[k].update(_pick(res[k], "worst_ratio_to_best_fixed", "mismatches_vs_fp32"))
    for k in ("gmm_host_fed", "cosine_host_fed"):
        if k in res:
            line[k] = _pick(res[k], "value", "wall_ms", "sum_ms", "overlap", "error")
    if "mfcc_inrepo" in res:
        line["mfcc_inrepo"] = {t: _stage_summary(v) for t, v in res["mfcc_inrepo"].items() if t in ("16k", "8k")}
    c3 = res.get("gmm_cfg3_shape")
    if isinstance(c3, dict):
        line["gmm_cfg3_shape"] = {t: _pick(c3[t], "value", "kernel_ms") for t in ("f32", "bf16x3") if t in c3}
        if "bf16x3_full_share" in c3:
            line["gmm_cfg3_shape"]["full_share"] = _pick(c3["bf16x3_full_share"], "value", "measured_s", "utterances_per_gpu")
    line["detail"] = detail_path
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_TARGET:   # never lose the contract's keys to the extras: drop summaries from the back until it fits
        for k in [k for k in list(line)[::-1] if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                             "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline",
                                                             "value_normalised", "build", "detail")]:
            del line[k]
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) < LINE_TARGET:
                break
    return text


def write_detail(res, path):
    """the full result (every stage's dict, env windows, close-call tables, prose) as strict JSON; returns the path written (or None)"""
    for p in (path, os.path.join("/tmp", "bench_detail.json")):
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as f:
                json.dump(clean_json(res), f, allow_nan=False)
            return p
        except OSError:
            continue
    return None


# ---------------------------------------------------------------------------------------------- box state (`env` in the detail file)
CLOSE_GMM_OFFSETS = (1.0, 0.1, 0.01)   # speaker-mean offsets (x std) of the close-call rows; 0.3 = the headline rows
CLOSE_COS_NOISE = (3.0, 4.0, 10.0)       # embedding noise of the close-call rows; 0.7 = the headline rows
NOMINAL_SCLK_MHZ = 2000       # value_normalised: the headline at this engine clock (what the boxes of the pool grant on the power cap: 1.85 - 2.05 GHz)
HBM_COPY_REF_GBS = 6290.0    # MI355X_MICROARCH.md: measured float4 copy
SYSFS_FILES = {"sclk_hz": "freq1_input", "mclk_hz": "freq2_input", "power_uw": "power1_input", "junction_mC": "temp2_input"}


def env_sampler_main():
    """Child process of bench.py (started BEFORE the parent touches the GPU; this process never does): reads the card's hwmon files
    in sysfs every ~5 ms and, when told to quit, prints every sample as JSON.  stdin lines: `card <pci bus address>` picks
    the card (until then nothing is read), `quit` ends."""
    import glob
    import select
    files, samples, card, stride, tick = {}, [], None, 1, 0
    while True:
        # ~5 ms between reads: a tight loop would hold a core and query the SMU continuously while the headline is measured
        r, _, _ = select.select([sys.stdin], [], [], 0.005 if files else 0.05)
        if r:
            line = sys.stdin.readline()
            if not line or line.strip() == "quit":
                break
            if line.startswith("card "):
                card = line.split()[1].lower()
                for c in glob.glob("/sys/class/drm/card*/device"):
                    if os.path.basename(os.path.realpath(c)).lower() == card:
                        for h in glob.glob(os.path.join(c, "hwmon", "hwmon*")):
                            for k, f in SYSFS_FILES.items():
                                if os.access(os.path.join(h, f), os.R_OK):
                                    files[k] = os.path.join(h, f)
        if files:
            row = {"t": time.time()}
            for k, f in files.items():
                try:
                    with open(f) as fh:
                        row[k] = int(fh.read().strip())
                except (OSError, ValueError):
                    pass
            tick += 1
            if tick % stride == 0:
                samples.append(row)
            if len(samples) >= 20000:   # bounded: keep every second sample and halve the rate from here on
                samples, stride = samples[::2], stride * 2
    print(json.dumps({"card": card, "files": sorted(files), "samples": samples}))


def env_sampler_start():
    import subprocess
    # (under rocprofv3 the profiler's preloaded library would come along and initialise the GPU in the child: it gets a clean environment)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
    try:
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--env-sampler"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                stderr=subprocess.DEVNULL, text=True, env=env)
    except OSError:
        return None


def env_summary(proc, windows):
    """ends the sampler and reduces its samples over the named time windows {name: (t0, t1)} (time.time() of this host)"""
    if proc is None:
        return {"error": "sampler did not start"}
    try:
        out, _ = proc.communicate("quit\n", timeout=20)
        doc = json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # (never lose the bench line to the bookkeeping)
        return {"error": repr(e)}
    res = {"card": doc.get("card"), "source": "sysfs hwmon (%s), sampled by a child process started before the GPU was touched" % ", ".join(doc.get("files", [])),
           "n_samples_total": len(doc["samples"])}
    for name, (t0, t1) in windows.items():
        rows = [r for r in doc["samples"] if t0 <= r["t"] <= t1]
        w = {"n_samples": len(rows), "seconds": t1 - t0}
        for k, scale, unit in (("sclk_hz", 1e-6, "sclk_mhz"), ("mclk_hz", 1e-6, "mclk_mhz"), ("power_uw", 1e-6, "power_w"), ("junction_mC", 1e-3, "junction_c")):
            v = [r[k] * scale for r in rows if k in r]
            if v:
                w[unit] = {"mean": float(np.mean(v)), "min": float(np.min(v)), "max": float(np.max(v))}
        res[name] = w
    return res


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` outside a torch.distributed launch: this process never touches the GPU (no HIP call, no
    torch.cuda) — it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a CHILD process (one rank per
    GPU, RCCL), passes its output through and exits with its code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SSP_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=100000, help="utterances per GPU (configs[1]: 100k)")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--variant", type=int, default=0, help="0 auto | 1 generic kernel | 2 fused workgroup kernel | 3 fused wave-stream kernel (| 4: the 2048-point one, librosa stage only)")
    ap.add_argument("--ref26-variant", type=int, default=0, help="kernel variant of the 26-d + CMVN stage (as --variant)")
    ap.add_argument("--inrepo-variant", type=int, default=0, help="kernel variant of the in-repo MFCC stage (as --variant)")
    ap.add_argument("--stages", default="mfcc,ref26,hostfed,inrepo,librosa,gmm,gmm4,cosine,closecalls,em,dnn,dvec,dtw,plp")
    ap.add_argument("--hostfed-utts", type=int, default=25000, help="utterances of the host-fed stage's pinned batch (25000 x 3 s = 4.8 GB in, 1.16 GB out)")
    ap.add_argument("--gmm4-utts", type=int, default=12000, help="utterances per GPU of the configs[3]-shaped sample (full: 150000 per GPU; SURVEY.md 8(d) asks for >= 12000)")
    ap.add_argument("--no-gmm4-full", dest="gmm4_full", action="store_false", help="skip the measured full per-GPU share of configs[3] (150000 utterances, bf16x3 path, ~12 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gmm-precision", type=int, default=0, help="0 exact-fp32 MFMA (parity path) | 1 bf16x3 split MFMA")
    ap.add_argument("--detail", default=os.environ.get("SSP_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json")),
                    help="where the full result goes (every stage's dict, env windows, close-call tables); the last stdout line is the compact summary")
    ap.add_argument("--full-line", action="store_true", help="print the FULL result as the last stdout line (tools/ab*.sh, stage_ms.sh read stage internals from it); the driver's run never sets this")
    ap.add_argument("--env-sampler", action="store_true", help="(internal) run as the sysfs sampler child")
    ap.add_argument("--no-env", action="store_true", help="no sysfs sampler child, no calibration kernels")
    ap.add_argument("--sustain-s", type=float, default=1.5, help="seconds the MFCC kernel is repeated behind the timed region while the sampler reads clocks / power (the timed region itself is ~0.2 s: too short for the governor to show its steady state)")
    args = ap.parse_args()
    if args.env_sampler:
        return env_sampler_main()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    sampler = env_sampler_start() if (int(os.environ.get("RANK", "0")) == 0 and not args.no_env) else None

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    if not os.environ.get("SSP_BENCH_REHEARSE") and torch.cuda.device_count() < max(world, local_rank + 1):
        # one clear line instead of a traceback per rank (exit code 2: the launch does not fit this box)
        if local_rank == 0:
            sys.stderr.write("bench.py: --gpus %d needs %d visible GPUs, this box has %d\n" % (world, world, torch.cuda.device_count()))
        raise SystemExit(2)
    # SSP_BENCH_REHEARSE=1: every rank on device 0 over gloo — rehearses the multi-rank control flow on a one-GPU box (numbers are
    # meaningless there); the driver's runs use one GPU per rank over RCCL
    rehearse = bool(os.environ.get("SSP_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import speech_signal_processing_amd as pkg
    from speech_signal_processing_amd import api
    from speech_signal_processing_amd.dist import all_gather_rows, decision_records, max_over_ranks

    def barrier():
        if world > 1:
            dist.barrier()

    stages = set(args.stages.split(","))
    fs = 16000
    n_samp = int(round(args.seconds * fs))
    n_utt = args.utts
    ctx = api.Context.for_torch(local_rank)
    if sampler is not None:
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            sampler.stdin.write("card %04x:%02x:%02x.0\n" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
            sampler.stdin.flush()
        except Exception:
            pass

    # ------------------------------------------------------------------ data: resident in HBM before timing
    audio = synth_audio_device(torch, n_utt, n_samp, fs, seed=1234 + rank, device=device)
    tables = pkg.preset_sidekit(fs=fs, delta_order=2, cmvn=0)
    plan = api.MfccPlan(ctx, tables)
    seg = api.Segments.from_lengths(ctx, np.full(n_utt, n_samp, dtype=np.int64))
    fseg = plan.frame_segments(seg)
    n_frames = fseg.total
    feats = torch.empty((n_frames, plan.d_out), dtype=torch.float32, device=device)
    flat = audio.view(-1)

    # set-up, not a step: the first call builds the plan's chunk table for these segments and first-touches the output pages
    plan.run(flat, seg, fseg, out=feats, variant=args.variant)
    torch.cuda.synchronize()

    # ------------------------------------------------------------------ MFCC: the timed region
    for _ in range(args.warmup):
        plan.run(flat, seg, fseg, out=feats, variant=args.variant)
    kernel_ms = []
    barrier()
    torch.cuda.synchronize()
    wall0 = time.time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, ms = plan.run(flat, seg, fseg, out=feats, variant=args.variant, timing=True)
        kernel_ms.append(ms)
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, device)
    wall1 = time.time()
    # ---- box state, behind the timed region (not part of `value`): the same launch repeated for --sustain-s seconds under the sampler,
    # then two calibration kernels (what this box sustains on a float4 copy and on packed-FMA chains)
    env = None
    if rank == 0 and not args.no_env:
        sustain_ms, ws0 = [], time.time()
        while time.time() - ws0 < args.sustain_s:
            _, ms = plan.run(flat, seg, fseg, out=feats, variant=args.variant, timing=True)
            sustain_ms.append(ms)
        torch.cuda.synchronize()
        ws1 = time.time()
        try:
            cal = ctx.calibrate(20.0)
        except Exception as e:
            cal = {"error": repr(e)}
        wc1 = time.time()
        env = env_summary(sampler, {"timed_region": (wall0, wall1), "sustained_mfcc": (ws0, ws1), "calibration": (ws1, wc1)})
        env["sustained_mfcc_kernel_ms"] = {"median": float(np.median(sustain_ms)), "n": len(sustain_ms), "last_quarter_median": float(np.median(sustain_ms[-max(1, len(sustain_ms) // 4):]))}
        env["calibration_kernels"] = cal
        if "error" not in cal:
            env["calibration_kernels"].update({"copy_ratio": cal["copy_gbs"] / HBM_COPY_REF_GBS, "fma_ratio": cal["fma_tflops"] / VALU_FP32_PEAK_TF,
                                               "reference": "MI355X_MICROARCH.md: 6.29 TB/s measured float4 copy; 157.3 TFLOP/s packed fp32 at 2.4 GHz"})
    total_frames = n_frames * world * args.steps
    value = total_frames / elapsed
    ms_kernel = float(np.median(kernel_ms))
    per_rank_ms = all_gather_rows(torch.tensor([[float(np.median(kernel_ms))]], dtype=torch.float64, device=device)).flatten().tolist()
    bytes_per_frame = tables.cfg.hop * 4 + plan.d_out * 4            # 160*4 read + 39*4 written
    algo_bytes = n_utt * n_samp * 4 + n_frames * plan.d_out * 4      # exact per launch: every sample read once, every feature written once
    achieved = algo_bytes / (ms_kernel * 1e-3) / 1e9
    traffic = None
    ksha = kernel_source_sha256()
    pmc_file = os.path.join(ROOT, "profiles", "mfcc_hbm_traffic.json")
    traffic_source = {"file": "profiles/mfcc_hbm_traffic.json", "kernel_source_sha256_now": ksha, "taken_on": None, "matches_this_build": False}
    if os.path.exists(pmc_file) and n_utt == 100000 and n_samp == 48000:  # the PMC passes were taken on exactly this workload
        try:
            pj = json.load(open(pmc_file))
            traffic_source["taken_on"] = pj.get("kernel_source_sha256")
            traffic_source["matches_this_build"] = pj.get("kernel_source_sha256") == ksha
            if traffic_source["matches_this_build"]:   # a counter reading of another kernel version is not quoted
                traffic = pj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # second bound of the same kernel: fp32 vector arithmetic, from a COUNT of the algorithm's additions and multiplications
    # (tools/flop_count.py — not a counter reading: an instruction census of the kernel's own code would reward executing more)
    flop_roof = None
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("flop_count", os.path.join(ROOT, "tools", "flop_count.py"))
        fc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(fc)
        nz = int(np.count_nonzero(np.asarray(tables.fbank)))
        cnt = fc.count(win=tables.cfg.win_len, n_fft=tables.cfg.n_fft, n_filt=tables.cfg.n_filt, fb_nonzero=nz, n_ceps=tables.cfg.n_ceps,
                       order=tables.cfg.delta_order, delta_N=tables.cfg.delta_N, preemph=tables.cfg.preemph_mode != 0, power=tables.cfg.spec_power)
        tf = n_frames * cnt["flop_per_frame"] / (ms_kernel * 1e-3) / 1e12
        mix = cnt["fma_fraction_of_peak"]
        flop_roof = {"bound": "valu-fp32", "achieved": tf, "peak": VALU_FP32_PEAK_TF, "unit": "TFLOP/s", "frac": tf / VALU_FP32_PEAK_TF,
                "flop_per_frame": cnt["flop_per_frame"], "lane_ops_per_frame": cnt["ops_per_frame"],
                "peak_mix_weighted": VALU_FP32_PEAK_TF * mix, "frac_mix_weighted": tf / (VALU_FP32_PEAK_TF * mix),
                "mix": "an FMA is 2 flop per lane-operation; the algorithm's additions and multiplications fuse into %.3f of that (flop / lane-ops / 2)" % mix,
                "stages": {k: v["flop"] for k, v in cnt["stages"].items()},
                "source": "tools/flop_count.py (counted additions + multiplications of the algorithm; the transform at the smaller of the counted "
                          "radix-4 factorisation and the published real split-radix count)",
                "power": "the pass sits on the 1400 W package cap (profiles/r03_clock_power.md: 1.96 GHz at 1385 W); the peak is quoted at 2.4 GHz"}
    except Exception as e:  # (never lose the bench line to the bookkeeping)
        flop_roof = {"error": repr(e)}
    result = {
        "metric": "MFCC frames/s (fused framing+preemph+window+rFFT+mel+log+DCT+delta+delta-delta, 39-d)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: %d x %.0f s synthetic 16 kHz utterances per GPU, 39-d MFCC (+delta+delta-delta), 400/160/512, 24 mel" % (n_utt, args.seconds),
                   "utterances_per_gpu": n_utt, "frames_per_gpu": n_frames, "d_out": plan.d_out,
                   "kernel_variant": args.variant, "parallelism": "utterance-sharded x%d" % world,
                   "world_size_observed": (dist.get_world_size() if world > 1 else 1),
                   "backend": (dist.get_backend() if world > 1 else None), "kernel_ms_per_rank": per_rank_ms},
        # `bound` names the roofline `peak` / `frac` are priced on (the MFCC pass -> HBM); `limiter` names what actually binds
        "roofline": {"bound": "hbm", "limiter": LIMITER_MFCC if args.variant in (0, 3) else None, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "mfcc_stream512_kernel<13,2,1,3,6,2,3,0,0> (+ scan + walk kernels; kernel_ms spans all three)" if args.variant in (0, 3) else "mfcc fused pass", "kernel_ms": ms_kernel,
                     "kernel_ms_stat": "median of the timed launches (hipEvents on the launch stream)",
                     "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_frame": bytes_per_frame},
        "roofline_flop": flop_roof,
    }
    if env is not None:
        result["env"] = env
        sclk = env.get("sustained_mfcc", {}).get("sclk_mhz", {}).get("mean")
        if sclk:
            # The pass sits on the package power cap and the clock the governor then grants differs by box (1.85 - 2.05 GHz over the boxes of
            # profiles/r05_box_spread.md: 10 % in time) while the CYCLES of a pass hold to 3 %: the clock under the sustained launch is the
            # yardstick.  (The calibration kernels are not: every box gives the same 137 - 143 TFLOP/s on the FMA chains.)
            result["value_normalised"] = {"value": value * NOMINAL_SCLK_MHZ / sclk, "by": "env.sustained_mfcc.sclk_mhz, to a nominal %d MHz" % NOMINAL_SCLK_MHZ,
                                          "ms_per_step": elapsed / args.steps * 1e3 * sclk / NOMINAL_SCLK_MHZ,
                                          "mega_cycles_per_step": elapsed / args.steps * 1e3 * sclk / 1e3}

    # ------------------------------------------------------------------ what GMM_UBM.extract_feature actually returns (GMM_UBM.py:89-93): [c, delta] 26-d,
    # scaled per utterance (sklearn.preprocessing.scale); the same resident audio
    if "ref26" in stages:
        rplan = api.MfccPlan(ctx, pkg.preset_sidekit(fs=fs, delta_order=1, cmvn=1))
        rfeat = torch.empty((n_frames, rplan.d_out), dtype=torch.float32, device=device)
        rplan.run(flat, seg, fseg, out=rfeat, variant=args.ref26_variant)
        rms = []
        for _ in range(max(2, min(args.steps, 5))):
            _, ms = rplan.run(flat, seg, fseg, out=rfeat, timing=True, variant=args.ref26_variant)
            rms.append(ms)
        r_ms = float(np.median(rms))
        r_bytes = n_utt * n_samp * 4 + n_frames * rplan.d_out * 4
        result["mfcc_ref26_cmvn"] = {
            "metric": "MFCC frames/s, the reference's extract_feature output: 13 cepstra + delta, per-utterance mean / variance scaling (26-d)",
            "value": n_frames / (r_ms * 1e-3), "unit": "frames/s", "d_out": rplan.d_out, "dtype": "f32",
            "roofline": {"bound": "hbm", "achieved": r_bytes / (r_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": r_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "mfcc_stream512_kernel<..., CM = 1> (the wave that walked an utterance sums its columns and rescales its own rows through L2; batches with multi-chunk utterances take cmvn_kernel instead)",
                         "kernel_ms": r_ms, "algorithmic_bytes_per_launch": r_bytes, "bytes_per_frame": tables.cfg.hop * 4 + rplan.d_out * 4}}
        del rplan, rfeat

    # ------------------------------------------------------------------ the host-fed path: what the reference-shaped callers use (GMM_UBM.py:24-50 reads
    # wav files into host arrays, :86-93 feeds them to the extractor).  A configs[1]-shaped batch in PINNED host memory through
    # ssp_mfcc_run(SSP_HOST): sliced copy-in / compute / copy-back pipeline inside the library; wall clock, next to what PCIe gives this box.
    if "hostfed" in stages and rank == 0:
        def stage_hostfed():   # (a function: an extra stage that fails — e.g. no page-locked memory to be had on a box — must not take the line with it)
            n_h = min(n_utt, args.hostfed_utts)
            hseg = api.Segments.from_lengths(ctx, np.full(n_h, n_samp, dtype=np.int64))
            hfseg = plan.frame_segments(hseg)
            pin_in = torch.empty(n_h * n_samp, dtype=torch.float32, pin_memory=True)
            pin_in.copy_(flat[:n_h * n_samp])
            pin_out = torch.empty((hfseg.total, plan.d_out), dtype=torch.float32, pin_memory=True)
            dev_tmp = torch.empty(n_h * n_samp, dtype=torch.float32, device=device)
            dev_out = torch.empty((hfseg.total, plan.d_out), dtype=torch.float32, device=device)
            torch.cuda.synchronize()

            def gbs(dst, src, reps=3):
                ts = []
                for _ in range(reps):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dst.copy_(src, non_blocking=True)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                return src.numel() * src.element_size() / float(np.median(ts)) / 1e9
            h2d = gbs(dev_tmp, pin_in)
            plan.run(dev_tmp, hseg, hfseg, out=dev_out, variant=args.variant)        # the device-pointer result of the same batch
            d2h = gbs(pin_out, dev_out)
            torch.cuda.synchronize()
            in_np, out_np = pin_in.numpy(), pin_out.numpy()

            def wall(src):
                ts = []
                plan.run(src, hseg, hfseg, out=out_np, variant=args.variant)
                for _ in range(3):
                    t0 = time.perf_counter()
                    plan.run(src, hseg, hfseg, out=out_np, variant=args.variant)     # (a host-pointer call returns when the features are in `out`)
                    ts.append(time.perf_counter() - t0)
                return float(np.median(ts))
            out_np[:] = 0
            w32 = wall(in_np)
            same32 = bool(torch.equal(torch.from_numpy(out_np).to(device), dev_out))
            in_b, out_b = n_h * n_samp * 4, int(hfseg.total) * plan.d_out * 4
            bound32 = max(in_b / (h2d * 1e9), out_b / (d2h * 1e9))                   # PCIe is full duplex: the longer direction bounds the pipeline
            # int16 PCM (utils/tools.py:45-47): half the bytes in; the device widens
            a16 = (audio[:n_h] * 20000.0).to(torch.int16).view(-1)
            pin16 = torch.empty(n_h * n_samp, dtype=torch.int16, pin_memory=True)
            pin16.copy_(a16)
            plan.run(a16.float(), hseg, hfseg, out=dev_out, variant=args.variant)
            torch.cuda.synchronize()
            out_np[:] = 0
            w16 = wall(pin16.numpy())
            same16 = bool(torch.equal(torch.from_numpy(out_np).to(device), dev_out))
            bound16 = max(in_b / 2 / (h2d * 1e9), out_b / (d2h * 1e9))
            result["mfcc_host_fed"] = {
                "metric": "MFCC frames/s, host-fed: a configs[1]-shaped batch in pinned host memory through ssp_mfcc_run(SSP_HOST) — sliced pipeline "
                          "(copy-in | compute | copy-back on three streams), wall clock of the call; never part of `value`",
                "value": hfseg.total / w32, "unit": "frames/s", "utterances": n_h, "wall_ms": w32 * 1e3, "dtype": "f32",
                "h2d_gbs": h2d, "d2h_gbs": d2h, "pcie_bound_ms": bound32 * 1e3, "frac_of_pcie_bound": bound32 / w32,
                "bits_equal_device_path": same32, "slice_mb": int(os.environ.get("SSP_HOST_SLICE_MB", "64")),
                "value_i16": hfseg.total / w16, "wall_ms_i16": w16 * 1e3, "pcie_bound_ms_i16": bound16 * 1e3, "frac_of_pcie_bound_i16": bound16 / w16,
                "bits_equal_device_path_i16": same16, "i16_over_f32": w32 / w16,
                "bound": "PCIe: max(input bytes / measured pinned H2D rate, feature bytes / measured D2H rate) of this box, this run"}
            # ---- the same call from PAGEABLE memory (what numpy hands over unless the caller allocates with api.pinned_empty): a fifth of the batch
            try:
                n_pg = max(1, n_h // 5)
                pg_seg = api.Segments.from_lengths(ctx, np.full(n_pg, n_samp, dtype=np.int64))
                pg_fseg = plan.frame_segments(pg_seg)
                pg_in = np.array(in_np[:n_pg * n_samp])                 # (a pageable copy)
                pg_out = np.empty((pg_fseg.total, plan.d_out), dtype=np.float32)
                plan.run(pg_in, pg_seg, pg_fseg, out=pg_out, variant=args.variant)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    plan.run(pg_in, pg_seg, pg_fseg, out=pg_out, variant=args.variant)
                    ts.append(time.perf_counter() - t0)
                wpg = float(np.median(ts))
                result["mfcc_host_fed"]["pageable"] = {"utterances": n_pg, "wall_ms": wpg * 1e3, "frames_per_s": pg_fseg.total / wpg,
                                                       "effective_gbs_in": n_pg * n_samp * 4 / wpg / 1e9,
                                                       "frac_of_pcie_bound": (bound32 * n_pg / n_h) / wpg}
                del pg_in, pg_out
            except Exception as e:
                result["mfcc_host_fed"]["pageable"] = {"error": repr(e)}
            # ---- the reference-shaped call itself: GMM_UBM.extract_feature(x, y) (GMM_UBM.py:72-118) on a list of int16 utterances as
            # utils.tools.read returns them (pageable memory, the shim's own context): list -> flat int16 -> ssp_mfcc_run_i16 -> 26-d scaled
            # features -> float64 rows per utterance.  Wall clock of the Python call (what a user of the reference's script waits for).
            try:
                from speech_signal_processing_amd import GMM_UBM as shim
                n_s = min(n_h, 2000)
                xs = [np.array(pin16[i * n_samp:(i + 1) * n_samp].numpy()) for i in range(n_s)]   # (pageable copies, one array per utterance)
                shim.extract_feature(xs[:50], [0] * 50)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    feat, _ = shim.extract_feature(xs, [0] * n_s)
                    ts.append(time.perf_counter() - t0)
                ws = float(np.median(ts))
                fr = int(sum(f.shape[0] for f in feat))
                xs32 = [x.astype(np.float32) for x in xs]
                t0 = time.perf_counter()
                shim.extract_feature(xs32, [0] * n_s)
                ws32 = time.perf_counter() - t0
                result["mfcc_host_fed"]["extract_feature_shim"] = {
                    "what": "GMM_UBM.extract_feature(list of %d int16 utterances of 3 s, pageable) -> list of (298, 26) float64, wall clock of the Python call" % n_s,
                    "wall_ms": ws * 1e3, "frames_per_s": fr / ws, "utterances": n_s, "wall_ms_float32_input": ws32 * 1e3,
                    "bytes_in": n_s * n_samp * 2, "bytes_out_float64": fr * 26 * 8}
                del xs, xs32, feat
            except Exception as e:  # (never lose the bench line to an extra)
                result["mfcc_host_fed"]["extract_feature_shim"] = {"error": repr(e)}
            del pin_in, pin_out, dev_tmp, dev_out, pin16, a16, in_np, out_np

        try:
            stage_hostfed()
        except Exception as e:
            result.setdefault("mfcc_host_fed", {})["error"] = repr(e)

    # ------------------------------------------------------------------ the reference-pinned dialect: in-repo MFCC (utils/processing.py:19-144)
    if "inrepo" in stages:
        result["mfcc_inrepo"] = {}
        for tag, ifs in (("16k", 16000), ("8k", 8000)):
            itab = pkg.preset_inrepo(ifs, 512, 256)
            iplan = api.MfccPlan(ctx, itab)
            # the same resident samples, read as `seconds` of audio at the dialect's rate (8 kHz: 200k utterances of 24000 samples)
            i_samp = int(round(args.seconds * ifs))
            i_utt = (n_utt * n_samp) // i_samp
            iseg = api.Segments.from_lengths(ctx, np.full(i_utt, i_samp, dtype=np.int64))
            ifseg = iplan.frame_segments(iseg)
            ifeat = torch.empty((ifseg.total, iplan.d_out), dtype=torch.float32, device=device)
            iplan.run(flat[: i_utt * i_samp], iseg, ifseg, out=ifeat, variant=args.inrepo_variant)
            ims = []
            barrier()
            torch.cuda.synchronize()
            t0i = time.perf_counter()
            for _ in range(max(2, min(args.steps, 5))):
                _, ms = iplan.run(flat[: i_utt * i_samp], iseg, ifseg, out=ifeat, timing=True, variant=args.inrepo_variant)
                ims.append(ms)
            torch.cuda.synchronize()
            barrier()
            i_elapsed = max_over_ranks(time.perf_counter() - t0i, device)
            i_ms = float(np.median(ims))
            i_bytes = i_utt * i_samp * 4 + ifseg.total * iplan.d_out * 4
            result["mfcc_inrepo"][tag] = {
                "metric": "in-repo MFCC frames/s (utils/processing.py:110-144: Hamming, |FFT|/L, 40 talkbox filters, log10, DCT-II, 13-d), "
                          "arithmetic pinned to the reference's own outputs (tests/golden/mfcc_inrepo.npz)",
                "value": ifseg.total * world * len(ims) / i_elapsed, "unit": "frames/s", "sample_rate": ifs, "frame": "512/256",
                "utterances_per_gpu": i_utt, "frames_per_gpu": int(ifseg.total), "dtype": "f32",
                "roofline": {"bound": "hbm", "achieved": i_bytes / (i_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": i_bytes / (i_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "kernel": "mfcc_stream512_kernel" if args.inrepo_variant in (0, 3) else "mfcc_fused512_kernel",
                             "kernel_ms": i_ms, "algorithmic_bytes_per_launch": i_bytes, "bytes_per_frame": 256 * 4 + 13 * 4}}
            del iplan, ifeat

    # ------------------------------------------------------------------ the librosa dialect of MFCC_DTW.MFCC_lib (MFCC_DTW.py:28-31): sr 8000, n_fft 2048, hop 512,
    # 128 mel filters, power_to_db with the utterance-wide top_db clamp, 13 coefficients; the same resident samples read as 8 kHz audio
    if "librosa" in stages:
        lplan = api.MfccPlan(ctx, pkg.preset_librosa(8000, 13))
        l_samp = int(round(args.seconds * 8000))
        l_utt = (n_utt * n_samp) // l_samp
        lseg = api.Segments.from_lengths(ctx, np.full(l_utt, l_samp, dtype=np.int64))
        lfseg = lplan.frame_segments(lseg)
        lfeat = torch.empty((lfseg.total, lplan.d_out), dtype=torch.float32, device=device)
        lplan.run(flat[: l_utt * l_samp], lseg, lfseg, out=lfeat)
        lms = []
        for _ in range(int(os.environ.get("SSP_BENCH_STAGE_REPS", "3"))):   # (tools/clock_probe.sh: hundreds of launches, to read the governor's steady state)
            _, ms = lplan.run(flat[: l_utt * l_samp], lseg, lfseg, out=lfeat, timing=True)
            lms.append(ms)
        l_ms = float(np.median(lms))
        l_bytes = l_utt * l_samp * 4 + lfseg.total * lplan.d_out * 4
        result["mfcc_librosa"] = {
            "metric": "librosa-dialect MFCC frames/s (MFCC_DTW.MFCC_lib: n_fft 2048 / hop 512, 128 mel, top_db 80, 13-d)", "value": lfseg.total / (l_ms * 1e-3),
            "unit": "frames/s", "utterances_per_gpu": l_utt, "frames_per_gpu": int(lfseg.total), "dtype": "f32",
            "roofline": {"bound": "hbm", "achieved": l_bytes / (l_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": l_bytes / (l_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "mfcc_stream2048_kernel (single-chunk utterances: the wave that walked an utterance also clamps its rows at its maximum - top_db and takes the DCT; multi-chunk batches add topdb_dct_chunk_kernel as a second pass)",
                         "kernel_ms": l_ms, "algorithmic_bytes_per_launch": l_bytes, "bytes_per_frame": 512 * 4 + 13 * 4}}
        del lplan, lfeat

    # ------------------------------------------------------------------ GMM-UBM scoring stage (configs[2])
    if "gmm" in stages:
        K, S, D = 64, 50, plan.d_out
        rng = np.random.default_rng(7)
        sub = feats[:: max(1, n_frames // 200000)]
        mean = sub.mean(0).double().cpu().numpy()
        std = sub.std(0).double().cpu().numpy()
        wts = rng.dirichlet(5 * np.ones(K))
        mu = mean + std * rng.standard_normal((K, D))
        cov = (std ** 2) * rng.uniform(0.5, 2.0, (K, D))
        mus = np.stack([mu] + [mu + 0.3 * std * rng.standard_normal((K, D)) for _ in range(S)])
        scorer = api.GmmScorer(ctx, np.stack([wts] * (S + 1)), mus, np.stack([cov] * (S + 1)), has_ubm=True)
        g_steps = max(2, min(args.steps, 5))
        flop = 4.0 * D * K * n_frames * (S + 1)

        def run_gmm(precision):
            r = scorer.score(feats, fseg, precision=precision)  # warm-up (allocates the per-frame scratch)
            gms = []
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(g_steps):
                r = scorer.score(feats, fseg, precision=precision, timing=True)
                gms.append(r["kernel_ms"])
                # RCCL all-gather of the compact per-utterance record (int32 argmax, fp32 best, fp32 ubm): 12 B per utterance
                gathered = all_gather_rows(decision_records(r))
            torch.cuda.synchronize()
            barrier()
            g_elapsed = max_over_ranks(time.perf_counter() - t0, device)
            return r, float(np.mean(gms)), g_elapsed, int(gathered.shape[0])

        r0, g_ms, g_elapsed, n_gath = run_gmm(0)
        fscores = n_frames * (S + 1) * world * g_steps
        result["gmm"] = {
            "metric": "GMM frame-scores/s (diag, K=%d, D=%d, %d models)" % (K, D, S + 1),
            "value": fscores / g_elapsed, "unit": "frame-scores/s", "ms_per_step": g_elapsed / g_steps * 1e3,
            "steps": g_steps, "dtype": "f32", "gathered_rows": n_gath, "record_bytes": 12,
            "record": "(int32 argmax, fp32 best score - ubm, fp32 ubm) per utterance, one all-gather",
            "config": {"workload": "configs[2]: the MFCC stream above vs 64-mix diag UBM + 50 speaker GMMs"},
            "roofline": {"bound": "mfma", "achieved": flop / (g_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF,
                         "unit": "TFLOP/s", "frac": flop / (g_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, "traffic": None,
                         "kernel": "gmm_loglik<fused epilogue> (v_mfma_f32_32x32x2_f32; per-utterance piece sums leave the kernel, [models x frames] "
                                   "never reaches HBM) + piece_reduce", "kernel_ms": g_ms,
                         "algorithmic_flop_per_launch": flop},
        }
        mb, msrc = mfma_busy_from_profile("gmm_mfma_util.json", "gmm_loglik_kernel", GMM_SOURCES)
        result["gmm"]["roofline"]["mfma_busy"], result["gmm"]["roofline"]["mfma_busy_source"] = mb, msrc
        if "hostfed" in stages and rank == 0:
            # the same scoring HOST-FED (GMM_UBM.py:181-197 hands host arrays): a quarter of the batch's features in pinned memory through
            # ssp_gmm_score(SSP_HOST) — rows copied in ahead of the kernels that score them; wall clock against copy + kernels
            try:
                u_q = max(2, n_utt // 4)
                f_q = int(fseg.offsets[u_q])
                pin_f = torch.empty((f_q, D), dtype=torch.float32, pin_memory=True)
                pin_f.copy_(feats[:f_q])
                seg_q = api.Segments.from_lengths(ctx, np.diff(fseg.offsets[:u_q + 1]))
                torch.cuda.synchronize()
                dev_q = scorer.score(feats[:f_q], seg_q, precision=0, timing=True)
                tcp = []
                tmp = torch.empty_like(feats[:f_q])
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tmp.copy_(pin_f, non_blocking=True)
                    torch.cuda.synchronize()
                    tcp.append(time.perf_counter() - t0)
                copy_ms = float(np.median(tcp)) * 1e3
                fnp = pin_f.numpy()
                scorer.score(fnp, seg_q, precision=0)
                tw = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    hq = scorer.score(fnp, seg_q, precision=0)
                    tw.append(time.perf_counter() - t0)
                wall_ms = float(np.median(tw)) * 1e3
                result["gmm_host_fed"] = {
                    "metric": "GMM frame-scores/s, host-fed: %d utterances' features in pinned host memory through ssp_gmm_score(SSP_HOST), fp32 path; wall clock of the call" % u_q,
                    "value": f_q * (S + 1) / (wall_ms * 1e-3), "unit": "frame-scores/s", "wall_ms": wall_ms, "copy_in_ms": copy_ms,
                    "kernel_ms_device_path": dev_q["kernel_ms"], "sum_ms": copy_ms + dev_q["kernel_ms"],
                    "overlap": max(copy_ms, dev_q["kernel_ms"]) / wall_ms,
                    "scores_equal_device_path": bool(torch.equal(torch.from_numpy(hq["scores"]).to(device), dev_q["scores"]))}
                del pin_f, tmp, fnp, hq, dev_q
            except Exception as e:  # (never lose the bench line to an extra)
                result["gmm_host_fed"] = {"error": repr(e)}
        # bf16 hi/lo split path: 3 bf16 MFMAs per k-step, same tolerance class; priced against the dense bf16 peak with
        # the ALGORITHMIC flops (the kernel executes 3x as many).  precision = 1 scores every utterance whose top-2 margin lies
        # inside the split-precision error band again on the fp32 path (timed with it), so its arg-max is the fp32 path's
        # (precision 3 = the calibrated band rounds 2 - 3 measured under this key; precision 1 = the PROVEN bound, reported beside it:
        #  about 100 x wider, so more utterances are scored twice)
        r1, g1_ms, g1_elapsed, _ = run_gmm(3)
        n_rescored = scorer.last_rescored
        rp, gp_ms, gp_elapsed, _ = run_gmm(1)
        n_rescored_proven = scorer.last_rescored
        r2 = scorer.score(feats, fseg, precision=2)
        sc0, sc1 = r0["scores"], r1["scores"]
        result["gmm_bf16x3"] = {
            "metric": "GMM frame-scores/s, bf16x3 split-precision MFMA path", "value": fscores / g1_elapsed,
            "unit": "frame-scores/s", "ms_per_step": g1_elapsed / g_steps * 1e3, "steps": g_steps, "dtype": "bf16x3->f32",
            "max_abs_diff_vs_fp32_path": float((sc1 - sc0).abs().max().item()),
            "max_abs_score": float(sc0.abs().max().item()),
            "argmax_agreement_vs_fp32_path": float((r0["argmax"] == r1["argmax"]).float().mean().item()),
            "argmax_mismatches_vs_fp32_path": int((r0["argmax"] != r1["argmax"]).sum().item()),
            "utterances_rescored_in_fp32": int(n_rescored), "utterances": int(n_utt),
            "argmax_agreement_without_rescoring": float((r0["argmax"] == r2["argmax"]).float().mean().item()),
            "roofline": {"bound": "mfma", "achieved": flop / (g1_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": flop / (g1_ms * 1e-3) / 1e12 / 2500.0, "traffic": None,
                         "kernel": "gmm_loglik_bf16x3<fused epilogue> (v_mfma_f32_32x32x16_bf16, 3 MFMAs per k-step) + piece_reduce + fp32 re-scoring of close calls",
                         "kernel_ms": g1_ms, "algorithmic_flop_per_launch": flop, "executed_mfma_flop_per_launch": 3 * flop * 80.0 / 78.0},
        }
        result["gmm_bf16x3"]["band"] = "calibrated (heuristic): 8e-5 (|UBM score| + 1) — ssp_gmm_score precision 3"
        result["gmm_bf16x3"]["precision"] = 3  # (until round 3 this key measured precision 1, then the only band)
        result["gmm_bf16x3_proven_band"] = {
            "metric": "GMM frame-scores/s, bf16x3 MFMA + fp32 re-scoring of every utterance whose top-2 margin is inside the PROVEN error bound (ssp_gmm_score precision 1)",
            "value": fscores / gp_elapsed, "unit": "frame-scores/s", "kernel_ms": gp_ms, "dtype": "bf16x3->f32",
            "utterances_rescored_in_fp32": int(n_rescored_proven), "utterances": int(n_utt), "precision": 1,
            "argmax_mismatches_vs_fp32_path": int((r0["argmax"] != rp["argmax"]).sum().item()),
            "roofline": {"bound": "mfma", "achieved": flop / (gp_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": flop / (gp_ms * 1e-3) / 1e12 / 2500.0, "traffic": None, "kernel_ms": gp_ms, "algorithmic_flop_per_launch": flop}}
        # precision "auto" (ssp_gmm_score precision 4): the proven-band guarantee at the cost of the cheaper of the split and the fp32 path —
        # a pilot on the first 2 % of the utterances prices the re-scoring (timed with the call)
        ra, ga_ms, ga_elapsed, _ = run_gmm(4)
        auto_info = scorer.last_auto
        result["gmm_auto"] = {
            "metric": "GMM frame-scores/s, precision auto (pilot on 2 % of the utterances, then the proven-band bf16x3 path or the fp32 path)",
            "value": fscores / ga_elapsed, "unit": "frame-scores/s", "kernel_ms": ga_ms, "dtype": "bf16x3->f32 | f32", **auto_info,
            "argmax_mismatches_vs_fp32_path": int((r0["argmax"] != ra["argmax"]).sum().item()),
            "ratio_to_best_fixed": ga_ms / min(g_ms, gp_ms), "fixed_ms": {"fp32": g_ms, "proven_band": gp_ms}}
        auto_ratios, auto_mism = [ga_ms / min(g_ms, gp_ms)], int((r0["argmax"] != ra["argmax"]).sum().item())
        r = r1
        del scorer, r, r0, r1, r2, rp, ra
        # ---- the same shape with the speaker models moved closer to the UBM (offset 0.3 std above): what the exact-arg-max guarantee
        # of the split-precision path costs when many utterances are close calls (verdict r4: the rows above are the best case)
        if "closecalls" in stages:
            pts = []
            for off in CLOSE_GMM_OFFSETS:
                rngc = np.random.default_rng(70)
                musc = np.stack([mu] + [mu + off * std * rngc.standard_normal((K, D)) for _ in range(S)])
                sc = api.GmmScorer(ctx, np.stack([wts] * (S + 1)), musc, np.stack([cov] * (S + 1)), has_ubm=True)
                ref = sc.score(feats, fseg, precision=0, timing=True)
                pt = {"speaker_offset_std": off, "fp32_kernel_ms": ref["kernel_ms"], "utterances": int(n_utt)}
                for prec, name in ((3, "heuristic_band"), (1, "proven_band")):
                    sc.score(feats, fseg, precision=prec)
                    ms2 = []
                    for _ in range(2):
                        rr = sc.score(feats, fseg, precision=prec, timing=True)
                        ms2.append(rr["kernel_ms"])
                    pt[name] = {"kernel_ms": float(np.mean(ms2)), "utterances_rescored_in_fp32": int(sc.last_rescored),
                                "fraction_rescored": float(sc.last_rescored) / n_utt, "speedup_vs_fp32": ref["kernel_ms"] / float(np.mean(ms2)),
                                "argmax_mismatches_vs_fp32_path": int((rr["argmax"] != ref["argmax"]).sum().item())}
                sc.score(feats, fseg, precision=4)
                ms2 = []
                for _ in range(2):
                    rr = sc.score(feats, fseg, precision=4, timing=True)
                    ms2.append(rr["kernel_ms"])
                best = min(pt["fp32_kernel_ms"], pt["proven_band"]["kernel_ms"])
                pt["auto"] = {"kernel_ms": float(np.mean(ms2)), **sc.last_auto, "ratio_to_best_fixed": float(np.mean(ms2)) / best,
                              "argmax_mismatches_vs_fp32_path": int((rr["argmax"] != ref["argmax"]).sum().item())}
                auto_ratios.append(pt["auto"]["ratio_to_best_fixed"])
                auto_mism += pt["auto"]["argmax_mismatches_vs_fp32_path"]
                pts.append(pt)
                del sc, ref, rr
            result["gmm_bf16x3_close_calls"] = {
                "what": "configs[2] shape, speaker means at `speaker_offset_std` x std from the UBM's (0.3 in the rows above): the share of utterances whose "
                        "top-2 margin falls inside the error band — scored again in fp32 on their candidate models — and what the pass then costs; "
                        "arg-max equality with the fp32 path is checked on every utterance; `auto` = precision 4 (pilot, then proven band or fp32), "
                        "`ratio_to_best_fixed` = its time over min(fp32, proven band) at the same point",
                "points": pts}
        result["gmm_auto"]["worst_ratio_to_best_fixed"] = float(max(auto_ratios))
        result["gmm_auto"]["points_checked"] = len(auto_ratios)
        result["gmm_auto"]["mismatches_vs_fp32"] = auto_mism

    # ------------------------------------------------------------------ configs[3] shape: 512-mix UBM + 1251 speaker models, a measured sample
    if "gmm4" in stages:
        K, S, D = 512, 1251, plan.d_out
        u4 = min(args.gmm4_utts, n_utt)
        f4 = int(fseg.offsets[u4])
        rng = np.random.default_rng(17)
        sub = feats[:: max(1, n_frames // 200000)]
        mean, std = sub.mean(0).double().cpu().numpy(), sub.std(0).double().cpu().numpy()
        wts = rng.dirichlet(5 * np.ones(K))
        mu = mean + std * rng.standard_normal((K, D))
        cov = (std ** 2) * rng.uniform(0.5, 2.0, (K, D))
        mus = np.empty((S + 1, K, D))
        mus[0] = mu
        for i in range(S):
            mus[i + 1] = mu + 0.3 * std * rng.standard_normal((K, D))
        scorer4 = api.GmmScorer(ctx, np.broadcast_to(wts, (S + 1, K)), mus, np.broadcast_to(cov, (S + 1, K, D)), has_ubm=True)
        del mus
        seg4 = api.Segments.from_lengths(ctx, np.diff(fseg.offsets[:u4 + 1]))
        out4 = {}
        for prec, tag in ((0, "f32"), (3, "bf16x3"), (1, "bf16x3_proven_band"), (4, "auto")):  # (3: calibrated band; 1: proven bound, more re-scoring; 4: pilot, then 1 or 0)
            scorer4.score(feats[:f4], seg4, precision=prec)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r4 = scorer4.score(feats[:f4], seg4, precision=prec, timing=True)
            gathered = all_gather_rows(decision_records(r4))
            torch.cuda.synchronize()
            barrier()
            dt4 = max_over_ranks(time.perf_counter() - t0, device)
            flop4 = 4.0 * D * K * f4 * (S + 1)
            out4[tag] = {"value": f4 * (S + 1) * world / dt4, "unit": "frame-scores/s", "kernel_ms": r4["kernel_ms"],
                         "tflops": flop4 / (r4["kernel_ms"] * 1e-3) / 1e12, "gathered_rows": int(gathered.shape[0])}
            if prec == 0:
                am4 = r4["argmax"].clone()
            else:
                out4[tag]["utterances_rescored_in_fp32"] = int(scorer4.last_rescored)
                out4[tag]["argmax_mismatches_vs_fp32_path"] = int((am4 != r4["argmax"]).sum().item())
                if prec == 4:
                    out4[tag].update(scorer4.last_auto)
        result["gmm_cfg3_shape"] = {
            "metric": "GMM frame-scores/s at the configs[3] shape (K=512, 1251 speakers + UBM, D=%d), sample of %d utterances per GPU" % (D, u4),
            "frames_per_gpu": f4, "full_config_utterances_per_gpu": 150000, "fraction_of_full_config": u4 / 150000.0,
            "measured_s_for_this_sample": {t: out4[t]["kernel_ms"] * 1e-3 for t in out4},
            "extrapolated_full_config_s_per_gpu": {t: 150000.0 / u4 * out4[t]["kernel_ms"] * 1e-3 for t in out4}, **out4}
        # ---- configs[3]'s FULL per-GPU share, measured: 150 000 utterances x 298 frames against the 1252 models on the bf16x3 path with
        # fp32 re-scoring of close calls (the fp32 path would take ~35 s per GPU: it stays a 12 000-utterance sample above).  The 50 000
        # utterances beyond the resident configs[1] batch are synthesised and run through the same MFCC plan here, outside the timing.
        if args.gmm4_full and n_utt >= 100000 and n_samp == 48000:
            U_full = 150000
            extra = U_full - n_utt
            T_utt = int(fseg.offsets[1])
            feats_full = torch.empty((U_full * T_utt, plan.d_out), dtype=torch.float32, device=device)
            feats_full[:n_frames] = feats
            if extra > 0:
                audio_x = synth_audio_device(torch, extra, n_samp, fs, seed=4321 + rank, device=device)
                seg_x = api.Segments.from_lengths(ctx, np.full(extra, n_samp, dtype=np.int64))
                fseg_x = plan.frame_segments(seg_x)
                plan.run(audio_x.view(-1), seg_x, fseg_x, out=feats_full[n_frames:], variant=args.variant)
                torch.cuda.synchronize()
                del audio_x
            seg_full = api.Segments.from_lengths(ctx, np.full(U_full, T_utt, dtype=np.int64))
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rf = scorer4.score(feats_full, seg_full, precision=3, timing=True)
            gathered = all_gather_rows(decision_records(rf))
            torch.cuda.synchronize()
            barrier()
            dtf = max_over_ranks(time.perf_counter() - t0, device)
            ff = U_full * T_utt
            flopf = 4.0 * D * K * ff * (S + 1)
            result["gmm_cfg3_shape"]["bf16x3_full_share"] = {
                "metric": "configs[3] per-GPU share, measured whole: %d utterances x %d frames vs 512-mix UBM + 1251 speaker models" % (U_full, T_utt),
                "utterances_per_gpu": U_full, "frames_per_gpu": ff, "fraction_of_full_config": 1.0,
                "measured_s": dtf, "kernel_s": rf["kernel_ms"] * 1e-3, "value": ff * (S + 1) * world / dtf, "unit": "frame-scores/s",
                "tflops_algorithmic": flopf / (rf["kernel_ms"] * 1e-3) / 1e12, "frac_of_bf16_peak": flopf / (rf["kernel_ms"] * 1e-3) / 1e12 / 2500.0,
                "utterances_rescored_in_fp32": int(scorer4.last_rescored), "gathered_rows": int(gathered.shape[0]),
                "argmax_mismatches_vs_fp32_sample": int((am4 != rf["argmax"][:u4]).sum().item()), "fp32_sample_utterances": u4,
                "dtype": "bf16x3->f32", "band": "calibrated (heuristic): ssp_gmm_score precision 3"}
            del feats_full, rf
        del scorer4, r4

    # ------------------------------------------------------------------ cosine stage (configs[4])
    if "cosine" in stages:
        N, S, d = 1000000, 1251, 256
        gen = torch.Generator(device=device)
        gen.manual_seed(11 + rank)
        Cn = torch.randn((S, d), generator=gen, device=device)
        lab = torch.randint(0, S, (N,), generator=gen, device=device)
        X = Cn[lab] + 0.7 * torch.randn((N, d), generator=gen, device=device)
        rc = api.cosine_identify(ctx, X, Cn)
        c_steps = max(2, min(args.steps, 5))
        cms = []
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(c_steps):
            rc = api.cosine_identify(ctx, X, Cn, timing=True)
            cms.append(rc["kernel_ms"])
        torch.cuda.synchronize()
        barrier()
        c_elapsed = max_over_ranks(time.perf_counter() - t0, device)
        acc = float((rc["argmin"].long() == lab).float().mean().item())
        c_ms = float(np.mean(cms))
        flop = 2.0 * d * N * S
        result["cosine"] = {
            "metric": "cosine pair-scores/s (N=1e6, S=1251, d=256)", "value": N * S * world * c_steps / c_elapsed,
            "unit": "pair-scores/s", "ms_per_step": c_elapsed / c_steps * 1e3, "steps": c_steps, "dtype": "f32",
            "argmin_accuracy": acc,
            "roofline": {"bound": "mfma", "achieved": flop / (c_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": flop / (c_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, "traffic": None, "kernel": "cosine_reg_kernel<32, false> (d <= 256: embeddings register-resident, centroid tiles by LDS-DMA)",
                         "kernel_ms": c_ms, "algorithmic_flop_per_launch": flop},
        }
        mb, msrc = mfma_busy_from_profile("cosine_mfma_util.json", "cosine_reg_kernel", COSINE_SOURCES)
        result["cosine"]["roofline"]["mfma_busy"], result["cosine"]["roofline"]["mfma_busy_source"] = mb, msrc
        if "hostfed" in stages and rank == 0:
            # host-fed (d_vector.py:315-319 hands host arrays): the embeddings in pinned memory through ssp_cosine_identify(SSP_HOST), arg-min + minimum
            try:
                pin_x = torch.empty((N, d), dtype=torch.float32, pin_memory=True)
                pin_x.copy_(X)
                tmpx = torch.empty_like(X)
                tcp = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    tmpx.copy_(pin_x, non_blocking=True)
                    torch.cuda.synchronize()
                    tcp.append(time.perf_counter() - t0)
                copy_ms = float(np.median(tcp)) * 1e3
                xnp, cnp = pin_x.numpy(), Cn.cpu().numpy()
                api.cosine_identify(ctx, xnp, cnp)
                tw = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    hc = api.cosine_identify(ctx, xnp, cnp)
                    tw.append(time.perf_counter() - t0)
                wall_ms = float(np.median(tw)) * 1e3
                result["cosine_host_fed"] = {
                    "metric": "cosine pair-scores/s, host-fed: 1e6 x 256 embeddings in pinned host memory through ssp_cosine_identify(SSP_HOST), fp32 path, arg-min + minimum; wall clock of the call",
                    "value": N * S / (wall_ms * 1e-3), "unit": "pair-scores/s", "wall_ms": wall_ms, "copy_in_ms": copy_ms, "kernel_ms_device_path": c_ms,
                    "sum_ms": copy_ms + c_ms, "overlap": max(copy_ms, c_ms) / wall_ms,
                    "argmin_equals_device_path": bool((torch.from_numpy(hc["argmin"]).to(device) == rc["argmin"]).all().item())}
                del pin_x, tmpx, xnp, hc
            except Exception as e:
                result["cosine_host_fed"] = {"error": repr(e)}
        # split precision (ssp_cosine_identify2 precision = 1): bf16 x 3 MFMA sweep keeping the two best cosines + fp32 re-scoring of the
        # rows inside the proven error band (device-side list, no host round trip); the fp32 path's arg-min on every row
        am0 = rc["argmin"].clone()
        r16 = api.cosine_identify(ctx, X, Cn, precision=1)
        ms16 = []
        for _ in range(c_steps):
            r16 = api.cosine_identify(ctx, X, Cn, timing=True, precision=1)
            ms16.append(r16["kernel_ms"])
        torch.cuda.synchronize()
        tw0 = time.perf_counter()
        for _ in range(c_steps):  # (what a caller who wants the arg-min waits: asynchronous calls, one wait at the end)
            api.cosine_identify(ctx, X, Cn, precision=1, counts=False)
        torch.cuda.synchronize()
        wall16 = (time.perf_counter() - tw0) / c_steps * 1e3
        c16 = float(np.mean(ms16))
        result["cosine_bf16x3"] = {
            "metric": "cosine pair-scores/s, split precision (bf16 x 3 MFMA + fp32 re-scoring of close calls), arg-min only", "value": N * S / (c16 * 1e-3),
            "unit": "pair-scores/s", "dtype": "bf16x3 (fp32 accumulate)", "argmin_equals_fp32_path": bool((r16["argmin"] == am0).all().item()),
            "rows_rescored_fp32": int(r16["rescored"]), "speedup_vs_fp32": c_ms / c16, "precision": 1, "wall_ms_per_call": wall16,
            "roofline": {"bound": "mfma", "achieved": flop / (c16 * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": flop / (c16 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, "frac_executed": 3.0 * flop / (c16 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                         "traffic": None, "kernel": "cosine_bf16x3_kernel<16> + cosine_reg_kernel<32> on the listed rows", "kernel_ms": c16,
                         "algorithmic_flop_per_launch": flop, "executed_flop_per_launch": 3.0 * flop}}
        # the cascade (precision 2): a sweep on the hi parts alone (one MFMA per k-step, bound 4e-3) in front; its close calls go to the
        # bf16 x 3 sweep, that one's to fp32.  How many rows each later stage takes depends on the data (here: well-separated embeddings)
        rcs = api.cosine_identify(ctx, X, Cn, precision=2)
        msc = []
        for _ in range(c_steps):
            rcs = api.cosine_identify(ctx, X, Cn, timing=True, precision=2)
            msc.append(rcs["kernel_ms"])
        torch.cuda.synchronize()
        tw0 = time.perf_counter()
        for _ in range(c_steps):
            api.cosine_identify(ctx, X, Cn, precision=2, counts=False)
        torch.cuda.synchronize()
        wallc = (time.perf_counter() - tw0) / c_steps * 1e3
        cc = float(np.mean(msc))
        result["cosine_bf16_cascade"] = {
            "metric": "cosine pair-scores/s, cascade (bf16 sweep -> bf16 x 3 on its close calls -> fp32 on theirs), arg-min only", "value": N * S / (cc * 1e-3),
            "unit": "pair-scores/s", "dtype": "bf16 / bf16x3 / f32 (fp32 accumulate)", "argmin_equals_fp32_path": bool((rcs["argmin"] == am0).all().item()),
            "rows_to_bf16x3": int(rcs["split_rows"]), "rows_rescored_fp32": int(rcs["rescored"]), "speedup_vs_fp32": c_ms / cc, "precision": 2, "wall_ms_per_call": wallc,
            "data_dependence": "synthetic embeddings with a smallest top-2 cosine gap of 0.44: no row needs a later stage; on data with closer "
                               "calls the later stages take the rows inside 8e-3 / 2.7e-4",
            "roofline": {"bound": "mfma", "achieved": flop / (cc * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": flop / (cc * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, "traffic": None,
                         "kernel": "cosine_bf16x3_kernel<16, 1> (+ <16, 3> and cosine_reg_kernel<32> on the listed rows)", "kernel_ms": cc,
                         "algorithmic_flop_per_launch": flop}}

        # precision "auto" (ssp_cosine_identify2 precision 3): pilot on 2 % of the rows, then the cascade, the bf16x3 sweep or fp32
        rau = api.cosine_identify(ctx, X, Cn, precision=3)
        msa = []
        for _ in range(c_steps):
            rau = api.cosine_identify(ctx, X, Cn, timing=True, precision=3)
            msa.append(rau["kernel_ms"])
        ca = float(np.mean(msa))
        result["cosine_auto"] = {
            "metric": "cosine pair-scores/s, precision auto (pilot on 2 % of the rows, then cascade | bf16x3 | fp32), arg-min only", "value": N * S / (ca * 1e-3),
            "unit": "pair-scores/s", "kernel_ms": ca, "dtype": "bf16 / bf16x3 / f32", **rau["auto"], "argmin_equals_fp32_path": bool((rau["argmin"] == am0).all().item()),
            "ratio_to_best_fixed": ca / min(c_ms, c16, cc), "fixed_ms": {"fp32": c_ms, "bf16x3": c16, "cascade": cc}}
        cos_ratios, cos_mism = [ca / min(c_ms, c16, cc)], int((rau["argmin"] != am0).sum().item())

        if "closecalls" in stages:
            pts = []
            Zc = torch.randn((N, d), generator=gen, device=device)
            for noise in CLOSE_COS_NOISE:
                Xc = Cn[lab] + noise * Zc
                ref = api.cosine_identify(ctx, Xc, Cn, timing=True)
                pt = {"embedding_noise": noise, "fp32_kernel_ms": ref["kernel_ms"], "rows": N,
                      "argmin_accuracy": float((ref["argmin"].long() == lab).float().mean().item())}
                for prec, name in ((1, "bf16x3"), (2, "cascade")):
                    api.cosine_identify(ctx, Xc, Cn, precision=prec)
                    ms2 = []
                    for _ in range(3):
                        rr = api.cosine_identify(ctx, Xc, Cn, timing=True, precision=prec)
                        ms2.append(rr["kernel_ms"])
                    pt[name] = {"kernel_ms": float(np.mean(ms2)), "rows_rescored_fp32": int(rr["rescored"]), "speedup_vs_fp32": ref["kernel_ms"] / float(np.mean(ms2)),
                                "argmin_equals_fp32_path": bool((rr["argmin"] == ref["argmin"]).all().item())}
                    if prec == 2:
                        pt[name]["rows_to_bf16x3"] = int(rr["split_rows"])
                        pt[name]["fraction_to_bf16x3"] = float(rr["split_rows"]) / N
                api.cosine_identify(ctx, Xc, Cn, precision=3)
                ms2 = []
                for _ in range(3):
                    rr = api.cosine_identify(ctx, Xc, Cn, timing=True, precision=3)
                    ms2.append(rr["kernel_ms"])
                best = min(pt["fp32_kernel_ms"], pt["bf16x3"]["kernel_ms"], pt["cascade"]["kernel_ms"])
                pt["auto"] = {"kernel_ms": float(np.mean(ms2)), **rr["auto"], "ratio_to_best_fixed": float(np.mean(ms2)) / best,
                              "argmin_equals_fp32_path": bool((rr["argmin"] == ref["argmin"]).all().item())}
                cos_ratios.append(pt["auto"]["ratio_to_best_fixed"])
                cos_mism += int((rr["argmin"] != ref["argmin"]).sum().item())
                pts.append(pt)
                del Xc
            result["cosine_close_calls"] = {
                "what": "configs[4] shape, embeddings = centroid + `embedding_noise` x N(0, 1) (0.7 in the rows above: no close call at all): rows whose two "
                        "best cosines lie inside a stage's proven band go to the next stage; arg-min equality with the fp32 path on every row; "
                        "`auto` = precision 3, `ratio_to_best_fixed` = its time over min(fp32, bf16x3, cascade) at the same point",
                "points": pts}
            del Zc
        result["cosine_auto"]["worst_ratio_to_best_fixed"] = float(max(cos_ratios))
        result["cosine_auto"]["points_checked"] = len(cos_ratios)
        result["cosine_auto"]["mismatches_vs_fp32"] = cos_mism

    # ------------------------------------------------------------------ widened stages; reported, not part of `value`
    if "em" in stages:
        K, D = 64, plan.d_out
        rng = np.random.default_rng(9)
        n_em = min(n_frames, 3000000)
        sub = feats[:n_em]
        mean, std = sub.mean(0).double().cpu().numpy(), sub.std(0).double().cpu().numpy()
        w0 = rng.dirichlet(5 * np.ones(K))
        mu0 = mean + std * rng.standard_normal((K, D))
        cov0 = (std ** 2) * rng.uniform(0.5, 2.0, (K, D))
        api.gmm_em_stats(ctx, w0, mu0, cov0, sub)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = api.gmm_em_stats(ctx, w0, mu0, cov0, sub, timing=True)
        dt = time.perf_counter() - t0
        result["gmm_em"] = {"metric": "GMM EM iteration (E step + M sums), frames/s (K=%d, D=%d)" % (K, D), "value": n_em / dt,
                            "unit": "frames/s", "kernel_ms": r["kernel_ms"], "frames": n_em,
                            "tflops": 12.0 * D * K * n_em / r["kernel_ms"] / 1e9, "dtype": "f32",
                            "roofline": {"bound": "mfma", "achieved": 12.0 * D * K * n_em / r["kernel_ms"] / 1e9, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                         "frac": 12.0 * D * K * n_em / r["kernel_ms"] / 1e9 / MFMA_F32_PEAK_TF, "traffic": None,
                                         "kernel": "gmm_em_acc_mfma_kernel<3, fused log-sum-exp> (v_mfma_f32_32x32x2_f32: log-probabilities and responsibility-weighted sums in one pass, D <= 47, K <= 64) + gmm_em_reduce_kernel",
                                         "kernel_ms": r["kernel_ms"],
                                         "algorithmic_flop_per_launch": 12.0 * D * K * n_em,
                                         "flop_per_frame": "12 D K: log-probability 4 D K (two D x K multiply-adds) + responsibility-weighted sums of x and x^2 8 D K"}}
    if "dnn" in stages:
        # the reference's fully connected d-vector network (d_vector.py:171-189) as one packed object: input layer (1274 -> 256) on the
        # tiled MFMA GEMM, the three following layers chained inside one kernel with the activations kept in registers
        Nd = 500000
        gen = torch.Generator(device=device)
        gen.manual_seed(5 + rank)
        dims = [1274, 256, 256, 256, 256]
        Xd = torch.randn((Nd, dims[0]), generator=gen, device=device)
        Wd = [(torch.randn((dims[i + 1], dims[i]), generator=gen, device=device) / dims[i] ** 0.5).cpu().numpy() for i in range(4)]
        bd = [(0.1 * torch.randn(dims[i + 1], generator=gen, device=device)).cpu().numpy() for i in range(4)]
        net = api.DnnForward(ctx, [(Wd[i], bd[i], i < 3) for i in range(4)])
        dms = []
        for rep in range(4):
            h, ms = net.forward(Xd, timing=True)
            if rep:
                dms.append(ms)
        tot = float(np.median(dms))
        flop = 2.0 * Nd * sum(dims[i] * dims[i + 1] for i in range(4))
        result["dvector_dnn"] = {"metric": "d-vector network forward 1274->256x4 (ssp_dnn_forward: input-layer GEMM + 3 layers chained in registers), embeddings/s",
                                 "value": Nd / tot * 1e3, "unit": "embeddings/s", "kernel_ms": tot, "embeddings": Nd, "dtype": "f32",
                                 "roofline": {"bound": "mfma", "achieved": flop / tot / 1e9, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                              "frac": flop / tot / 1e9 / MFMA_F32_PEAK_TF, "traffic": None,
                                              "kernel": "dense_kernel (v_mfma_f32_32x32x2_f32) + dnn_chain_kernel (v_mfma_f32_16x16x4_f32)"}}
        del Xd, h, net
    if "dvec" in stages:
        # the d-vector recogniser end to end on device-resident audio (d_vector.py:80-115 chunking + sidekit MFCC -> (98, 13) ->
        # 1274-d input -> Dense(256) x 4 -> cosine against 1251 enrolment centroids -> arg-min), zero-copy between the stages
        n_ch_utt = min(n_utt, 100000)
        chunks = audio[:n_ch_utt].reshape(-1, fs)                      # 1 s chunks: (3 n_utt, 16000)
        n_ch = int(chunks.shape[0])
        plan13 = api.MfccPlan(ctx, pkg.preset_sidekit(fs=fs, delta_order=0, cmvn=0))
        seg13 = api.Segments.from_lengths(ctx, np.full(n_ch, fs, dtype=np.int64))
        fseg13 = plan13.frame_segments(seg13)
        gen = torch.Generator(device=device)
        gen.manual_seed(23 + rank)
        dims = [98 * 13, 256, 256, 256, 256]
        We = [(torch.randn((dims[i + 1], dims[i]), generator=gen, device=device) / dims[i] ** 0.5).cpu().numpy() for i in range(4)]
        net_e = api.DnnForward(ctx, [(We[i], None, i < 3) for i in range(4)])
        Ce = torch.randn((1251, 256), generator=gen, device=device)
        f13 = torch.empty((fseg13.total, 13), dtype=torch.float32, device=device)

        def run_e2e(cos_precision=0):
            plan13.run(chunks.reshape(-1), seg13, fseg13, out=f13)
            h = net_e.forward(f13.view(n_ch, 98 * 13))
            return api.cosine_identify(ctx, h, Ce, minval=False, precision=cos_precision)["argmin"]

        def timed(cos_precision):
            run_e2e(cos_precision)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run_e2e(cos_precision)
            torch.cuda.synchronize()
            barrier()
            return out, max_over_ranks(time.perf_counter() - t0, device)
        ids, dt = timed(0)
        ids2, dt2 = timed(2)   # the same with the cosine stage as the split-precision cascade (same indices, proven)
        result["dvector_pipeline"] = {"metric": "d-vector recogniser end to end: 1 s chunks -> MFCC 98x13 -> Dense 1274-256x4 -> cosine vs 1251 -> arg-min",
                                      "value": n_ch * world / dt, "unit": "chunks/s", "chunks_per_gpu": n_ch, "ms": dt * 1e3,
                                      "audio_seconds_per_second": n_ch * world / dt, "dtype": "f32", "ids_checksum": int(ids.long().sum().item()),
                                      "with_cosine_cascade": {"value": n_ch * world / dt2, "unit": "chunks/s", "ms": dt2 * 1e3,
                                                              "ids_equal": bool((ids == ids2).all().item())}}
        del chunks, f13, ids, ids2
    if "dtw" in stages:
        rng = np.random.default_rng(13)
        nq, nt, L = 128, 64, 1222
        Q = [rng.standard_normal(L).astype(np.float32) for _ in range(nq)]
        T = [rng.standard_normal(L).astype(np.float32) for _ in range(nt)]
        api.dtw_distances(ctx, Q[:4], T[:4])
        _, ms = api.dtw_distances(ctx, Q, T, timing=True)
        result["dtw"] = {"metric": "DTW matcher, pairs/s (1222-element flattened MFCC sequences)", "value": nq * nt / ms * 1e3,
                         "unit": "pairs/s", "kernel_ms": ms, "cell_updates_per_s": nq * nt * L * L / ms * 1e3, "dtype": "f32",
                         "roofline": {"bound": "valu", "achieved": 4.0 * nq * nt * L * L / ms / 1e9, "peak": MFMA_F32_PEAK_TF / 2, "unit": "Tops/s",
                                      "frac": 4.0 * nq * nt * L * L / ms / 1e9 / (MFMA_F32_PEAK_TF / 2), "traffic": None, "kernel": "dtw_kernel",
                                      "kernel_ms": ms, "ops_per_cell": "4 (|x - y|, two mins, one add): not FMA work, so the peak is half the fp32 vector FLOP rate"}}

    if "plp" in stages:
        # PLP features (sidekit plp): Bark front end through the MFCC pass (the wave-stream kernel's dense-band instance) + RASTA /
        # Levinson / cepstrum back end, on the resident audio
        n_p = n_utt
        pplan = api.MfccPlan(ctx, pkg.preset_sidekit_plp(fs=fs))
        pseg = api.Segments.from_lengths(ctx, np.full(n_p, n_samp, dtype=np.int64))
        pfs = pplan.frame_segments(pseg)
        logspec = torch.empty((pfs.total, pplan.d_out), dtype=torch.float32, device=device)
        sl = flat[:n_p * n_samp]
        pplan.run(sl, pseg, pfs, out=logspec)
        api.plp_post(ctx, logspec, pfs, fs / 2.0)
        ms_f = float(np.median([pplan.run(sl, pseg, pfs, out=logspec, timing=True)[1] for _ in range(3)]))
        ms_b = float(np.median([api.plp_post(ctx, logspec, pfs, fs / 2.0, timing=True)[1] for _ in range(3)]))
        p_bytes = n_p * n_samp * 4 + pfs.total * pplan.d_out * 4
        result["plp"] = {"metric": "PLP frames/s (13-d, RASTA; Bark front end + LPC-cepstrum back end)", "value": pfs.total / (ms_f + ms_b) * 1e3,
                         "unit": "frames/s", "front_ms": ms_f, "back_ms": ms_b, "utterances": n_p, "dtype": "f32",
                         "front_roofline": {"bound": "hbm", "achieved": p_bytes / (ms_f * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": p_bytes / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": "mfcc_stream512_kernel<dense bands>",
                                            "kernel_ms": ms_f, "bytes_per_frame": tables.cfg.hop * 4 + pplan.d_out * 4}}
        del logspec

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_mfcc(20000, n_samp, fs)
        if "mfcc_inrepo" in result:
            result["mfcc_inrepo"]["cpu_baseline_reference_loop"] = cpu_baseline_mfcc_loop(n_samp, fs)
        try:  # SURVEY.md 8(d)(ii) best-effort CPU figure: the same oracle in min(usable cores, 128) worker processes, 1 BLAS thread each
            # (a child process: no fork from this GPU process; the tool states host / usable cores, the cgroup quota and the 16-worker figure)
            import subprocess
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_parallel_baseline.py"), "auto", "8"],
                                 capture_output=True, text=True, timeout=180)
            result["cpu_baseline_parallel"] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:  # pragma: no cover
            result["cpu_baseline_parallel"] = {"error": repr(e)}
        if "gmm" in result:
            result["gmm"]["cpu_baseline"] = cpu_baseline_gmm(plan.d_out, 64, 51)
            result["gmm"]["cpu_baseline_reference_loop"] = cpu_baseline_gmm_loop(plan.d_out, 64, 50)
        if "cosine" in result:
            result["cosine"]["cpu_baseline"] = cpu_baseline_cosine(256, 1251)
        if "gmm_em" in result:
            result["gmm_em"]["cpu_baseline"] = cpu_baseline_em(plan.d_out, 64)
        if "dvector_dnn" in result:
            result["dvector_dnn"]["cpu_baseline"] = cpu_baseline_dnn([1274, 256, 256, 256, 256])
        if "dtw" in result:
            result["dtw"]["cpu_baseline"] = cpu_baseline_dtw(1222)
        if "plp" in result:
            result["plp"]["cpu_baseline"] = cpu_baseline_plp(n_samp, fs)
    elif rank == 0:
        # the key stays in the line: the CPU baseline is a rank-0, one-GPU measurement (its worker processes would compete with the
        # other ranks for the host's cores)
        result["cpu_baseline"] = {"skipped": "world > 1" if world > 1 else "--no-cpu-baseline"}
    try:  # how the shipped libsspgpu.so came to be (speech_signal_processing_amd/build.py records it)
        result["build"] = {k: v for k, v in json.load(open(os.path.join(ROOT, "speech_signal_processing_amd", "build_info.json"))).items()
                           if k in ("build_mode", "compiled_sources", "lib_bytes")}
    except Exception:
        result["build"] = None
    if rank == 0:
        # the full result goes to the detail file; stdout carries exactly ONE line, the compact one (compact_line)
        detail = write_detail(result, args.detail)
        sys.stdout.flush()
        if args.full_line:
            print(json.dumps(clean_json(result), allow_nan=False), flush=True)
        else:
            print(compact_line(result, os.path.relpath(detail, ROOT) if detail and detail.startswith(ROOT) else detail), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
