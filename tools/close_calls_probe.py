"""Calibration of bench.py's close-call rows (GPU box): which synthetic separations make ~1 / 10 / 50 % of the rows escalate in the
split-precision scorers.   python tools/close_calls_probe.py  -> prints one line per separation"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
dev = torch.device("cuda", 0)
ctx = api.Context.for_torch(0)
N, S, d = 1000000, 1251, 256
gen = torch.Generator(device=dev); gen.manual_seed(11)
Cn = torch.randn((S, d), generator=gen, device=dev)
lab = torch.randint(0, S, (N,), generator=gen, device=dev)
Z = torch.randn((N, d), generator=gen, device=dev)
for noise in (0.7, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0):
    X = Cn[lab] + noise * Z
    r0 = api.cosine_identify(ctx, X, Cn, timing=True)
    r1 = api.cosine_identify(ctx, X, Cn, timing=True, precision=1)
    r2 = api.cosine_identify(ctx, X, Cn, timing=True, precision=2)
    print("cosine noise %.1f: fp32 %.2f ms | bf16x3 %.2f ms rescored %d eq %s | cascade %.2f ms to_bf16x3 %d rescored %d eq %s | acc %.3f" % (
        noise, r0["kernel_ms"], r1["kernel_ms"], r1["rescored"], bool((r1["argmin"] == r0["argmin"]).all()), r2["kernel_ms"], r2["split_rows"], r2["rescored"],
        bool((r2["argmin"] == r0["argmin"]).all()), float((r0["argmin"].long() == lab).float().mean())), flush=True)
del X, Z
# GMM: configs[2] shape on synthetic 39-d features
n_utt, T, D, K, Sg = 100000, 298, 39, 64, 50
rng = np.random.default_rng(7)
feats = torch.randn((n_utt * T, D), generator=gen, device=dev)
seg = api.Segments.from_lengths(ctx, np.full(n_utt, T, dtype=np.int64))
wts = rng.dirichlet(5 * np.ones(K)); mu = rng.standard_normal((K, D)); cov = rng.uniform(0.5, 2.0, (K, D))
for off in (0.3, 0.1, 0.05, 0.03, 0.02, 0.01, 0.005):
    mus = np.stack([mu] + [mu + off * rng.standard_normal((K, D)) for _ in range(Sg)])
    sc = api.GmmScorer(ctx, np.stack([wts] * (Sg + 1)), mus, np.stack([cov] * (Sg + 1)), has_ubm=True)
    r0 = sc.score(feats, seg, precision=0, timing=True)
    r3 = sc.score(feats, seg, precision=3, timing=True); n3 = sc.last_rescored
    r1 = sc.score(feats, seg, precision=1, timing=True); n1 = sc.last_rescored
    print("gmm offset %.3f: fp32 %.1f ms | heuristic %.1f ms rescored %d mism %d | proven %.1f ms rescored %d mism %d" % (
        off, r0["kernel_ms"], r3["kernel_ms"], n3, int((r3["argmax"] != r0["argmax"]).sum()), r1["kernel_ms"], n1, int((r1["argmax"] != r0["argmax"]).sum())), flush=True)
    del sc
