"""GPU box: does `precision = auto` keep its promise away from the bench's shapes?  For a grid of model shapes and speaker separations,
kernel time of ssp_gmm_score at precision 0 (fp32), 1 (proven band) and 4 (auto), and of ssp_cosine_identify2 at 0 / 1 / 2 / 3; prints one
row per point with auto's time over the best fixed choice, and the worst ratio.
    python tools/auto_sweep.py [utterances]"""
import sys, time, json
import numpy as np
import torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api

ctx = api.Context.for_torch(0)
dev = torch.device("cuda", 0)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
T = 298
rng = np.random.default_rng(3)
rows, worst = [], 0.0


def t_ms(fn, reps=2):
    fn()
    ms = []
    for _ in range(reps):
        ms.append(fn()["kernel_ms"])
    return float(np.mean(ms))


ONLY = sys.argv[2] if len(sys.argv) > 2 else "both"
for K, D, S in ((16, 26, 10), (64, 39, 50), (32, 13, 200), (128, 39, 20), (512, 39, 100)) if ONLY in ("both", "gmm") else ():
    u = U if K * S <= 64 * 50 else max(2000, U * 64 * 50 // (K * S))
    g = torch.Generator(device=dev).manual_seed(K + S)
    X = torch.randn((u * T, D), generator=g, device=dev)
    seg = api.Segments.from_lengths(ctx, [T] * u)
    w = rng.dirichlet(5 * np.ones(K))
    mu = rng.standard_normal((K, D))
    cov = rng.uniform(0.5, 2.0, (K, D))
    for off in (1.0, 0.3, 0.1, 0.03, 0.01):
        mus = np.stack([mu] + [mu + off * np.sqrt(cov) * rng.standard_normal((K, D)) for _ in range(S)])
        sc = api.GmmScorer(ctx, np.broadcast_to(w, (S + 1, K)), mus, np.broadcast_to(cov, (S + 1, K, D)), has_ubm=True)
        r0 = sc.score(X, seg, precision=0)
        t0 = t_ms(lambda: sc.score(X, seg, precision=0, timing=True))
        t1 = t_ms(lambda: sc.score(X, seg, precision=1, timing=True))
        listed = sc.last_rescored
        ta = t_ms(lambda: sc.score(X, seg, precision=4, timing=True))
        ra = sc.score(X, seg, precision=4)
        info = sc.last_auto
        ratio = ta / min(t0, t1)
        worst = max(worst, ratio)
        rows.append({"scorer": "gmm", "K": K, "D": D, "S": S, "utterances": u, "offset_std": off, "fp32_ms": t0, "proven_ms": t1, "listed": listed / u,
                     "auto_ms": ta, "auto_used": info["precision_used"], "predicted": info["predicted_cost_of_precision_1"], "ratio": ratio,
                     "argmax_equal": bool((ra["argmax"] == r0["argmax"]).all().item())})
        print(json.dumps(rows[-1]), flush=True)
        del sc
    del X

for N, S, d in ((1000000, 1251, 256), (400000, 300, 128), (200000, 5000, 64), (1000000, 40, 256), (100000, 1251, 256), (30000, 100, 192)) if ONLY in ("both", "cosine") else ():
    g = torch.Generator(device=dev).manual_seed(N % 1000 + S)
    Cn = torch.randn((S, d), generator=g, device=dev)
    lab = torch.randint(0, S, (N,), generator=g, device=dev)
    Z = torch.randn((N, d), generator=g, device=dev)
    for noise in (0.7, 3.0, 6.0, 12.0, 40.0):
        X = Cn[lab] + noise * Z
        r0 = api.cosine_identify(ctx, X, Cn)
        ts = [t_ms(lambda p=p: api.cosine_identify(ctx, X, Cn, precision=p, timing=True), 3) for p in (0, 1, 2)]
        ta = t_ms(lambda: api.cosine_identify(ctx, X, Cn, precision=3, timing=True), 3)
        ra = api.cosine_identify(ctx, X, Cn, precision=3)
        ratio = ta / min(ts)
        worst = max(worst, ratio)
        r2 = api.cosine_identify(ctx, X, Cn, precision=2)
        rows.append({"scorer": "cosine", "N": N, "S": S, "d": d, "noise": noise, "fp32_ms": ts[0], "bf16x3_ms": ts[1], "cascade_ms": ts[2], "auto_ms": ta,
                     "auto_used": ra["auto"]["precision_used"], "to_x3": ra["auto"]["pilot_to_bf16x3"] / max(1, ra["auto"]["pilot_rows"]),
                     "f1": r2["split_rows"] / N, "f2": r2["rescored"] / N, "ratio": ratio,
                     "argmin_equal": bool((ra["argmin"] == r0["argmin"]).all().item())})
        print(json.dumps(rows[-1]), flush=True)
        del X
print("auto_sweep: %d points, worst auto / best fixed = %.3f, all decisions equal to fp32: %s" % (
    len(rows), worst, all(r.get("argmax_equal", True) and r.get("argmin_equal", True) for r in rows)))
