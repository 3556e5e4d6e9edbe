"""Randomised LARGE batches through the scorers, on device arrays: GMM-UBM scoring (fp32 path against a float64 torch evaluation of
GMM_UBM.py:181-197 / sklearn's score on a sample of utterances; the split-precision modes 1 and 3 against the fp32 path's arg-max on
EVERY utterance, with speaker models anywhere from well apart to a thousandth of a standard deviation from the UBM) and cosine
identification (d_vector.py:315-319: fp32 distances against float64 on a sample of rows; split-precision modes 1 and 2 against the fp32
arg-min on every row, centroids in near-duplicate pairs).  tools/fuzz_scoring.py walks the shape space with a few hundred frames per
case; this one reaches what only a large batch does: device-side candidate lists of thousands of close calls, the re-scoring launches,
ragged utterances (empty ones among them) over many workgroups.  Run on the GPU box:   python tools/fuzz_scoring_batch.py [seed] [cases]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api

ctx = api.default_context()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(seed)
tg = torch.Generator(device="cuda").manual_seed(seed)
dev = "cuda"
t_start = time.time()
worst_sc = worst_cos = 0.0
LOG2PI = float(np.log(2.0 * np.pi))


def gmm_scores_f64(w, mu, cov, X, offs, pick):
    """mean_t logsumexp_k lp (sk:mixture/_gaussian_mixture.py:413-512, _base.py:337-373) in float64 on the device, utterances `pick`."""
    out = np.zeros((len(pick), mu.shape[0]))
    for i, u in enumerate(pick):
        x = X[offs[u]: offs[u + 1]].double()                                  # (T, D)
        for m in range(mu.shape[0]):
            prec = 1.0 / cov[m]                                                # (K, D)
            lp = -0.5 * (mu.shape[2] * LOG2PI + (mu[m] ** 2 * prec).sum(1) - 2.0 * x @ (mu[m] * prec).T + (x * x) @ prec.T) \
                + 0.5 * torch.log(prec).sum(1) + torch.log(w[m])
            out[i, m] = float(torch.logsumexp(lp, dim=1).mean())
    return out


for case in range(n_cases):
    # ---- GMM-UBM
    K, D = int(rng.choice([8, 16, 32, 64, 128])), int(rng.choice([13, 26, 39, 64]))
    S = int(rng.integers(2, 80))
    M = S + 1
    off_scale = 10.0 ** rng.uniform(-3.0, 0.3)
    w1 = rng.dirichlet(5 * np.ones(K))
    mu0 = 2.0 * rng.standard_normal((K, D))
    cv1 = rng.uniform(0.5, 2.0, (K, D))
    mu = np.concatenate([mu0[None], mu0[None] + off_scale * np.sqrt(cv1)[None] * rng.standard_normal((S, K, D))])
    if rng.random() < 0.3:   # independent weights / variances per model (the reference trains independent GMMs, GMM_UBM.py:158-160)
        w = rng.dirichlet(5 * np.ones(K), M)
        cv = rng.uniform(0.5, 2.0, (M, K, D))
    else:
        w, cv = np.repeat(w1[None], M, 0), np.repeat(cv1[None], M, 0)
    n_utt = int(rng.integers(500, 20000))
    kind = rng.random(n_utt)
    lens = np.where(kind < 0.02, 0, np.where(kind < 0.12, rng.integers(1, 10, n_utt), rng.integers(20, 400, n_utt)))
    offs = np.concatenate([[0], np.cumsum(lens)])
    F = int(offs[-1])
    spk = torch.as_tensor(rng.integers(0, S, n_utt), device=dev)
    uid = torch.repeat_interleave(torch.arange(n_utt, device=dev), torch.as_tensor(lens, device=dev))
    comp = torch.multinomial(torch.as_tensor(w1, device=dev).float(), F, replacement=True, generator=tg) if F else torch.zeros(0, dtype=torch.long, device=dev)
    mu_t, cv_t, w_t = (torch.as_tensor(a, device=dev) for a in (mu, cv, w))
    X = (mu_t[spk[uid] + 1, comp] + torch.sqrt(cv_t[spk[uid] + 1, comp]) * torch.randn(F, D, device=dev, generator=tg, dtype=torch.float64)).float()
    seg = api.Segments.from_lengths(ctx, [int(v) for v in lens])
    sc = api.GmmScorer(ctx, w, mu, cv, has_ubm=True)
    r0 = sc.score(X, seg, precision=0)
    a0, s0 = r0["argmax"], r0["scores"]
    nz = torch.as_tensor(lens > 0, device=dev)
    listed = {}
    for prec in (1, 3, 4):   # (4 = auto: a pilot picks 1 or 0)
        a1 = sc.score(X, seg, precision=prec)["argmax"]
        bad = torch.nonzero((a0 != a1) & nz).flatten()
        assert bad.numel() == 0, (case, "gmm split-precision arg-max differs from fp32's", prec, K, D, S, off_scale, "utterances",
                                  bad[:8].tolist(), "lens", [int(lens[b]) for b in bad[:8].tolist()])
        listed[prec] = getattr(sc, "last_rescored", None) if prec != 4 else "auto->%d" % sc.last_auto["precision_used"]
    pick = [int(u) for u in rng.choice(np.nonzero(lens > 0)[0], 24, replace=False)]
    ref = gmm_scores_f64(w_t, mu_t, cv_t, X, offs, pick)
    got = s0[torch.as_tensor(pick, device=dev)].cpu().numpy().astype(np.float64)
    e = float(np.max(np.abs(got - ref) / np.abs(ref)))
    worst_sc = max(worst_sc, e)
    assert e <= 1e-4, (case, "gmm scores vs float64", K, D, S, e)
    d = ref[:, 1:] - ref[:, :1]
    srt = np.sort(d, axis=1)
    clear = srt[:, -1] - srt[:, -2] > 2e-4 * np.abs(ref).max(axis=1)
    am = a0[torch.as_tensor(pick, device=dev)].cpu().numpy()
    assert (am[clear] == d.argmax(1)[clear]).all(), (case, "gmm arg-max vs float64", K, D, S)
    # ---- cosine
    N, Sc, dd = int(rng.integers(10000, 400000)), int(rng.integers(2, 2000)), int(rng.choice([64, 128, 200, 256]))
    Cn = torch.randn(Sc, dd, device=dev, generator=tg)
    eps = 10.0 ** rng.uniform(-7, -1)
    Cn[1::2] = Cn[0::2][: Cn[1::2].shape[0]] * (1.0 + eps * torch.randn(Cn[1::2].shape, device=dev, generator=tg))   # near-duplicate pairs
    lab = torch.randint(0, Sc, (N,), device=dev, generator=tg)
    Xc = Cn[lab] + float(10.0 ** rng.uniform(-2, 1)) * torch.randn(N, dd, device=dev, generator=tg)
    c0 = api.cosine_identify(ctx, Xc, Cn)
    for prec in (1, 2, 3):   # (3 = auto: a pilot picks 2, 1 or 0)
        c1 = api.cosine_identify(ctx, Xc, Cn, precision=prec)
        bad = torch.nonzero(c0["argmin"] != c1["argmin"]).flatten()
        assert bad.numel() == 0, (case, "cosine split-precision arg-min differs from fp32's", prec, N, Sc, dd, eps, bad[:8].tolist())
    rows = torch.as_tensor(rng.choice(N, 512, replace=False), device=dev)
    xs, cs = Xc[rows].double(), Cn.double()
    refd = (1.0 - (xs @ cs.T) / (xs.norm(dim=1, keepdim=True) * cs.norm(dim=1)[None])).clamp(0.0, 2.0)
    gotd = api.cosine_identify(ctx, Xc[rows].contiguous(), Cn, dist=True)["dist"].double()
    e = float((gotd - refd).abs().max())
    worst_cos = max(worst_cos, e)
    assert e < 2e-5, (case, "cosine distances vs float64", N, Sc, dd, e)
    print("case %d: gmm K %d D %d S %d offset %.1e, %d utterances / %d frames (listed: %s); cosine %d x %d x %d pairs at %.0e" %
          (case, K, D, S, off_scale, n_utt, F, listed, N, Sc, dd, eps), flush=True)
print("fuzz_scoring_batch OK: %d cases, worst score error %.2e (relative), worst cosine distance error %.2e, %.1f s" %
      (n_cases, worst_sc, worst_cos, time.time() - t_start))
