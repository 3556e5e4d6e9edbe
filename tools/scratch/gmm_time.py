"""main-kernel time of the bf16x3 GMM scorer (precision 2: no re-scoring) on configs[2]-shaped random data"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
U, T, D, K, S = 100000, 298, 39, 64, 50
g = torch.Generator(device='cuda'); g.manual_seed(3)
feats = torch.randn((U * T, D), generator=g, device='cuda')
rng = np.random.default_rng(7)
w = rng.dirichlet(5 * np.ones(K), size=S + 1)
mu = rng.standard_normal((S + 1, K, D)) * 0.5
cv = rng.uniform(0.5, 1.5, size=(S + 1, K, D))
sc = api.GmmScorer(ctx, w, mu, cv, has_ubm=True)
seg = api.Segments.from_lengths(ctx, np.full(U, T, dtype=np.int64))
for prec in (2, 0):
    sc.score(feats, seg, precision=prec)
    ms = [sc.score(feats, seg, precision=prec, timing=True)["kernel_ms"] for _ in range(3)]
    print("precision", prec, "kernel_ms", ["%.2f" % m for m in ms])
