import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
for n, K, D in ((3000000, 64, 39), (3000000, 512, 39), (30000000, 64, 39)):
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    X = torch.randn((n, D), generator=g, device='cuda')
    rng = np.random.default_rng(0)
    w = rng.dirichlet(5 * np.ones(K)); mu = rng.standard_normal((K, D)); cov = rng.uniform(0.5, 2, (K, D))
    api.gmm_em_stats(ctx, w, mu, cov, X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = api.gmm_em_stats(ctx, w, mu, cov, X, timing=True)
    wall = (time.perf_counter() - t0) * 1e3
    print("n=%d K=%d D=%d  wall %.2f ms  kernel %.2f ms  -> %.3g frame-mixture updates/s, %.2f TFLOP/s (12 D flop per frame-mixture)" % (n, K, D, wall, r["kernel_ms"], n * K / r["kernel_ms"] * 1e3, 12.0 * D * n * K / r["kernel_ms"] / 1e9))
