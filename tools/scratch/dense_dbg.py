import sys, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
for (N, d, u) in [(1000, 1274, 256), (128, 64, 128), (128, 1274, 128), (130, 300, 256), (1000, 1274, 100)]:
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    X = torch.randn((N, d), generator=g, device='cuda'); W = torch.randn((u, d), generator=g, device='cuda') / d ** 0.5
    b = torch.randn(u, generator=g, device='cuda')
    import os
    os.environ["SSP_DENSE_NO_REG"] = "1"
    Y = api.dense_forward(ctx, X, W, b, relu=False)
    ref = (X.double() @ W.double().T + b.double()).float()
    err = (Y - ref).abs()
    bad = (err > 1e-3).nonzero()
    print(N, d, u, "max err", err.max().item(), "n bad", len(bad), "first bad", bad[:5].tolist(), "rows bad", sorted(set((bad[:, 0]).tolist()))[:10], "cols bad", sorted(set((bad[:, 1]).tolist()))[:10])
