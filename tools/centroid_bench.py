import sys, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api, _lib
import ctypes as C
ctx = api.Context.for_torch(0)
for N, S, d in ((1000000, 1251, 256), (100000, 1251, 256), (20000, 50, 256)):
    X = torch.randn((N, d), device='cuda')
    lab = torch.randint(0, S, (N,), device='cuda', dtype=torch.int32)
    import time
    api.centroids(ctx, X, lab, S); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = api.centroids(ctx, X, lab, S); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("centroids N=%d S=%d d=%d: %.2f ms wall" % (N, S, d, dt * 1e3))
