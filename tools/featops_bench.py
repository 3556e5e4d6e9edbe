"""Stand-alone delta / CMVN kernels on a 100k-utterance feature matrix (device resident)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
n_utt, T = 100000, 298
for dim in (13, 26, 39):
    x = torch.randn((n_utt * T, dim), device='cuda')
    fseg = api.Segments.from_lengths(ctx, np.full(n_utt, T, dtype=np.int64))
    api.delta_features(ctx, x, fseg, 2); api.cmvn_features(ctx, x, fseg)
    md = min(api.delta_features(ctx, x, fseg, 2, timing=True)[1] for _ in range(3))
    mc = min(api.cmvn_features(ctx, x, fseg, timing=True)[1] for _ in range(3))
    gb = x.numel() * 8 / 1e9
    print("dim %d: delta %.2f ms (%.0f GB/s)   cmvn %.2f ms (%.0f GB/s of 1 read + 1 write)" % (dim, md, gb / md * 1e3, mc, gb / mc * 1e3))
    del x
