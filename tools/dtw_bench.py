"""All-pairs DTW throughput: n_q x n_t pairs of flattened-MFCC-like sequences (94 frames x 13 = 1222 scalars, MFCC_DTW.py:54)."""
import sys, numpy as np
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
ctx = api.default_context()
rng = np.random.default_rng(0)
nq, nt, L = 256, 128, 1222
Q = [rng.standard_normal(L).astype(np.float32) for _ in range(nq)]
T = [rng.standard_normal(L).astype(np.float32) for _ in range(nt)]
api.dtw_distances(ctx, Q[:8], T[:8])
d, ms = api.dtw_distances(ctx, Q, T, timing=True)
cells = nq * nt * L * L
print("%d x %d pairs of %d: %.2f ms -> %.3g pairs/s, %.3g cell updates/s" % (nq, nt, L, ms, nq * nt / ms * 1e3, cells / ms * 1e3))
