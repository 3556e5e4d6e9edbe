"""d-vector network forward throughput (1274 -> 256 x 4, d_vector.py:171-189) on N precomputed 1-s MFCC chunks."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
g = torch.Generator(device='cuda'); g.manual_seed(1)
X = torch.randn((N, 1274), generator=g, device='cuda')
dims = [1274, 256, 256, 256, 256]
Ws = [torch.randn((dims[i + 1], dims[i]), generator=g, device='cuda') / dims[i] ** 0.5 for i in range(4)]
bs = [torch.zeros(dims[i + 1], device='cuda') for i in range(4)]
for rep in range(2):
    h, tot = X, 0.0
    for i in range(4):
        h, ms = api.dense_forward(ctx, h, Ws[i], bs[i], relu=i < 3, timing=True)
        tot += ms
        if rep: print("layer %d: %.3f ms  %.1f TFLOP/s" % (i, ms, 2.0 * N * dims[i] * dims[i + 1] / ms / 1e9))
flop = 2.0 * N * sum(dims[i] * dims[i + 1] for i in range(4))
print("N=%d: %.2f ms total -> %.3g embeddings/s, %.1f TFLOP/s (fp32 MFMA peak 157.3)" % (N, tot, N / tot * 1e3, flop / tot / 1e9))
