"""d-vector network forward throughput (1274 -> 256 x 4, d_vector.py:171-189) on N precomputed 1-s MFCC chunks: the packed network
(ssp_dnn: input layer GEMM + hidden / output layers chained in registers) against one ssp_dense_forward launch per layer."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
g = torch.Generator(device='cuda'); g.manual_seed(1)
X = torch.randn((N, 1274), generator=g, device='cuda')
dims = [1274, 256, 256, 256, 256]
Ws = [torch.randn((dims[i + 1], dims[i]), generator=g, device='cuda') / dims[i] ** 0.5 for i in range(4)]
bs = [0.1 * torch.randn(dims[i + 1], generator=g, device='cuda') for i in range(4)]
net = api.DnnForward(ctx, [(Ws[i].cpu().numpy(), bs[i].cpu().numpy(), i < 3) for i in range(4)])
for rep in range(2):
    h, tot = X, 0.0
    for i in range(4):
        h, ms = api.dense_forward(ctx, h, Ws[i], bs[i], relu=i < 3, timing=True)
        tot += ms
        if rep: print("layer %d: %.3f ms  %.1f TFLOP/s" % (i, ms, 2.0 * N * dims[i] * dims[i + 1] / ms / 1e9))
flop = 2.0 * N * sum(dims[i] * dims[i + 1] for i in range(4))
print("N=%d per-layer launches: %.2f ms total -> %.3g embeddings/s, %.1f TFLOP/s (fp32 MFMA peak 157.3)" % (N, tot, N / tot * 1e3, flop / tot / 1e9))
for rep in range(3):
    y, ms = net.forward(X, timing=True)
print("N=%d packed network:     %.2f ms total -> %.3g embeddings/s, %.1f TFLOP/s = %.3f of peak; max |diff| vs per-layer %.2e" % (
    N, ms, N / ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3, (y - h).abs().max().item()))
