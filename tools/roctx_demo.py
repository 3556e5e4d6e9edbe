"""roctx ranges around the C-ABI calls (SSP_ROCTX=1): run under `rocprofv3 --marker-trace --kernel-trace --stats`."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
ctx = api.Context.for_torch(0)
U, n = 2000, 48000
x = (0.1 * torch.randn(U * n, device='cuda')).contiguous()
seg = api.Segments.from_lengths(ctx, np.full(U, n, dtype=np.int64))
plan = api.MfccPlan(ctx, pkg.preset_sidekit(delta_order=2))
fseg = plan.frame_segments(seg)
feats = plan.run(x, seg, fseg)
rng = np.random.default_rng(0)
K, S, D = 64, 10, 39
sc = api.GmmScorer(ctx, rng.dirichlet(5 * np.ones(K), size=S + 1), rng.standard_normal((S + 1, K, D)), rng.uniform(0.5, 1.5, (S + 1, K, D)))
for _ in range(3):
    feats = plan.run(x, seg, fseg)
    sc.score(feats, fseg, precision=1)
torch.cuda.synchronize()
print("done")
