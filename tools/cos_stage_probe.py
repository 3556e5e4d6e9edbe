"""GPU box, under `rocprofv3 --kernel-trace --stats`: one shape of the cosine scorer at every precision — which kernel of a split-precision call
takes the time when rows are listed?    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/cos_stage_probe.py N S d noise"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_signal_processing_amd import api
N, S, d, noise = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
ctx = api.Context.for_torch(0)
g = torch.Generator(device="cuda").manual_seed(1)
Cn = torch.randn((S, d), generator=g, device="cuda")
lab = torch.randint(0, S, (N,), generator=g, device="cuda")
X = Cn[lab] + noise * torch.randn((N, d), generator=g, device="cuda")
for p in (0, 1, 2, 3):
    for _ in range(3):
        r = api.cosine_identify(ctx, X, Cn, precision=p, timing=True)
    print("precision", p, "kernel_ms %.3f" % r["kernel_ms"], {k: v for k, v in r.items() if k in ("rescored", "split_rows", "auto")}, flush=True)
