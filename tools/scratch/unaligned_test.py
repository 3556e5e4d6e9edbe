import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
os.environ["SSP_STREAM_UNALIGNED"] = "1"
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api
from test_gpu_parity import _run_plan, synth_audio
import torch
for order in (0, 2):
    tables = pkg.preset_sidekit(delta_order=order)
    sigs = [synth_audio(u, 16000 + 37 * u + (u % 4), 16000) for u in range(12)] + [synth_audio(50, 48001, 16000), synth_audio(51, 401, 16000), synth_audio(52, 90003, 16000)]
    g3, _ = _run_plan(api, tables, sigs, variant=3)
    g2, _ = _run_plan(api, tables, sigs, variant=2)
    worst = max(float(np.abs(a - b).max()) for a, b in zip(g3, g2))
    print("host order", order, "max |stream - workgroup| =", worst)
    g3d, _ = _run_plan(api, tables, sigs, variant=3, device=True)
    print("device identical to host:", all(np.array_equal(a, b) for a, b in zip(g3, g3d)))
    # device pointer itself misaligned by 1..3 floats
    ctx = api.default_context(torch_stream=True)
    plan = api.MfccPlan(ctx, tables)
    for mis in (1, 2, 3):
        flat = np.concatenate(sigs).astype(np.float32)
        buf = torch.zeros(len(flat) + 8, device='cuda')
        buf[mis:mis + len(flat)] = torch.from_numpy(flat).cuda()
        seg = api.Segments.from_lengths(ctx, [len(s) for s in sigs]); fseg = plan.frame_segments(seg)
        out = plan.run(buf[mis:mis + len(flat)], seg, fseg, variant=3).cpu().numpy()
        ref = np.concatenate(g3)
        print(" base misaligned by", mis, "floats: identical", np.array_equal(out, ref), float(np.abs(out - ref).max()))
