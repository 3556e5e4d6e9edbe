"""GPU box: what bounds the host-fed MFCC path (bench.py stage `hostfed`)?  (a) pinned H2D / D2H rates alone and both directions at once
(two streams), (b) the sliced pipeline's wall time over slice sizes, float32 and int16 input.
    python tools/hostfed_probe.py [utterances]"""
import os, sys, time, json, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mb, n_h):
    os.environ["SSP_HOST_SLICE_MB"] = str(mb)
    import torch
    import speech_signal_processing_amd as pkg
    from speech_signal_processing_amd import api
    dev = torch.device("cuda", 0)
    ctx = api.Context.for_torch(0)
    n_samp = 48000
    plan = api.MfccPlan(ctx, pkg.preset_sidekit(delta_order=2))
    seg = api.Segments.from_lengths(ctx, np.full(n_h, n_samp, dtype=np.int64))
    fseg = plan.frame_segments(seg)
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(n_h * n_samp, generator=g, device=dev) * 0.1
    pin = torch.empty(n_h * n_samp, dtype=torch.float32, pin_memory=True); pin.copy_(x)
    pin16 = torch.empty(n_h * n_samp, dtype=torch.int16, pin_memory=True); pin16.copy_((x * 20000).to(torch.int16))
    out = torch.empty((fseg.total, plan.d_out), dtype=torch.float32, pin_memory=True)
    res = {"slice_mb": mb}
    if mb == 64:
        d_in = torch.empty_like(x); d_out = torch.empty((fseg.total, plan.d_out), device=dev)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        def t(fn, reps=3):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return float(np.median(ts))
        a = t(lambda: d_in.copy_(pin, non_blocking=True)); b = t(lambda: out.copy_(d_out, non_blocking=True))
        def both():
            with torch.cuda.stream(s1): d_in.copy_(pin, non_blocking=True)
            with torch.cuda.stream(s2): out.copy_(d_out, non_blocking=True)
        c = t(both)
        res["h2d_gbs"] = pin.numel() * 4 / a / 1e9; res["d2h_gbs"] = out.numel() * 4 / b / 1e9
        res["both_ms"] = c * 1e3; res["h2d_ms"] = a * 1e3; res["d2h_ms"] = b * 1e3
        del d_in, d_out
    for name, src in (("f32", pin.numpy()), ("i16", pin16.numpy())):
        plan.run(src, seg, fseg, out=out.numpy())
        if os.environ.get("SSP_HOST_TRACE_ONCE"):
            os.environ["SSP_HOST_TRACE"] = "1"
            sys.stderr.write("---- %s\n" % name)
            plan.run(src, seg, fseg, out=out.numpy())
            del os.environ["SSP_HOST_TRACE"]
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); plan.run(src, seg, fseg, out=out.numpy()); ts.append(time.perf_counter() - t0)
        res[name + "_ms"] = float(np.median(ts)) * 1e3
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        n_h = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
        for mb in [int(v) for v in os.environ.get("PROBE_MB", "8,16,32,64,128,256,1024").split(",")]:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(mb), str(n_h)], capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:], flush=True)
            if os.environ.get("SSP_HOST_TRACE_ONCE"):
                print(out.stderr, flush=True)
