"""Randomised ragged batches through the HOST-FED paths of the MFCC pass (round 6) against the device-pointer path on the same samples:
the sliced copy-in / compute / copy-back pipeline of ssp_mfcc_run(SSP_HOST) with random slice sizes (SSP_HOST_SLICE_MB is read per
call), float32 and int16 input (ssp_mfcc_run_i16: device-side widening), pinned and pageable memory, every dialect and every kernel
variant the plan has.  The sliced path launches the batch's work table in utterance ranges with data pointers biased by each slice's
first offsets: any kernel that addressed samples or rows other than through the batch's absolute offsets would differ here.  Bits must
be EQUAL (the work table is the same on both sides).  Run on the GPU box:
    python tools/fuzz_hostfed.py [seed] [cases]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
import speech_signal_processing_amd as pkg
from speech_signal_processing_amd import api

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(seed)
ctx = api.default_context()                      # the library's own stream: host pointers
tctx = api.default_context(torch_stream=True)    # torch's stream: device pointers
t_start = time.time()
n_sliced = 0
for case in range(n_cases):
    dialect = str(rng.choice(["sidekit", "sidekit", "sidekit_cmvn", "inrepo", "plp", "librosa", "sidekit13"]))
    min_len = 1
    if dialect == "sidekit":
        tables = pkg.preset_sidekit(delta_order=int(rng.integers(1, 3)))
    elif dialect == "sidekit13":
        tables = pkg.preset_sidekit(delta_order=0)
    elif dialect == "sidekit_cmvn":
        tables = pkg.preset_sidekit(delta_order=int(rng.integers(1, 3)), cmvn=1)
    elif dialect == "inrepo":
        tables = pkg.preset_inrepo(int(rng.choice([8000, 16000])), 512, int(rng.choice([160, 256])))
    elif dialect == "plp":
        tables = pkg.preset_sidekit_plp()
    else:
        tables = pkg.preset_librosa(int(rng.choice([8000, 16000])), 13)   # (two-pass top_db: the host path stages this one whole)
        min_len = 1025
    n_utt = int(rng.integers(40, 1500))
    kind = rng.random(n_utt)
    lens = np.where(kind < 0.05, rng.integers(min_len, min_len + 900, n_utt),
           np.where(kind < 0.97, rng.integers(4000, 60000, n_utt), rng.integers(200000, 1500000, n_utt)))
    if rng.random() < 0.25:
        lens[:] = int(rng.integers(8000, 60000))
    lens = [int(v) for v in lens]
    total = int(np.sum(lens))
    x16 = (rng.standard_normal(total) * float(10.0 ** rng.uniform(1.0, 3.8))).clip(-32768, 32767).astype(np.int16)
    if rng.random() < 0.3:   # a silent stretch (ln 0 = -inf: the walk kernel inside a slice)
        a = int(rng.integers(0, max(1, total - 5000)))
        x16[a:a + int(rng.integers(400, 5000))] = 0
    x32 = x16.astype(np.float32)
    slice_mb = int(rng.choice([1, 1, 2, 3, 5]))
    os.environ["SSP_HOST_SLICE_MB"] = str(slice_mb)
    sliced = total * 4 >= 2 * (slice_mb << 20) and n_utt >= 2 and tables.cfg.n_fft != 2048
    n_sliced += int(sliced)
    plan, tplan = api.MfccPlan(ctx, tables), api.MfccPlan(tctx, tables)
    seg, tseg = api.Segments.from_lengths(ctx, lens), api.Segments.from_lengths(tctx, lens)
    fseg, tfseg = plan.frame_segments(seg), tplan.frame_segments(tseg)
    xd = torch.from_numpy(x32).cuda()
    variants = [0] + ([int(rng.choice([1, 2, 3]))] if tables.cfg.n_fft == 512 and dialect != "plp" else [])
    for variant in variants:
        try:
            want = tplan.run(xd, tseg, tfseg, variant=variant).cpu().numpy()
        except NotImplementedError:
            continue   # (the plan has no instance of that kernel)
        pinned = rng.random() < 0.5
        for name, src in (("f32", x32), ("i16", x16)):
            if pinned:
                buf = api.pinned_empty(src.shape, src.dtype)
                buf[:] = src
                src = buf
            got = plan.run(src, seg, fseg, variant=variant)
            assert np.array_equal(got, want, equal_nan=True), (case, dialect, variant, name, "pinned" if pinned else "pageable", slice_mb, n_utt,
                                                               int(np.argmax((got != want) & ~(np.isnan(got) & np.isnan(want))) // got.shape[1]))
        # int16 through device pointers (widened slice by slice on the device)
        got = tplan.run(torch.from_numpy(x16).cuda(), tseg, tfseg, variant=variant).cpu().numpy()
        assert np.array_equal(got, want, equal_nan=True), (case, dialect, variant, "i16 device")
    print("case %d: %s, %d utterances / %.1f MB, slice %d MiB (%s), variants %s" % (case, dialect, n_utt, total * 4 / 1e6, slice_mb,
                                                                                  "sliced" if sliced else "whole", variants), flush=True)
print("fuzz_hostfed OK: %d cases (%d through the sliced pipeline), %.1f s" % (n_cases, n_sliced, time.time() - t_start))
