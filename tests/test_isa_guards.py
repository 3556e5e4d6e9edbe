"""Guards on the generated gfx950 ISA (hipcc cross-compiles here, no GPU): the instruction classes that round 3 found hiding in hot loops —
a library expansion nobody asked for (an if-converted `sqrtf`, `powf(expf())`, the division expansion) or a spill reload in a frame
loop — must not come back unnoticed.  Each check compiles ONE small source to assembly (seconds)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "speech_signal_processing_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _isa(src, tmp_path, extra=(), with_depth=False):
    """kernel name -> instruction lines (with_depth: (loop depth from the compiler's block comments, line) pairs)"""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path / (src + ".s"))
    sys.path.insert(0, ROOT)
    from speech_signal_processing_amd.build import SOURCE_FLAGS  # (the per-source flags of the shipped build: the guards read what ships)
    extra = tuple(SOURCE_FLAGS.get(src, [])) + tuple(extra)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-munsafe-fp-atomics", "-Wno-pass-failed", *extra,
                        "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, cur, depth = {}, None, 0
    for line in open(out):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            depth = 0
            continue
        if line.startswith(".LBB"):
            d = re.search(r"Depth=(\d+)", line)
            depth = int(d.group(1)) if d else 0
            continue
        t = line.strip()
        if cur is None or not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        if op == "s_endpgm":
            cur = None
            continue
        kernels[cur].append((depth, t) if with_depth else t)
    return kernels


def _count(instrs, pattern):
    return sum(1 for t in instrs if re.match(pattern, t.split()[0]))


def test_stream2048_power_dialects_carry_no_square_root(tmp_path):
    """mfcc_stream2048_kernel<POWER = 2, *>: no v_sqrt (the magnitude branch was once if-converted into the power dialects: 290 of 845
    vector instructions per frame); the magnitude instances use the hardware instruction without libm's fix-up sequence; and the frame
    loop of every instance is free of scratch traffic (at most the one spill outside it)."""
    k = _isa("mfcc_stream2k.hip", tmp_path)
    inst = {n: v for n, v in k.items() if "mfcc_stream2048_kernel" in n}
    assert len(inst) == 6, list(inst)
    for n, v in inst.items():
        power = int(re.search(r"kernelILi(\d)ELi", n).group(1))
        n_sqrt = _count(v, r"v_sqrt_f32")
        if power == 2:
            assert n_sqrt == 0, (n, n_sqrt)
        else:
            assert 0 < n_sqrt <= 20 and _count(v, r"v_cmp_class_f32") == 0, (n, n_sqrt)   # 17 bins per lane, no denormal rescue
        assert _count(v, r"scratch_load") <= 2, (n, _count(v, r"scratch_load"))
        assert _count(v, r"v_div_(scale|fmas|fixup)") == 0, n


def _lane_mask_loops_behind(instrs, marker):
    """loop tests on lane masks (`s_andn2_b64 exec, exec, s[..]`: lanes leave the loop separately) behind the first `marker` instruction"""
    seen, n = False, 0
    for t in instrs:
        seen = seen or t.startswith(marker)
        if seen and re.match(r"s_andn2_b64 exec, exec", t):
            n += 1
    return n


def test_persistent_loops_stay_wave_uniform(tmp_path):
    """Round 4's hang of mfcc_stream2048_kernel (a ragged batch), root-caused in round 5: a lane-0 block at the BOTTOM of a persistent
    claim loop and the lane-0 block of the claim at its TOP are neighbours across the back edge; the compiler threads the other lanes
    around both and rebuilds the loop as one over lane masks in which lanes leave separately — with lane 0 gone the claim is skipped,
    readfirstlane returns the initial 0 and the wave walks item 0 for ever.  tools/microbench/lane0_loop.hip is the reduction: its
    FORM 3 / 4 must still show the loop test on lane masks (the detector means something with this compiler), FORM 1 — the cure, stores
    from all lanes with all but one out of range — must not; and no persistent-loop kernel of the library may show one behind its claim
    (the 2048-point kernel has two honest divergent loops in FRONT of its barrier: table copies)."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    red = os.path.join(ROOT, "tools", "microbench", "lane0_loop.hip")
    counts = {}
    for form in (1, 3, 4):
        out = str(tmp_path / ("lane0_%d.s" % form))
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-DFORM=%d" % form, "-o", out, red], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        counts[form] = sum(1 for line in open(out) if re.match(r"\s*s_andn2_b64 exec, exec", line))
    assert counts[1] == 0 and counts[3] >= 1 and counts[4] >= 1, counts
    k = _isa("mfcc_stream2k.hip", tmp_path)
    inst = {n: v for n, v in k.items() if "mfcc_stream2048_kernel" in n}
    assert len(inst) >= 4
    for n, v in inst.items():
        assert sum(1 for t in v if t.startswith("s_barrier")) == 1, n
        assert _lane_mask_loops_behind(v, "s_barrier") == 0, n
        assert sum(1 for t in v if t.startswith("buffer_atomic_")) == 3, n  # the chunk maximum: three bounds-checked atomics, no lane-0 block


def test_plp_cepstrum_kernel_is_lean(tmp_path):
    """plp_cep_fixed_kernel<21, 12>: a frame is ~800 instructions — one v_exp per band, twelve reciprocals, one logarithm; no division
    expansion, no ldexp / frexp (powf), no LDS reads for the tables (they are scalar operands)."""
    k = _isa("plp.hip", tmp_path)
    name = [n for n in k if "plp_cep_fixed_kernelILi21ELi12E" in n]
    assert len(name) == 1, list(k)
    v = k[name[0]]
    assert len(v) < 1000, len(v)
    assert _count(v, r"v_div_(scale|fmas|fixup)") == 0 and _count(v, r"v_(ldexp|frexp)") == 0
    assert _count(v, r"ds_read") == 0
    assert _count(v, r"v_exp_f32") == 19 and _count(v, r"v_rcp_f32") == 12 and _count(v, r"v_log_f32") == 1


def test_dtw_cell_is_three_instructions(tmp_path):
    """dtw1_kernel<20>: the recurrence loop holds no per-cell select (the kernel's selects are the 20 of the template load's bounds and
    the 20 of the final pick of the last valid column)."""
    k = _isa("dtw.hip", tmp_path)
    name = [n for n in k if "dtw1_kernelILi20E" in n]
    assert len(name) == 1, list(k)
    v = k[name[0]]
    assert _count(v, r"v_cndmask") <= 45, _count(v, r"v_cndmask")
    assert _count(v, r"v_min3_f32") >= 20


def test_stream512_quad_loops_are_free_of_scratch(tmp_path):
    """mfcc_stream512_kernel<..., WALK = 0> (benchmark instances): a spill reload inside the quad loop waits on vmcnt behind the sample DMA —
    its whole latency, every quad (12.4 instead of 10.1 ms with 16 spilled registers once).  No instance carries scratch at all since
    round 5 (the scaling instance <..., CM = 1> spilled five registers per chunk in round 4).  NOTHING of the non-finite case sits in
    this kernel: no compare against infinity, no sticky accumulate, no flag store, no branch of the rare case — the scan kernel reads
    the pollution off the stored rows and the third kernel (WALK = 1, mfcc_stream_walk.hip) redoes flagged chunks.  And the chunk loop
    stays a plain wave-uniform loop: three formulations of a chunk verdict turned it into a loop over lane masks
    (`s_andn2_b64 exec, exec, ...` as the loop test), one of which hung on the GPU."""
    k = _isa("mfcc_stream.hip", tmp_path, extra=("-DSSP_FAST_MINIMAL",), with_depth=True)
    inst = {n: v for n, v in k.items() if "mfcc_stream512_kernel" in n}
    head = [n for n in inst if "ILi13ELi2ELi1ELi3ELi6ELi2ELi3ELi0ELi0E" in n]
    assert len(head) == 1, list(inst)
    for n, v in inst.items():
        assert "Li0EEEv" in n, n  # first kernels only in this unit
        assert not [t for _, t in v if t.startswith("scratch_")], n
        assert not [t for _, t in v if re.match(r"s_andn2_b64 exec, exec", t)], n
        assert not [t for d, t in v if d >= 2 and re.match(r"v_cmp_(nlg|class)_f32", t)], n
        assert not [t for _, t in v if t.startswith(("s_swappc", "s_setpc"))], n
        assert not [t for d, t in v if d >= 1 and re.match(r"v_fma(c)?_f32\S* v\d+, 0, ", t)], n  # (round 5's sticky verdict: gone)
    scan = [v for n, v in k.items() if "mfcc_stream_scan_kernel" in n]
    assert len(scan) == 1 and not [t for _, t in scan[0] if t.startswith("scratch_")]


def test_stream512_walk_kernels(tmp_path):
    """the third kernel of a wave-stream launch (mfcc_stream_walk.hip): a plain wave-uniform loop nest as well (no loop over lane masks),
    no scratch, the legacy product on the window rows, and an early exit on the "any chunk flagged" word in front of everything else"""
    k = _isa("mfcc_stream_walk.hip", tmp_path, extra=("-DSSP_FAST_MINIMAL",), with_depth=True)
    inst = {n: v for n, v in k.items() if "mfcc_stream512_kernel" in n}
    assert inst and all("Li1EEEv" in n for n in inst), list(inst)
    for n, v in inst.items():
        assert not [t for _, t in v if t.startswith("scratch_")], n
        assert not [t for _, t in v if re.match(r"s_andn2_b64 exec, exec", t)], n
        assert sum(1 for _, t in v if t.startswith("v_mul_legacy_f32")) >= 26, n
        first = [t.split()[0] for _, t in v[:12]]
        assert "s_endpgm" in first or any(t.startswith("s_cbranch") for t in first), (n, first)


def test_cosine_split_precision_tile_loop_is_free_of_scratch(tmp_path):
    """cosine_bf16x3_kernel<16, 3> sits exactly on the 168-register line of three waves per SIMD: what it spills must stay in the prologue /
    epilogue (loop depth 0), the tile loop holds 48 matrix instructions per tile (16 k-steps x 3 products) and no scratch access."""
    k = _isa("cosine.hip", tmp_path, with_depth=True)
    name = [n for n in k if "cosine_bf16x3_kernelILi16ELi3E" in n]
    assert len(name) == 1, list(k)
    v = k[name[0]]
    assert not [t for d, t in v if d >= 1 and t.startswith("scratch_")]
    assert sum(1 for d, t in v if d >= 1 and t.startswith("v_mfma_f32_32x32x16_bf16")) == 48
