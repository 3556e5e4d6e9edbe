"""CPU-side tests: host tables vs the oracle and the reference golden vectors, C-ABI surface, sharding logic."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from oracle import ref_cpu as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inrepo_tables_match_reference(golden):
    import speech_signal_processing_amd as pkg
    g = golden("mfcc_inrepo")
    for fs, L in [(8000, 512), (16000, 512), (16000, 256), (8000, 1024)]:
        fb, fr = pkg.frontend.mfccInitFilterBanks(fs, L)
        np.testing.assert_allclose(fb, g[f"fbank_{fs}_{L}"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(fr, g[f"freqs_{fs}_{L}"], rtol=0, atol=1e-12)
        t = pkg.preset_inrepo(fs, L, L // 2)
        cfg, w, fold, dct = O.inrepo_tables(fs, L, L // 2)
        np.testing.assert_allclose(t.window, w, atol=1e-7)
        np.testing.assert_allclose(t.fbank, fold, rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(t.dct, dct, atol=1e-7)
        assert t.cfg.as_dict() == {k: pytest.approx(v) for k, v in cfg.items()}


def test_sidekit_and_librosa_tables_match_oracle():
    import speech_signal_processing_amd as pkg
    for kw in (dict(), dict(window="hamming", delta_order=2, cmvn=1), dict(fs=8000, maxfreq=4000)):
        t = pkg.preset_sidekit(**kw)
        cfg, w, fb, dct = O.sidekit_tables(**kw)
        assert t.cfg.as_dict() == {k: pytest.approx(v) for k, v in cfg.items()}
        np.testing.assert_allclose(t.window, w, atol=1e-7)
        np.testing.assert_allclose(t.fbank, fb, rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(t.dct, dct, atol=1e-7)
    t = pkg.preset_sidekit()
    assert t.fbank.shape == (24, 257) and int((t.fbank != 0).sum()) == 454
    t = pkg.preset_librosa(8000, 13)
    cfg, w, fb, dct = O.librosa_tables(8000, 13)
    assert t.cfg.as_dict() == {k: pytest.approx(v) for k, v in cfg.items()}
    np.testing.assert_allclose(t.window, w, atol=1e-7)
    np.testing.assert_allclose(t.fbank, fb, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(t.dct, dct, atol=1e-7)


def test_dct_matrix_is_scipy_ortho():
    from scipy.fftpack import dct
    import speech_signal_processing_amd as pkg
    x = np.random.default_rng(0).standard_normal((5, 40))
    np.testing.assert_allclose(x @ pkg.frontend.dct2_ortho(40, 0, 13).T, dct(x, type=2, norm="ortho", axis=-1)[:, :13], atol=1e-12)
    np.testing.assert_allclose(x @ pkg.frontend.dct2_ortho(40, 1, 13).T, dct(x, type=2, norm="ortho", axis=-1)[:, 1:14], atol=1e-12)


def test_num_frames_rules():
    import speech_signal_processing_amd as pkg
    sk, ir, lb = pkg.preset_sidekit().cfg, pkg.preset_inrepo().cfg, pkg.preset_librosa().cfg
    assert [sk.num_frames(n) for n in (0, 399, 400, 559, 560, 16000, 48000)] == [0, 0, 1, 1, 2, 98, 298]
    assert [ir.num_frames(n) for n in (0, 1, 256, 257, 24000)] == [0, 1, 1, 2, 94]
    assert [lb.num_frames(n) for n in (0, 1025, 48000)] == [0, 3, 94]
    for cfg in (sk, ir, lb):
        for n in (0, 1, 400, 5077, 48000):
            assert cfg.num_frames(n) == O.num_frames(n, cfg.as_dict())


def test_c_abi_exports_every_declared_symbol():
    """The .so loads without a GPU and exports exactly the entry points include/ssp.h declares."""
    from speech_signal_processing_amd import _lib
    header = open(os.path.join(ROOT, "include", "ssp.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(ssp_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ssp_abi_version() == _lib.ABI_VERSION == 4
    assert ctypes.sizeof(_lib.ssp_mfcc_cfg) == 18 * 4


def test_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a gfx950 device every compute entry point raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speech_signal_processing_amd import api, _lib
    with pytest.raises(_lib.SspError):
        api.Context(0)
    from speech_signal_processing_amd.utils import processing as P
    with pytest.raises(_lib.SspError):
        P.MFCC(np.zeros(1000, dtype=np.float32))
    with pytest.raises(_lib.SspError):
        P.MFCC(np.zeros(1000, dtype=np.float32), 8000, 400, 160)      # the any-frame-size path
    from speech_signal_processing_amd import GMM_UBM, MFCC_DTW, d_vector
    from speech_signal_processing_amd.gmm_train import GaussianMixture
    with pytest.raises(_lib.SspError):
        GaussianMixture(n_components=2).fit(np.zeros((10, 3), dtype=np.float32))   # EM training
    with pytest.raises(_lib.SspError):
        MFCC_DTW.distance_dtw(np.zeros(5), np.ones(7))                            # DTW matcher
    with pytest.raises(_lib.SspError):
        MFCC_DTW.generate_template([np.zeros(5), np.ones(7)])
    with pytest.raises(_lib.SspError):
        d_vector.cosine_scores(np.zeros((2, 4), np.float32), np.ones((3, 4), np.float32))
    with pytest.raises(_lib.SspError):
        GMM_UBM.delta(np.zeros((5, 3)))
    with pytest.raises(_lib.SspError):
        GMM_UBM.plp(np.zeros(16000, dtype=np.float32))                           # PLP features
    with pytest.raises(_lib.SspError):
        GMM_UBM.extract_feature([np.zeros(16000, dtype=np.float32)], [0], feature_type='PLP')


def test_frame_count_and_dim_helpers_need_no_gpu():
    from speech_signal_processing_amd import _lib, api, preset_sidekit
    lib = _lib.load()
    cs = api._cfg_struct(preset_sidekit(delta_order=2).cfg)
    n = ctypes.c_int64()
    assert lib.ssp_mfcc_num_frames(ctypes.byref(cs), 48000, ctypes.byref(n)) == 0 and n.value == 298
    d = ctypes.c_int32()
    assert lib.ssp_mfcc_out_dim(ctypes.byref(cs), ctypes.byref(d)) == 0 and d.value == 39
    assert lib.ssp_mfcc_num_frames(None, 10, ctypes.byref(n)) == _lib.SSP_ERR_INVALID
    assert b"bad argument" in lib.ssp_last_error()


def test_model_pickles_round_trip(tmp_path):
    """the reference's persistence formats (GMM_UBM.py:173-179 model pickles) need no GPU: a fitted-looking
    gmm_train.GaussianMixture survives pickle and save_models / load_models write the reference's two file names"""
    import pickle
    from speech_signal_processing_amd import GMM_UBM
    from speech_signal_processing_amd.gmm_train import GaussianMixture
    rng = np.random.default_rng(0)
    def fake(K=3, D=5):
        g = GaussianMixture(n_components=K)
        g.weights_, g.means_, g.covariances_ = np.full(K, 1 / K), rng.standard_normal((K, D)), rng.uniform(1, 2, (K, D))
        g.precisions_cholesky_ = 1 / np.sqrt(g.covariances_)
        return g
    g2 = pickle.loads(pickle.dumps(fake()))
    assert g2.means_.shape == (3, 5) and g2._ctx is None
    gm, ubm = [fake(), fake()], fake()
    GMM_UBM.save_models(gm, ubm, str(tmp_path / "Model"))
    assert sorted(p.name for p in (tmp_path / "Model").iterdir()) == ["GMM_MFCC_model.pkl", "UBM_MFCC_model.pkl"]
    gm2, ubm2 = GMM_UBM.load_models(str(tmp_path / "Model"))
    assert len(gm2) == 2 and np.array_equal(ubm2.means_, ubm.means_)


def test_dft_tables_any_size():
    """host tables of the any-frame-size MFCC path: DFT matrix + folded filterbank reproduce fbank . |FFT| for even and odd L"""
    from speech_signal_processing_amd.utils import processing as P
    from speech_signal_processing_amd import frontend as F
    rng = np.random.default_rng(0)
    for L in (400, 401, 10):
        Wt, fold = P._dft_tables(16000, L)
        bank, _ = F.mfccInitFilterBanks(16000, L)
        x = rng.standard_normal(L)
        nb = L // 2 + 1
        re, im = Wt[:nb].astype(np.float64) @ x, Wt[nb:].astype(np.float64) @ x
        assert np.abs(bank @ np.abs(np.fft.fft(x)) - fold.astype(np.float64) @ np.sqrt(re * re + im * im)).max() < 1e-6


def test_plp_front_tables_match_oracle():
    """Bark filterbank / band count of the PLP front end against the oracle's restatement of fft2barkmx"""
    import speech_signal_processing_amd as pkg
    for fs in (8000, 16000, 44100):
        t = pkg.preset_sidekit_plp(fs=fs)
        cfg, w, fb, eye = O.sidekit_plp_tables(fs)
        assert t.cfg.as_dict() == {k: pytest.approx(v) for k, v in cfg.items()}
        assert t.cfg.n_filt == O.plp_num_bands(fs) == pkg.frontend.plp_num_bands(fs)
        np.testing.assert_allclose(t.fbank, fb, rtol=1e-6, atol=1e-30)
        np.testing.assert_allclose(t.window, w, atol=1e-7)
        assert np.array_equal(t.dct, np.eye(t.cfg.n_filt, dtype=np.float32))


def test_wav_read_matches_scipy(tmp_path):
    """utils/tools.py:45-58 — read() returns what scipy.io.wavfile.read (the reference's call) returns; save_wave_file round-trips."""
    from scipy.io import wavfile
    from speech_signal_processing_amd.utils import tools
    rng = np.random.default_rng(0)
    cases = {"i16": (rng.integers(-30000, 30000, 4001)).astype(np.int16), "f32": rng.standard_normal(777).astype(np.float32),
             "u8": rng.integers(0, 255, 100).astype(np.uint8), "i32": rng.integers(-2 ** 30, 2 ** 30, 55).astype(np.int32),
             "stereo": rng.integers(-3000, 3000, (321, 2)).astype(np.int16)}
    for name, x in cases.items():
        path = str(tmp_path / (name + ".wav"))
        wavfile.write(path, 16000 if name != "u8" else 8000, x)
        fs0, a0 = wavfile.read(path)
        fs1, a1 = tools.read(path)
        assert fs0 == fs1 and a0.dtype == a1.dtype and a0.shape == a1.shape and np.array_equal(a0, a1), name
    path = str(tmp_path / "saved.wav")
    tools.save_wave_file(path, [cases["i16"][:2000].tobytes(), cases["i16"][2000:].tobytes()], framerate=16000)
    fs, a = tools.read(path)
    assert fs == 16000 and np.array_equal(a, cases["i16"])
    w = tools.wave_read(path)
    assert w.getnframes() == 4001 and w.getframerate() == 16000
    w.close()
    t0 = tools.get_time()
    assert tools.get_time(t0) >= 0.0
    with pytest.raises(ValueError):
        (tmp_path / "bad.wav").write_bytes(b"not a wav file at all")
        tools.read(str(tmp_path / "bad.wav"))


def _build_c_example(out):
    import subprocess
    pkg = os.path.join(ROOT, "speech_signal_processing_amd")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "examples", "score_shard.c"), "-o", out, "-L", pkg, "-lsspgpu", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_c_abi_compiles_and_links_from_plain_c(tmp_path):
    """include/ssp.h is a C header (no C++, no torch types) and every entry point examples/score_shard.c uses — context, plan, segments,
    MFCC, GMM pack / score, communicator, all-gather — links against libsspgpu.so from gcc."""
    from speech_signal_processing_amd import _lib
    _lib.load()
    r = _build_c_example(str(tmp_path / "score_shard"))
    assert r.returncode == 0, r.stderr


def test_build_stamp_and_kernel_hash(tmp_path, monkeypatch):
    """build.py: an object is up to date only if it was compiled with today's command line; bench.py: a PMC reading is quoted only for
    the kernel source it was taken on."""
    from speech_signal_processing_amd import build as b
    obj = str(tmp_path / "x.o")
    assert not b._stamp_ok("gmm.hip", obj)                      # no stamp yet
    open(obj + ".cmd", "w").write(b._stamp_text("gmm.hip", obj))
    assert b.HERE not in b._stamp_text("gmm.hip", obj)          # (the stamp survives the tree being copied to another path)
    assert b._stamp_ok("gmm.hip", obj)
    assert "-fno-slp-vectorize" in b._cmd("gmm.hip", obj) and "-fno-slp-vectorize" not in b._cmd("ctx.hip", obj)
    monkeypatch.setattr(b, "FLAGS", b.FLAGS + ["-DSSP_SOMETHING"])
    assert not b._stamp_ok("gmm.hip", obj)                      # a flag changed: rebuild
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_sha256()
    assert len(h) == 64 and h == bench.kernel_source_sha256()
    import json
    for name in ("mfcc_hbm_traffic.json", "mfcc_valu_lds_pmc.json"):
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert len(j["kernel_source_sha256"]) == 64              # (whether it matches the tree is for bench.py to report, not a test)


def test_vote_matches_the_reference_rule():
    """MFCC_DTW.py:220-229: most frequent label; ties go to the label met first (stable sort over insertion order)."""
    from speech_signal_processing_amd.MFCC_DTW import vote
    assert vote([3, 1, 3, 2, 1, 3]) == 3
    assert vote(["b", "a", "a", "b"]) == "b"      # two each: the first met
    assert vote([7]) == 7
    assert vote(np.array([2, 2, 5, 5, 5])) == 5


def test_flop_count_script_matches_known_transform_counts():
    """tools/flop_count.py (the algorithmic roofline of bench.py's `roofline_flop`): its symbolic radix-4 run reproduces the textbook
    operation counts of small complex FFTs (4: 16 additions; 8: 52 + 4; 16: 144 + 24 — equal to split-radix there), never undercuts the
    split-radix count 4 N log2 N - 6 N + 8 on larger ones, counts structural zeros as free, and the headline dialect comes to 11 300."""
    import importlib.util
    import math
    spec = importlib.util.spec_from_file_location("flop_count", os.path.join(ROOT, "tools", "flop_count.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    for n, add, mul in ((4, 16, 0), (8, 52, 4), (16, 144, 24)):
        c = fc.Count()
        fc.dft(c, [fc.Z() for _ in range(n)])
        assert (c.add, c.mul) == (add, mul)
    for n in (64, 256, 1024):
        c = fc.Count()
        fc.dft(c, [fc.Z() for _ in range(n)])
        assert c.flop >= 4 * n * math.log2(n) - 6 * n + 8
    full, pruned = fc.Count(), fc.Count()
    fc.dft(full, [fc.Z() for _ in range(256)])
    fc.dft(pruned, [fc.Z(zero=k >= 200) for k in range(256)])
    assert pruned.flop < full.flop
    r = fc.count()
    assert r["flop_per_frame"] == 11300 and r["ops_per_frame"] == 8917 and abs(r["fma_fraction_of_peak"] - 0.634) < 1e-3
    assert r["stages"]["real FFT (pruned, N/2 complex + split)"]["flop"] == 7684   # the published real split-radix count is the smaller one


def _write_wav(path, data, rate=16000):
    import wave
    data = np.asarray(data)
    ch = 1 if data.ndim == 1 else data.shape[1]
    with wave.open(str(path), "wb") as w:
        w.setnchannels(ch)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(np.ascontiguousarray(data.astype("<i2")).tobytes())


def test_mfcc_dtw_load_test_walks_the_tree_like_the_reference(tmp_path):
    """MFCC_DTW.load_test (MFCC_DTW.py:155-184): <path>/<speaker>/*.wav, first channel of a stereo file, every second sample, the label
    of a file is its directory; a caller's own mfcc_extract is applied per file (this module's own go through one batched launch, which
    needs the GPU: tests/test_gpu_parity.py)."""
    from speech_signal_processing_amd import MFCC_DTW
    rng = np.random.default_rng(0)
    want = {}
    for spk, n in (("anna", 2), ("ben", 1)):
        (tmp_path / spk).mkdir()
        for i in range(n):
            mono = (3000 * rng.standard_normal(400 + 37 * i)).astype(np.int16)
            data = np.stack([mono, -mono], axis=1) if (spk, i) == ("anna", 1) else mono
            _write_wav(tmp_path / spk / ("u%d.wav" % i), data)
            want[(spk, "u%d.wav" % i)] = mono[::2]
    seen = []
    x, y = MFCC_DTW.load_test(str(tmp_path), mfcc_extract=lambda a: (seen.append(np.asarray(a).copy()), np.asarray(a, dtype=np.float64)[:5])[1])
    assert sorted(y) == ["anna", "anna", "ben"] and len(x) == 3 and all(v.shape == (5,) for v in x)
    got = sorted(a.tobytes() for a in seen)
    assert got == sorted(v.tobytes() for v in want.values())
    sx, sy = MFCC_DTW.sample(list(range(32)), [i // 8 for i in range(32)], sample_num=2, whole_num=8)
    assert len(sx) == 8 and sy == [0, 0, 1, 1, 2, 2, 3, 3] and [v % 8 for v in sx[:2]] == [v % 8 for v in sx[2:4]]


def test_d_vector_constructors_take_the_reference_arguments(tmp_path, monkeypatch):
    """d_vector.nn_model(n_class=40) (d_vector.py:165) constructs, positionally too; the store stays a keyword (or round 4's positional
    path).  Data_gen.extract_feature(feature_type, datatype) — the reference's signature (d_vector.py:59) — serves a cached
    feature/<datatype>_<type>_*.pkl pair without touching audio or the GPU, like the reference does."""
    from speech_signal_processing_amd import d_vector
    assert d_vector.nn_model(n_class=40).n_class == 40 and d_vector.nn_model(17).n_class == 17
    m = d_vector.nn_model(str(tmp_path / "d.pkl"))
    assert m.store == str(tmp_path / "d.pkl") and m.n_class == 40
    assert d_vector.nn_model(n_class=3, store="x.pkl").store == "x.pkl"
    monkeypatch.chdir(tmp_path)
    gen = d_vector.Data_gen()
    (tmp_path / "feature").mkdir()
    gen.save([np.ones((98, 13))], "dev_MFCC_feature")
    gen.save(["spk"], "dev_MFCC_label")
    f, lab = gen.extract_feature(feature_type="MFCC", datatype="dev")
    assert lab == ["spk"] and f[0].shape == (98, 13)
    f2, lab2 = gen.extract_feature("MFCC", "dev")
    assert lab2 == ["spk"]
    with pytest.raises(TypeError):
        gen.extract_feature([np.zeros(16000)])


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under speech_signal_processing_amd/ (Python or C++) names it, and in bench.py /
    __graft_entry__.py only the cpu_baseline_* legs and smoke() import it"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "speech_signal_processing_amd")
    for dp, _, files in os.walk(pkg):
        if "_obj" in dp or "__pycache__" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dp, f)  # (comments may cite it)
                assert not re.search(r"(dlopen|LoadLibrary|CDLL)\([^)]*oracle", text), os.path.join(dp, f)
    bench = open(os.path.join(root, "bench.py")).read()
    for m in re.finditer(r"from oracle import", bench):
        head = bench[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)[-1]
        assert fn.startswith("cpu_baseline"), fn
    entry = open(os.path.join(root, "__graft_entry__.py")).read()
    for m in re.finditer(r"from oracle import|import oracle", entry):
        fn = re.findall(r"^def (\w+)\(", entry[:m.start()], re.M)[-1]
        assert fn == "smoke", fn


def test_staging_pool_under_thread_sanitizer(tmp_path):
    """SURVEY section 5 (race detection): the SSP_HOST staging pool's slot logic (csrc/staging.hpp, the code round 5 put a mutex into)
    driven by eight threads on one pool and by two pools side by side, compiled with -fsanitize=thread (host code only; the buffer type
    is malloc-backed, no HIP runtime): no report from ThreadSanitizer, no slot ever held by two threads."""
    import subprocess
    exe = str(tmp_path / "stagepool_tsan")
    src = os.path.join(ROOT, "tests", "native", "stagepool_threads.cpp")
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", src, "-o", exe], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, "8", "20000"], capture_output=True, text=True, timeout=300,
                       env={**{k: v for k, v in os.environ.items() if k != "LD_PRELOAD"},   # (a sanitizer run of the CPU suite preloads another runtime)
                            "TSAN_OPTIONS": "halt_on_error=0:report_signal_unsafe=0"})
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert "ownership errors 0" in r.stdout


def test_flatten_signals_keeps_int16_pcm():
    """api.flatten_signals (what every reference-shaped extractor feeds the device from): a list whose utterances are ALL int16 PCM stays
    int16 (ssp_mfcc_run_i16 widens on the device: half the PCIe bytes, no host pass); anything else becomes float32 as before."""
    from speech_signal_processing_amd import api
    rng = np.random.default_rng(0)
    a, b = (rng.integers(-3000, 3000, 1000)).astype(np.int16), (rng.integers(-3000, 3000, 777)).astype(np.int16)
    flat, lens = api.flatten_signals([a, b.reshape(-1, 1)])
    assert flat.dtype == np.int16 and lens == [1000, 777] and np.array_equal(flat, np.concatenate([a, b]))
    one, lens1 = api.flatten_signals([a])
    assert one.dtype == np.int16 and lens1 == [1000] and np.shares_memory(one, a)            # a single utterance: no copy at all
    flat, lens = api.flatten_signals([a, b.astype(np.float64)])                                  # mixed: float32, the integer VALUES (no 1/32768)
    assert flat.dtype == np.float32 and lens == [1000, 777] and np.array_equal(flat[:1000], a.astype(np.float32))
    flat, lens = api.flatten_signals([])
    assert flat.dtype == np.float32 and flat.shape == (0,) and lens == []
    flat, lens = api.flatten_signals([np.zeros(0, np.int16), a])
    assert flat.dtype == np.int16 and lens == [0, 1000]
