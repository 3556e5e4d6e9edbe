"""bench.py's output contract (CPU): the LAST stdout line is one compact, strict-JSON object under 4 KB carrying the contract's keys,
`roofline` and `cpu_baseline`; every stage's full dict, the env windows and the close-call tables live in the detail file.  Round 5's
line had grown to 22.8 KB and the driver could not parse it (BENCH_r05.json: parsed null) — this pins the size and the shape on a
canned full result (profiles/r05_bench_line.json: that very 22.8 KB line)."""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _strict(text):
    def no_const(c):
        raise ValueError("non-finite constant %s in the line" % c)
    return json.loads(text, parse_constant=no_const)


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))


def test_compact_line_of_a_full_result():
    import bench
    res = _canned()
    assert len(json.dumps(res)) > 20000                       # the canned result IS the line that did not parse
    text = bench.compact_line(res, "gpurun_out/bench_detail.json")
    assert "\n" not in text and len(text.encode()) < bench.LINE_LIMIT == 4096
    line = _strict(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(res["value"], rel=1e-5) and line["ms_per_step"] == pytest.approx(res["ms_per_step"], rel=1e-5)
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["dtype"] == "f32" and line["vs_baseline"] is None
    for k in ("workload", "utterances_per_gpu", "frames_per_gpu", "d_out", "parallelism", "world_size_observed", "backend", "kernel_ms_per_rank"):
        assert k in line["config"], k
    assert "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "bytes_per_frame"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert rf["achieved"] == pytest.approx(rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9, rel=1e-4)
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "host_cores"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port")
    assert set(line["value_normalised"]) == {"value", "ms_per_step"}
    assert line["detail"] == "gpurun_out/bench_detail.json" and "build" in line
    # one small summary per other stage
    for k in ("gmm", "cosine", "mfcc_ref26_cmvn", "mfcc_librosa", "plp", "gmm_em", "dvector_dnn", "dtw"):
        assert set(line[k]) <= {"value", "kernel_ms", "frac", "mfma_busy", "gathered_rows", "record_bytes", "ms_per_step"}, (k, line[k])
        assert line[k]["value"] > 0
    assert line["gmm"]["gathered_rows"] == res["gmm"]["gathered_rows"] and line["gmm"]["record_bytes"] == 12
    assert "env" not in line or set(line["env"]) == {"sclk_mhz", "power_w"}     # the per-window statistics stay in the detail file
    for k in ("gmm_bf16x3_close_calls", "cosine_close_calls", "roofline_flop"):
        assert k not in line


def test_compact_line_of_the_round6_detail():
    """the full result of the round-6 driver command (profiles/r06_bench_detail.json: every stage incl. the host-fed and precision-auto
    rows) -> the committed line (profiles/r06_bench_line.json) again, under the limit, every stage summarised"""
    import bench
    res = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
    text = bench.compact_line(res, "gpurun_out/r06/bench_detail.json")
    assert len(text.encode()) < bench.LINE_TARGET < bench.LINE_LIMIT
    line = _strict(text)
    for k in CONTRACT + ("roofline", "cpu_baseline", "cpu_baseline_parallel", "value_normalised", "build", "detail", "mfcc_host_fed", "gmm_auto", "cosine_auto",
                         "gmm_host_fed", "cosine_host_fed", "gmm_cfg3_shape", "mfcc_inrepo", "plp", "dtw"):
        assert k in line, k
    assert line["roofline"]["traffic"] and 0.95 < line["roofline"]["traffic"] / line["roofline"]["algorithmic_bytes_per_launch"] < 1.1
    assert line["gmm"]["mfma_busy"] > 0.5 and line["cosine"]["mfma_busy"] > 0.5          # quoted: the counters were taken on this source
    assert line["cpu_baseline_parallel"]["usable_cores"] <= line["cpu_baseline_parallel"]["host_cores"]
    committed = _strict(open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read())
    assert committed["value"] == line["value"] and set(committed) == set(line)


def test_compact_line_eight_ranks_and_non_finite_values():
    import bench
    res = _canned()
    res["n_gpus"] = 8
    res["config"].update(world_size_observed=8, backend="nccl", parallelism="utterance-sharded x8", kernel_ms_per_rank=[9.4359123 + 0.01 * i for i in range(8)])
    res["cpu_baseline"] = {"skipped": "world > 1"}
    res.pop("cpu_baseline_parallel", None)
    res["roofline"]["traffic"] = float("nan")                # a counter that could not be read must not break the line
    res["gmm"]["value"] = float("inf")
    text = bench.compact_line(res, None)
    assert len(text.encode()) < 4096
    line = _strict(text)
    assert line["n_gpus"] == 8 and len(line["config"]["kernel_ms_per_rank"]) == 8 and line["config"]["backend"] == "nccl"
    assert line["roofline"]["traffic"] is None and line["gmm"]["value"] is None
    assert line["cpu_baseline"] == {"skipped": "world > 1"}


def test_compact_line_drops_extras_before_contract_keys():
    """a result with absurdly many / long extras still yields a line under the limit with every contract key"""
    import bench
    res = _canned()
    res["config"]["kernel_ms_per_rank"] = [9.4] * 8
    res["cpu_baseline"]["sample"] = "x" * 5000
    res["roofline"]["kernel"] = "k" * 5000
    res["config"]["workload"] = "w" * 5000
    for k in list(res):
        if isinstance(res[k], dict) and "value" in res[k] and k != "cpu_baseline":
            res[k]["unit"] = "u" * 300
    text = bench.compact_line(res, "d")
    assert len(text.encode()) < 4096
    line = _strict(text)
    for k in CONTRACT + ("roofline", "cpu_baseline", "detail"):
        assert k in line, k


def test_detail_file_is_strict_json(tmp_path):
    import numpy as np
    import bench
    res = _canned()
    res["x"] = {"nan": float("nan"), "np": np.float32(1.5), "arr": np.arange(3), "i": np.int64(7), "b": np.bool_(True), "inf": [float("-inf")]}
    p = bench.write_detail(res, str(tmp_path / "sub" / "bench_detail.json"))
    assert p and os.path.exists(p)
    back = _strict(open(p).read())
    assert back["x"] == {"nan": None, "np": 1.5, "arr": [0, 1, 2], "i": 7, "b": True, "inf": [None]}
    assert back["gmm_bf16x3_close_calls"]["points"] and back["env"]["sustained_mfcc"]          # the detail keeps what the line dropped
    assert math.isclose(back["value"], res["value"])


def test_tools_read_the_detail_file():
    """tools/readme_table.py takes the full result (the detail file); the driver-facing line no longer carries what it needs"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "readme_table.py"), os.path.join(ROOT, "profiles", "r05_bench_line.json")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "frames/s" in out.stdout
