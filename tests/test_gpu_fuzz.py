"""Randomised regression (GPU): a few dozen random shapes / configurations per run through the same cross-checks as
tools/fuzz_mfcc.py, tools/fuzz_scoring.py (HIP path vs the float64 oracle, fused vs generic MFCC kernel) and tools/fuzz_mfcc_batch.py (machine-filling
ragged batches: the stream kernels vs the generic kernel on the same device arrays, the oracle on a sample of utterances) and
tools/fuzz_scoring_batch.py (large GMM / cosine batches: split-precision and precision-auto arg-max / arg-min against the fp32 path on every
row) and tools/fuzz_hostfed.py (the sliced host-fed pipeline and int16 input against the device-pointer path, bit for bit)."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed,cases", [("fuzz_mfcc.py", 101, 25), ("fuzz_scoring.py", 102, 15), ("fuzz_mfcc_batch.py", 103, 20),
                                               ("fuzz_scoring_batch.py", 104, 10), ("fuzz_hostfed.py", 106, 8)])
def test_fuzz(script, seed, cases, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", [script, str(seed), str(cases)])
    runpy.run_path(os.path.join(ROOT, "tools", script), run_name="__main__")
    assert "OK" in capsys.readouterr().out


def test_fuzz_walk_kernel_under_load(monkeypatch, capsys):
    """Half of a machine-filling batch's utterances hold a silent stretch (a tenth of those a NaN sample too): the scan kernel flags and
    the WALK instance walks again thousands of chunks per launch — the claim-by-ballot loop under load, not a handful of chunks."""
    monkeypatch.setenv("FUZZ_JUNK_FRAC", "0.5")
    monkeypatch.setenv("FUZZ_DIALECTS", "sidekit,sidekit,inrepo")
    monkeypatch.setattr(sys, "argv", ["fuzz_mfcc_batch.py", "105", "6"])
    runpy.run_path(os.path.join(ROOT, "tools", "fuzz_mfcc_batch.py"), run_name="__main__")
    assert "OK" in capsys.readouterr().out
