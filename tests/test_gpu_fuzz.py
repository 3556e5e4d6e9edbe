"""Randomised regression (GPU): a few dozen random shapes / configurations per run through the same cross-checks as
tools/fuzz_mfcc.py and tools/fuzz_scoring.py (HIP path vs the float64 oracle, fused vs generic MFCC kernel)."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed,cases", [("fuzz_mfcc.py", 101, 25), ("fuzz_scoring.py", 102, 15)])
def test_fuzz(script, seed, cases, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", [script, str(seed), str(cases)])
    runpy.run_path(os.path.join(ROOT, "tools", script), run_name="__main__")
    assert "OK" in capsys.readouterr().out
