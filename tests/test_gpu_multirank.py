"""The N-rank launch path of bench.py on the GPU box: `python bench.py --gpus 2` starts two ranks itself (torch.distributed.run as a
child process), every rank runs the HIP MFCC pass and the HIP GMM scorer on its own utterance shard and the compact per-utterance
decisions are all-gathered.  With two devices visible the ranks use one GPU each over RCCL ("nccl"); on a one-GPU box both ranks
share device 0 and the collective runs over gloo (SSP_BENCH_REHEARSE=1) — same control flow, same kernels."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_hip_scorer():
    import torch
    env = dict(os.environ)
    two_gpus = torch.cuda.device_count() >= 2
    if not two_gpus:
        env["SSP_BENCH_REHEARSE"] = "1"
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    utts = 600
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--utts", str(utts), "--steps", "2", "--warmup", "1",
           "--stages", "mfcc,gmm", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["config"]["world_size_observed"] == 2
    assert r["config"]["backend"] == ("nccl" if two_gpus else "gloo")
    assert len(r["config"]["kernel_ms_per_rank"]) == 2 and all(ms > 0 for ms in r["config"]["kernel_ms_per_rank"])
    assert r["gmm"]["gathered_rows"] == 2 * utts          # every rank sees every utterance's decision after the gather
    assert r["value"] > 0 and r["gmm"]["value"] > 0
    # weak scaling bookkeeping: the whole-job frame count is both ranks' frames
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 - 2 * r["config"]["frames_per_gpu"]) <= 1e-6 * 2 * r["config"]["frames_per_gpu"]
