"""Best-effort CPU figure for the MFCC path (CPU baseline (ii)): the oracle's vectorised numpy restatement run in
W worker processes (one BLAS thread each) for a few seconds.  Standalone on purpose: bench.py starts it as a child process, so no
process that has initialised the GPU forks workers.  Prints one JSON object."""
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def work(args):
    budget_s, n_samp, fs = args
    import numpy as np
    from oracle import ref_cpu as O
    cfg, w, fb, dct = O.sidekit_tables(delta_order=2)
    rng = np.random.default_rng(os.getpid())
    x = np.clip(0.3 * np.sin(2 * np.pi * 120 * np.arange(n_samp) / fs) + 0.05 * rng.standard_normal(n_samp), -1, 1).astype(np.float32)
    frames, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        frames += O.mfcc_pipeline(x, cfg, w, fb, dct).shape[0]
    return frames, time.perf_counter() - t0


def main():
    import multiprocessing as mp
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else min(16, os.cpu_count() or 1)
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    n_samp, fs = 48000, 16000
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(work, [(budget, n_samp, fs)] * workers)
    frames = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    print(json.dumps({"value": frames / wall, "unit": "frames/s", "cores": workers, "kind": "port",
                      "sample": "%d worker processes x oracle.ref_cpu.mfcc_pipeline on 3 s utterances (39-d), 1 BLAS thread each, %.1f s"
                                % (workers, wall), "host_cores": os.cpu_count()}))


if __name__ == "__main__":
    main()
