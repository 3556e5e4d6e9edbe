"""Best-effort CPU figure for the MFCC path (CPU baseline (ii)): the oracle's vectorised numpy restatement run in
W worker processes (one BLAS thread each) for a few seconds.  Standalone on purpose: bench.py starts it as a child process, so no
process that has initialised the GPU forks workers.  Prints one JSON object.

    python tools/cpu_parallel_baseline.py [workers | auto] [seconds]

`auto` = min(cores this process may run on, 128): the cores of os.cpu_count() cut by the scheduler affinity and by the cgroup's CPU quota
(a GPU box of the pool shows 256 host cores; what the container may use is what those two say).  The line states all of them, the
worker count used and the BLAS thread count per worker (threadpoolctl) — and, when more than 16 workers were used, the 16-worker figure
beside it (`value_16_workers`), so that the scaling over cores can be read from one run."""
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAX_WORKERS = 128


def cgroup_cpu_quota():
    """CPUs the cgroup grants (v2 cpu.max, v1 cfs quota / period), or None when unlimited / unreadable"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def usable_cores():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    q = cgroup_cpu_quota()
    if q:
        n = min(n, max(1, int(q + 0.5)))
    return n


_STATE = {}


def work(args):
    budget_s, start_at = args
    O, cfg, w, fb, dct, x = _STATE["O"], _STATE["cfg"], _STATE["w"], _STATE["fb"], _STATE["dct"], _STATE["x"]
    while time.time() < start_at:          # every worker starts its clock together (pool start-up is not part of the figure)
        time.sleep(0.001)
    frames, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        frames += O.mfcc_pipeline(x, cfg, w, fb, dct).shape[0]
    return frames, time.perf_counter() - t0


def run(workers, budget):
    import multiprocessing as mp
    with mp.get_context("fork").Pool(workers) as pool:
        start_at = time.time() + 0.5 + 0.01 * workers
        res = pool.map(work, [(budget, start_at)] * workers, chunksize=1)
    return sum(r[0] for r in res) / max(r[1] for r in res), max(r[1] for r in res)


def main():
    import numpy as np
    from oracle import ref_cpu as O
    arg = sys.argv[1] if len(sys.argv) > 1 else "auto"
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    usable = usable_cores()
    workers = min(usable, MAX_WORKERS) if arg == "auto" else int(arg)
    n_samp, fs = 48000, 16000
    # tables and the utterance are built once, before the fork (the workers share them copy-on-write)
    cfg, w, fb, dct = O.sidekit_tables(delta_order=2)
    rng = np.random.default_rng(5)
    x = np.clip(0.3 * np.sin(2 * np.pi * 120 * np.arange(n_samp) / fs) + 0.05 * rng.standard_normal(n_samp), -1, 1).astype(np.float32)
    _STATE.update(O=O, cfg=cfg, w=w, fb=fb, dct=dct, x=x)
    O.mfcc_pipeline(x, cfg, w, fb, dct)
    blas = None
    try:
        from threadpoolctl import threadpool_info
        blas = sorted({int(p.get("num_threads", 0)) for p in threadpool_info()}) or None
    except Exception:
        pass
    out = {"unit": "frames/s", "cores": workers, "kind": "port", "host_cores": os.cpu_count(), "usable_cores": usable,
           "cgroup_cpu_quota": cgroup_cpu_quota(), "blas_threads_per_worker": (blas[-1] if blas else None)}
    if workers > 16:
        out["value_16_workers"], _ = run(16, min(budget, 4.0))
    value, wall = run(workers, budget)
    out["value"] = value
    if "value_16_workers" in out:
        out["scaling_over_16_workers"] = value / out["value_16_workers"]
    out["sample"] = "%d worker processes x oracle.ref_cpu.mfcc_pipeline on 3 s utterances (39-d), %s BLAS thread(s) each, %.1f s; %d of %d host cores usable" % (
        workers, out["blas_threads_per_worker"], wall, usable, os.cpu_count() or 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
