"""Build libsspgpu.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m speech_signal_processing_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsspgpu.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
SOURCES = ["ctx.hip", "calib.hip", "comm.hip", "mfcc.hip", "mfcc_fast.hip", "mfcc_stream.hip", "mfcc_stream_walk.hip", "mfcc_stream2k.hip", "mfcc_plan.hip", "feat_ops.hip", "gmm.hip", "gmm_em.hip", "cosine.hip", "dense.hip", "dnn_chain.hip", "dtw.hip", "fastdtw.hip", "plp.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-munsafe-fp-atomics", "-Wno-pass-failed"] + os.environ.get("SSP_EXTRA_FLAGS", "").split()


# per-source flags.  gmm.hip: MFMA accumulators in ordinary VGPRs (the log-sum-exp epilogue reads all of them: from AGPRs that is one
# v_accvgpr_read per element) and no SLP re-packing of the epilogue's scalar adds into v_pk_add_f32 (slow beside MFMAs)
# mfcc_stream*.hip: no SLP re-packing either — the kernel's packed operations are written as such; what the vectorizer adds costs the
# headline instance eight registers (168 -> 160), which buy three more resident twiddles (SSP_STREAM_NTW 15): -1.3 % on the headline pass
SOURCE_FLAGS = {"gmm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
                "mfcc_stream.hip": ["-fno-slp-vectorize"], "mfcc_stream_walk.hip": ["-fno-slp-vectorize"]}


def _deps():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "ssp.h"))
    return out


def _cmd(src: str, obj: str) -> list:
    return [HIPCC, *FLAGS, *SOURCE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]


def _stamp_text(src: str, obj: str) -> str:
    """the command line with the package directory abstracted: the tree is copied to another path on the GPU box, and objects that travel
    with it must still count as built with today's flags there"""
    return " ".join(_cmd(src, obj)).replace(os.path.dirname(obj), "@OBJ@").replace(HERE, "@PKG@")


def _stamp_ok(src: str, obj: str) -> bool:
    """the object was compiled with exactly today's command line (flags are part of the up-to-date check: changing SSP_EXTRA_FLAGS or
    SOURCE_FLAGS must not silently reuse objects built with the old ones)"""
    try:
        with open(obj + ".cmd") as f:
            return f.read() == _stamp_text(src, obj)
    except OSError:
        return False


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(p) > t for p in _deps()):
        return True
    if os.path.isdir(OBJ_DIR):  # (a shipped .so without its objects — the GPU box — is taken as it is)
        for s in SOURCES:
            obj = os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")
            if os.path.exists(obj) and not _stamp_ok(s, obj):
                return True
    return False


def _compile(src: str) -> str:
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    srcp = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(p) for p in _deps() if p.endswith((".hpp", ".h")) or p == srcp)
    if os.path.exists(obj) and os.path.getmtime(obj) > newest and _stamp_ok(src, obj):
        return obj
    cmd = _cmd(src, obj)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    with open(obj + ".cmd", "w") as f:
        f.write(_stamp_text(src, obj))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


INFO = os.path.join(HERE, "build_info.json")  # what the last build() call did (git-ignored; read back by bench.py)


def _record(mode: str, compiled) -> None:
    import json
    import time
    try:
        with open(INFO, "w") as f:
            json.dump({"build_mode": mode, "compiled_sources": list(compiled), "time": time.time(),
                       "lib_bytes": os.path.getsize(LIB) if os.path.exists(LIB) else 0, "hipcc": HIPCC, "flags": FLAGS}, f)
    except OSError:
        pass


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile (hipcc, gfx950) what is out of date and link libsspgpu.so.  SSP_FORCE_BUILD=1 (or force=True) recompiles every
    source.  Records in build_info.json whether the call compiled ("compiled" / "forced") or found the shipped library up to date
    ("reused")."""
    force = force or bool(os.environ.get("SSP_FORCE_BUILD"))
    if not force and not needs_build():
        _record("reused", [])
        if verbose:
            print("reused", LIB)
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one builder at a time (ranks of a multi-process launch may all arrive here): the others wait, then find everything up to date
    import fcntl
    lock = open(os.path.join(HERE, ".build.lock"), "w")  # (outside OBJ_DIR: the force path empties that directory under the lock)
    fcntl.flock(lock, fcntl.LOCK_EX)
    if not force and not needs_build():
        _record("reused", [])
        if verbose:
            print("reused", LIB)
        return LIB
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    before = {s: os.path.getmtime(os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")) if os.path.exists(os.path.join(OBJ_DIR, os.path.splitext(s)[0] + ".o")) else 0.0
              for s in SOURCES}
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    compiled = [s for s, o in zip(SOURCES, objs) if os.path.getmtime(o) > before[s]]
    tmp = LIB + ".tmp.%d" % os.getpid()  # link beside the target, then rename: a reader never maps a half-written library
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp, LIB)
    _record("forced" if force else "compiled", compiled)
    if verbose:
        print("built", LIB, "(%d of %d sources compiled)" % (len(compiled), len(SOURCES)))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
