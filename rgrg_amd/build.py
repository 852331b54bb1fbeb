"""Build ``librgrg_hip.so`` (gfx950) in-tree with hipcc.

The library is the product's only compute path; there is no fallback.  hipcc
cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build
container (``__graft_entry__.build``).
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librgrg_hip.so")
SOURCES = ("runtime.hip", "gemm_f32.hip", "gemm_bf16.hip", "detector_ops.hip", "det_train.hip", "decoder.hip", "train_ops.hip", "attn_train16.hip")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the RGRG HIP library cannot be built (ROCm toolchain required)")


HASH_PATH = os.path.join(LIB_DIR, "librgrg_hip.srchash")
HEADERS = ("common.h", "skinny_direct.inc", "persistent.inc", "gemm_kp.inc")


def _source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "rgrg_hip.h")]
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def is_stale() -> bool:
    """True when the library is missing or was built from other sources.  Content hash, not mtimes: the snapshot
    that carries the built library to a GPU box does not preserve file times."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as f:
        return f.read().strip() != _source_hash()


def _compile(hipcc: str, src: str, tmpdir: str, verbose: bool) -> str:
    obj = os.path.join(tmpdir, src.replace(".hip", ".o"))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
           "-Wno-pass-failed", "-c", os.path.join(CSRC, src), "-o", obj]
    cmd[1:1] = os.environ.get("RGRG_HIPCC_FLAGS", "").split()   # experiments (-D switches of a kernel under study)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``rgrg_amd/lib/librgrg_hip.so``.

    Safe against concurrent callers (every rank of a torchrun job calls ``_hip.load()`` at the same time): the build
    runs under an exclusive ``flock`` on the lib directory, objects go to a private temporary directory, and the
    library and its source hash are put in place with ``os.replace``; a rank that waited for the lock re-checks the
    hash and returns without rebuilding."""
    if not force and not is_stale():
        return LIB_PATH
    import fcntl
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = _hipcc()
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():  # another process built it while this one waited
                return LIB_PATH
            with tempfile.TemporaryDirectory(prefix="build.", dir=LIB_DIR) as tmpdir:
                with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
                    objs = list(pool.map(lambda src: _compile(hipcc, src, tmpdir, verbose), SOURCES))
                tmp = os.path.join(tmpdir, "librgrg_hip.so")
                subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs, check=True)
                os.replace(tmp, LIB_PATH)
            tmp_hash = HASH_PATH + f".{os.getpid()}"
            with open(tmp_hash, "w") as f:
                f.write(_source_hash() + "\n")
            os.replace(tmp_hash, HASH_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
