"""Builds csrc/librsik_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["rsik_lib.hip"]
DEPS = ["rsik_lib.hip", "rsik_kernel_solve.hpp", "rsik_kernel_discrete.hpp", "rsik_kernel_continuous.hpp", "rsik_kernel_pipeline.hpp", "rsik_kernel_stages.hpp",
        "rsik_kernel_state.hpp", "rsik_comm.hpp", "rsik_device.hpp", "rsik_math.hpp", "rsik_poly_gen.hpp",
        os.path.join("..", "..", "include", "rsik.h")]
OUT = os.path.join(CSRC, "librsik_hip.so")

# -ffp-contract=off: the compiler never fuses on its own (HIP's default "fast" mode fuses in the backend and ignores
# `#pragma clang fp contract`), so decision points of the reach test keep the reference's NumPy (unfused) rounding;
# everywhere else fused multiply-adds are written out explicitly (rsik_device.hpp: dot, cross, madd, ...).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")
    return exe


def source_hash() -> str:
    """sha256 over every source file the library is compiled from plus the compiler flags.  The build embeds it
    (-DRSIK_SOURCE_HASH, returned by rsik_build_id()), so a library that was built from other sources — e.g. a stale
    .so that travelled to the GPU box after an edit — is recognised whatever the file times say."""
    h = hashlib.sha256()
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as fh:
            h.update(d.encode() + b"\0" + fh.read() + b"\0")
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:32]


_MARK = re.compile(rb"RSIK_SRC_HASH=([0-9a-f]{32})")


def built_hash(path: str = OUT) -> str:
    """The source hash embedded in a built library ("" if there is none), read from the file without loading it."""
    try:
        with open(path, "rb") as fh:
            m = _MARK.search(fh.read())
    except OSError:
        return ""
    return m.group(1).decode() if m else ""


def needs_build() -> bool:
    return built_hash() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles the library if it is missing or stale.  Safe when several ranks (bench.py --gpus N, torchrun, pytest-xdist)
    find a stale library at once: they queue on a lock file, the first one compiles into a temporary file of its own and
    renames it into place, the others re-check under the lock and load the result."""
    if not (force or needs_build()):
        return OUT
    import fcntl

    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():  # (another process may have built it while this one waited)
                tmp = f"{OUT}.{os.getpid()}.tmp"
                cmd = [hipcc()] + HIPCC_FLAGS + [f'-DRSIK_SOURCE_HASH="{source_hash()}"'] + SOURCES + ["-o", tmp]
                if verbose:
                    print(" ".join(cmd))
                try:
                    subprocess.check_call(cmd, cwd=CSRC)
                    os.replace(tmp, OUT)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
