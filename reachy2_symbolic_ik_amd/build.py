"""Builds csrc/librsik_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["rsik_lib.hip"]
DEPS = ["rsik_lib.hip", "rsik_device.hpp", "rsik_math.hpp", "rsik_poly_gen.hpp", os.path.join("..", "..", "include", "rsik.h")]
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


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc()] + HIPCC_FLAGS + SOURCES + ["-o", OUT]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
