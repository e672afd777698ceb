#!/usr/bin/env python3
"""Builds build/variants/<name>.so from the working tree with extra compiler flags, for the A/B scripts:
scripts/build_variant.py cb16 -DRSIK_CHAIN_BATCH=16   (the embedded source hash is the tree's, so the Python side loads it)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from reachy2_symbolic_ik_amd import build as B

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.normpath(os.path.join(B.CSRC, "..", "..", "build", "variants", name + ".so"))
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = [B.hipcc()] + B.HIPCC_FLAGS + [f'-DRSIK_SOURCE_HASH="{B.source_hash()}"'] + extra + B.SOURCES + ["-o", out]
subprocess.run(cmd, check=True, cwd=B.CSRC)
print(os.path.normpath(out))
