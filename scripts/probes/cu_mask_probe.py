#!/usr/bin/env python3
"""Round 6: what does a CU-masked stream cost a chip-filling kernel?  (hipExtStreamCreateWithCUMask)

    python scripts/probes/cu_mask_probe.py

Config 2's solve kernel (1 Mi poses, ~31 us) launched K times back to back on (a) torch's stream, (b) a stream whose CU mask has all
256 bits set, (c) the lowest 240 / 224 / 192 bits (KFD deals mask bit i to XCC i mod 8, then round-robin over its shader engines:
the lowest 8 m bits = m CUs of every XCD), and then the question behind it: a lone-wave kernel (config 5's theta kernel needs a SIMD
with 276 free registers) launched on an UNMASKED stream while a masked stream keeps the other CUs full — does it start at once?
Measured with a 64-wave spin kernel of the library's debug entry (rsik_debug_math op 8 = the clock monitor waves): its first
timestamp against the launch.
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reachy2_symbolic_ik_amd import SymbolicIK  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]


def masked_stream(bits_set, total_bits=256):
    words = (total_bits + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(bits_set):
        mask[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), words, mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def main():
    n = 1 << 20
    pos, eul = bench.make_config2_poses(n, device=0)
    ik = bench._quiet(SymbolicIK, "r_arm", device=0)
    import numpy as np

    soa = torch.as_tensor(np.ascontiguousarray(np.concatenate([pos.T, eul.T], axis=0))).cuda()
    p = ik.solve_batch(soa, want_elbow=False, plan_only=True)
    launch = p["launch"]
    K = 200
    streams = {"torch default": None, "mask 256/256": masked_stream(256), "mask 240/256": masked_stream(240), "mask 224/256": masked_stream(224),
               "mask 192/256": masked_stream(192), "mask 128/256": masked_stream(128)}
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    for rnd in range(2):
        for name, st in streams.items():
            ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
            with ctx:
                cur = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(20):
                    launch(cur.cuda_stream)
                e0.record(cur)
                for _ in range(K):
                    launch(cur.cuda_stream)
                e1.record(cur)
                cur.synchronize()
                print(f"round {rnd}  {name:14s} {e0.elapsed_time(e1) / K * 1e3:7.2f} us per 1 Mi-pose solve launch", flush=True)
    # two masked streams side by side (what the pipeline's prepare and joints streams would be) against two unmasked ones
    for name, pair in (("two unmasked streams", (torch.cuda.Stream(), torch.cuda.Stream())), ("two streams, mask 240/256", (masked_stream(240), masked_stream(240)))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in pair:
            for _ in range(K):
                launch(st.cuda_stream)
        for st in pair:
            st.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name:28s} {dt / (2 * K) * 1e6:7.2f} us per launch (host clock, 2 x {K} launches)", flush=True)


if __name__ == "__main__":
    main()
