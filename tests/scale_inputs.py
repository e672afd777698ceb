"""Seeded input generators of the G14 golden set (BASELINE-scale digests), shared by oracle/gen_golden.py — which runs the
imported reference over them in the build container — and by the GPU parity test that regenerates them on the GPU box.

NumPy + libm only: nothing here touches the reference, the product or the checker.  Everything that has to come out bit
for bit on another host is built from `numpy.random.default_rng` streams (stable by NumPy's compatibility policy) and
from `math.cos / math.sin` (the image's glibc, identical here and on the GPU box) — not from `numpy.cos`, whose SIMD
kernels depend on the host CPU's instruction set.
"""
import hashlib
import math

import numpy as np

SEED = 20250204                      # bench.py's workload seed
SHOULDER = {"r_arm": np.array([0.0, -0.2, 0.0]), "l_arm": np.array([0.0, 0.2, 0.0])}
N_CONFIG2 = 1 << 20                  # poses per arm, every outcome kept
N_CONFIG3 = 1 << 18                  # goal matrices whose is_reachable state is "reachable"
CHUNK_CONFIG2 = 1 << 22              # bench.make_config2_poses draws positions, then Euler angles, in chunks of this many
CHUNK_CONFIG3 = 1 << 21              # bench.make_config3_matrices likewise
SUBSAMPLE = 64                       # every 64th row also carries its numbers (joints, interval), not only its digest
N_TRAJ_CONFIG5, N_STEPS_CONFIG5 = 512, 1000   # G16: trajectories of config 5's generator walked by the reference itself
SUBSAMPLE_TRAJ = 32                  # every 32nd trajectory carries its joints and theta at every step


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def config2_unfiltered(arm, n=N_CONFIG2):
    """The first n poses bench.make_config2_poses draws (pos = shoulder + U(-0.7, 0.7)^3, eul = U(-pi, pi)^3, seed 20250204,
    drawn as one chunk of 2^22) BEFORE its reachability filter; for l_arm the same recipe around the left shoulder from the
    next seed.  Returns pos[n, 3], eul[n, 3]."""
    rng = np.random.default_rng(SEED + (0 if arm == "r_arm" else 1))
    pos = SHOULDER[arm] + rng.uniform(-0.7, 0.7, size=(CHUNK_CONFIG2, 3))
    eul = rng.uniform(-np.pi, np.pi, size=(CHUNK_CONFIG2, 3))
    return pos[:n].copy(), eul[:n].copy()


def config3_candidates():
    """Generator over the chunks of candidate poses bench.make_config3_matrices draws (seed 20250204 + 3): (pos, eul) pairs of
    2^21 rows each, endlessly."""
    rng = np.random.default_rng(SEED + 3)
    while True:
        pos = SHOULDER["r_arm"] + rng.uniform(-0.7, 0.7, size=(CHUNK_CONFIG3, 3))
        eul = rng.uniform(-np.pi, np.pi, size=(CHUNK_CONFIG3, 3))
        yield pos, eul


def matrices_from_pose(pos, eul):
    """4x4 goal matrices Rz(yaw) Ry(pitch) Rx(roll) | pos of scipy's "xyz" (extrinsic) convention, entry by entry with libm's
    cos / sin and IEEE products and sums in a fixed order: the same bits on every host."""
    n = len(pos)
    cs = np.array([[math.cos(v) for v in row] + [math.sin(v) for v in row] for row in eul.tolist()]).reshape(n, 6)
    ca, cb, cc, sa, sb, sc = (cs[:, k] for k in range(6))
    M = np.zeros((n, 4, 4))
    M[:, 0, 0] = cc * cb; M[:, 0, 1] = cc * sb * sa - sc * ca; M[:, 0, 2] = cc * sb * ca + sc * sa
    M[:, 1, 0] = sc * cb; M[:, 1, 1] = sc * sb * sa + cc * ca; M[:, 1, 2] = sc * sb * ca - cc * sa
    M[:, 2, 0] = -sb; M[:, 2, 1] = cb * sa; M[:, 2, 2] = cb * ca
    M[:, :3, 3] = pos
    M[:, 3, 3] = 1.0
    return M


def config3_from_kept(kept_bits, n=N_CONFIG3):
    """The n goal matrices of the set from the committed filter result: `kept_bits` = numpy.packbits of the reference's
    is_reachable flag over the candidate chunks, in order (as many whole chunks as the generator needed).
    Returns (pos_all, eul_all, kept_mask, M[n, 4, 4]): every candidate, which of them were kept, the first n kept as matrices."""
    kept = np.unpackbits(kept_bits).astype(bool)
    assert kept.size % CHUNK_CONFIG3 == 0
    gen = config3_candidates()
    chunks = [next(gen) for _ in range(kept.size // CHUNK_CONFIG3)]
    pos = np.concatenate([c[0] for c in chunks])
    eul = np.concatenate([c[1] for c in chunks])
    idx = np.flatnonzero(kept)[:n]
    assert idx.size == n
    return pos, eul, kept, matrices_from_pose(pos[idx], eul[idx])


def config5_trajectories(n_traj=N_TRAJ_CONFIG5, n_steps=N_STEPS_CONFIG5):
    """Goal matrices [n_steps, n_traj, 4, 4] of config 5's task-space generator (shaped like the reference's tests/test_sdk.py:38-63, as
    bench.make_config5_trajectories: centre (0.65, -0.2, 0; 0, -pi/2, 0), amplitudes 0.35 m / pi/6 rad, frequencies 0.6 ... 0.47,
    t = k / 120 + 11 + phase), phases from default_rng(20250204 + 5) * 40 s, sines and cosines from libm: the same bits on every host."""
    phase = np.random.default_rng(SEED + 5).uniform(0.0, 40.0, size=n_traj)
    c0 = [0.65, -0.2, 0.0, 0.0, -math.pi / 2, 0.0]
    amp = [0.35, 0.35, 0.35, math.pi / 6, math.pi / 6, math.pi / 6]
    freq = [0.6, 0.34, 0.78, 0.18, 0.31, 0.47]
    t = (np.arange(n_steps)[:, None] / 120.0 + 11.0) + phase[None, :]
    flat = t.reshape(-1).tolist()
    v = [np.array([c + a * math.sin(f * x) for x in flat]) for c, a, f in zip(c0, amp, freq)]
    pos = np.stack(v[:3], axis=1)
    eul = np.stack(v[3:], axis=1)
    return matrices_from_pose(pos, eul).reshape(n_steps, n_traj, 4, 4)


def mirror_matrices(M):
    """The l_arm goals that mirror r_arm goals (pose_l = (x, -y, z; -roll, pitch, -yaw), the reference's
    test_random_reachability.py:156-166 rule): M_l = S M S with S = diag(1, -1, 1, 1) — sign flips only, exact."""
    s = np.array([1.0, -1.0, 1.0, 1.0])
    return M * (s[:, None] * s[None, :])
