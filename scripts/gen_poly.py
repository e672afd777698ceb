#!/usr/bin/env python3
"""Derives the polynomial coefficients used by reachy2_symbolic_ik_amd/csrc/rsik_math.hpp.

Near-minimax fits by Chebyshev interpolation at 60-digit precision (mpmath), rounded to binary64:
  atan(a)  = a * P(a^2)                    a in [0, 1]
  sin(r)   = r + r^3 * S(r^2)              |r| <= pi/4
  cos(r)   = 1 - r^2/2 + r^4 * C(r^2)      |r| <= pi/4
Prints C arrays (highest degree first, for Horner) and the measured max error of a binary64 evaluation.

    python scripts/gen_poly.py [atan degrees ...]
"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(g, lo, hi, n):
    """Monomial coefficients c[0..n] (in u) of the degree-n interpolant of g at the n+1 Chebyshev nodes of [lo, hi]."""
    nodes = [(lo + hi) / 2 + (hi - lo) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * (n + 1))) for k in range(n + 1)]
    A = mp.matrix(n + 1, n + 1)
    b = mp.matrix(n + 1, 1)
    for i, u in enumerate(nodes):
        for j in range(n + 1):
            A[i, j] = u**j
        b[i] = g(u)
    c = mp.lu_solve(A, b)
    return [c[j] for j in range(n + 1)]


def horner(c, u):
    acc = np.full_like(u, c[-1])
    for k in range(len(c) - 2, -1, -1):
        acc = acc * u + c[k]
    return acc


def emit(name, c):
    print(f"// {name}: {len(c)} coefficients, highest degree first")
    print("{" + ", ".join(f"{float(x)!r}" for x in reversed(c)) + "}")


def g_atan(u):
    if u == 0:
        return mp.mpf(1)
    r = mp.sqrt(u)
    return mp.atan(r) / r


def g_sin(u):  # (sin r - r)/r^3
    if u == 0:
        return -mp.mpf(1) / 6
    r = mp.sqrt(u)
    return (mp.sin(r) - r) / r**3


def g_cos(u):  # (cos r - 1 + r^2/2)/r^4
    if u == 0:
        return mp.mpf(1) / 24
    r = mp.sqrt(u)
    return (mp.cos(r) - 1 + u / 2) / u**2


def main():
    xs = np.linspace(0.0, 1.0, 200001)
    degs = [int(a) for a in sys.argv[1:]] or [19, 20, 21, 22]
    for n in degs:
        c = cheb_fit(g_atan, mp.mpf(0), mp.mpf(1), n)
        cd = [float(x) for x in c]
        approx = xs * horner(cd, xs * xs)
        exact = np.array([float(mp.atan(mp.mpf(float(x)))) for x in xs[::50]])
        err = np.max(np.abs(approx[::50] - exact))
        print(f"atan degree {n} in a^2: max abs err {err:.3e}")
        emit(f"atan P degree {n}", c)
    rs = np.linspace(-np.pi / 4, np.pi / 4, 200001)
    for n in (5, 6):
        c = cheb_fit(g_sin, mp.mpf(0), (mp.pi / 4) ** 2, n)
        cd = [float(x) for x in c]
        approx = rs + rs**3 * horner(cd, rs * rs)
        exact = np.array([float(mp.sin(mp.mpf(float(x)))) for x in rs[::50]])
        print(f"sin degree {n}: max abs err {np.max(np.abs(approx[::50] - exact)):.3e}")
        emit(f"sin S degree {n}", c)
    for n in (5, 6):
        c = cheb_fit(g_cos, mp.mpf(0), (mp.pi / 4) ** 2, n)
        cd = [float(x) for x in c]
        u = rs * rs
        approx = 1.0 - 0.5 * u + u * u * horner(cd, u)
        exact = np.array([float(mp.cos(mp.mpf(float(x)))) for x in rs[::50]])
        print(f"cos degree {n}: max abs err {np.max(np.abs(approx[::50] - exact)):.3e}")
        emit(f"cos C degree {n}", c)


if __name__ == "__main__":
    main()
