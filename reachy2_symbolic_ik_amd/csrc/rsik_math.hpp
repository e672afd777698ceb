// rsik_math.hpp — fp64 elementary functions tailored to this path on gfx950.
//
// The per-pose IK is fp64-VALU-issue bound (every wave64 fp64 instruction occupies a SIMD for 4 cycles) and the
// general-purpose device-library calls dominate its instruction count (atan2 ~100, sincos ~75 executed,
// sqrt 22, IEEE division 12 VALU instructions each).  The inputs here are bounded, finite and never denormal, so the
// special-case scaffolding can go:
//   fast_rcp      v_rcp_f64 + 2 Newton steps                      5 instr, <= 1 ulp
//   sqrt_cr       v_rsq_f64 + Goldschmidt + 2 residual fix-ups   10 instr, correctly rounded for normal x > 0
//   sqrt_rsqrt    the same sequence, also returns 1/sqrt(x)      12 instr
//   fast_atan2    min/max quotient + degree-20 odd polynomial    ~40 instr, abs error < 3e-16 rad
//   fast_sincos   3-term Cody-Waite reduction + degree-5 kernels ~36 instr, |x| < 1e5, abs error < 2e-16
// Polynomial coefficients and the pi/2 split are derived by scripts/gen_poly.py (Chebyshev interpolation at 60 digits).
#pragma once

#include <hip/hip_runtime.h>

namespace rsik {

__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// Correctly rounded sqrt for normal, finite x > 0 (x == 0 returns NaN: callers guard exact zeros).
__device__ __forceinline__ double sqrt_cr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return g;
}

// s = sqrt(x) (correctly rounded), rs = 1/sqrt(x) (<= 1 ulp), normal finite x > 0.
__device__ __forceinline__ void sqrt_rsqrt(double x, double& s, double& rs) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    s = g;
    rs = h + h;
}

__device__ __forceinline__ double rsqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    return h + h;
}

// atan2 for finite arguments.  atan2(0, +0) = 0, atan2(0, -0) = pi like the C library.
__device__ __forceinline__ double fast_atan2(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    double a = mn * fast_rcp(mx);
    a = (mx == 0.0) ? 0.0 : a;
    const double s = a * a;
    // atan(a) = a + a*s*Q(s) on [0, 1]; Q from scripts/gen_poly.py (degree-20 fit of atan(sqrt(s))/sqrt(s))
    double q = 1.2631178430477426e-05;
    q = fma(q, s, -0.00014617088163625013);
    q = fma(q, s, 0.0008033604181626027);
    q = fma(q, s, -0.0028047655531701315);
    q = fma(q, s, 0.007038646202989813);
    q = fma(q, s, -0.013674139288399478);
    q = fma(q, s, 0.021740213830758134);
    q = fma(q, s, -0.02970071773623423);
    q = fma(q, s, 0.03651081352721035);
    q = fma(q, s, -0.042125723963855326);
    q = fma(q, s, 0.04719723992321112);
    q = fma(q, s, -0.052527255573225747);
    q = fma(q, s, 0.05880342002401543);
    q = fma(q, s, -0.06666371187721098);
    q = fma(q, s, 0.0769227555520563);
    q = fma(q, s, -0.09090906605656898);
    q = fma(q, s, 0.11111110982087126);
    q = fma(q, s, -0.14285714281592693);
    q = fma(q, s, 0.19999999999929946);
    q = fma(q, s, -0.3333333333333286);
    double r = fma(a * s, q, a);
    r = (ay > ax) ? (1.5707963267948966 - r) : r;
    r = __builtin_signbit(x) ? (3.141592653589793 - r) : r;
    return __builtin_copysign(r, y);
}

// sin and cos of |x| < ~1e5 (three-term Cody-Waite reduction of pi/2, exact for |k| < 2^20).
__device__ __forceinline__ void fast_sincos(double x, double* sn, double* cs) {
    const double k = __builtin_rint(x * 0.6366197723675814);
    double r = fma(-k, 1.5707963267341256, x);      // 33-bit head of pi/2: product exact
    r = fma(-k, 6.077100506303966e-11, r);         // next 33 bits
    r = fma(-k, 2.0222662487959506e-21, r);        // tail
    const int q = (int)k;
    const double z = r * r;
    double ps = 1.5918129294866608e-10;
    ps = fma(ps, z, -2.5051131845003624e-08);
    ps = fma(ps, z, 2.755731610255244e-06);
    ps = fma(ps, z, -0.00019841269836758574);
    ps = fma(ps, z, 0.008333333333330948);
    ps = fma(ps, z, -0.16666666666666666);
    const double s0 = fma(r * z, ps, r);
    double pc = -1.1382632425521717e-11;
    pc = fma(pc, z, 2.08761462684032e-09);
    pc = fma(pc, z, -2.7557317271729793e-07);
    pc = fma(pc, z, 2.480158729876569e-05);
    pc = fma(pc, z, -0.0013888888888887398);
    pc = fma(pc, z, 0.041666666666666664);
    const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
    const bool swap = (q & 1) != 0;
    double s = swap ? c0 : s0;
    double c = swap ? s0 : c0;
    s = (q & 2) ? -s : s;
    c = ((q + 1) & 2) ? -c : c;
    *sn = s;
    *cs = c;
}

}  // namespace rsik
