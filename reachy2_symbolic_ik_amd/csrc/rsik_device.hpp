// rsik_device.hpp — per-pose analytic IK math for gfx950 (one pose per lane, IEEE float64).
//
// Not a translation of the reference's 4x4-matrix / scipy formulation: every frame is reduced
// to the 3-vectors that are actually consumed, all rotations about a joint axis are built from
// normalised vector components (no sin/cos of an angle that was just produced by atan2), and the
// least-squares plane/plane line of symbolic_ik.py:570-606 is replaced by its closed form.
// Decision points (threshold comparisons that fix the reachability flag / state / interval
// orientation) keep the reference's operand order so flags stay bit-exact; they are marked [D].
//
// Reference citations: S: = src/reachy2_symbolic_ik/symbolic_ik.py, U: = utils.py, C: = control_ik.py.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rsik.h"
#include "rsik_math.hpp"

namespace rsik {

// Conditions that are false for generic input (projections, clamps, exact singularities).  -DRSIK_HOT_ONLY compiles
// them out: an ANALYSIS build whose static instruction mix is the executed mix of the common path (scripts/isa_hist.py).
#ifdef RSIK_HOT_ONLY
#define RSIK_RARE(c) (false)
#else
#define RSIK_RARE(c) (__builtin_expect(!!(c), 0))
#endif


// -DRSIK_ISA_MARKS: named comment lines in the assembly (scripts/isa_sections.py counts instructions per section;
// the markers pin the schedule at their position, so this is an ANALYSIS build)
#ifdef RSIK_ISA_MARKS
#define RSIK_MARK(name) asm volatile("; RSIK_MARK " name)
#else
#define RSIK_MARK(name)
#endif

constexpr double kPi = 3.141592653589793;  // == math.pi
constexpr double kTwoPi = 2 * kPi;

struct ArmC {
    double v[RSIK_ARM_CONSTS_COUNT];
};

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
// The translation unit is compiled with -ffp-contract=off: fused multiply-adds are written out where the
// rounding of the reference (NumPy, unfused) does not matter; decision points use the *_d forms.
__device__ __forceinline__ double dot(V3 a, V3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x))};
}
// a * s + b
__device__ __forceinline__ V3 madd(V3 a, double s, V3 b) { return {fma(a.x, s, b.x), fma(a.y, s, b.y), fma(a.z, s, b.z)}; }
// Decision points are evaluated with the reference's (NumPy, unfused) rounding.
__device__ __forceinline__ double dot_d(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double norm(V3 a) { return sqrt_cr(dot_d(a, a)); }
// |a| (correctly rounded) and a/|a| through one v_rsq_f64 sequence instead of a sqrt and three IEEE divisions
__device__ __forceinline__ V3 normalized(V3 a, double& len) {
    double rs;
    sqrt_rsqrt(dot(a, a), len, rs);
    return a * rs;
}
__device__ __forceinline__ V3 normalized(V3 a) {
    return a * rsqrt_fast(dot(a, a));
}

// numpy.isclose(a, b), default rtol/atol: |a-b| <= 1e-8 + 1e-5*|b|
__device__ __forceinline__ bool np_isclose(double a, double b) { return fabs(a - b) <= (1e-8 + 1e-5 * fabs(b)); }
// include/rsik.h "Rows that are not numbers": a goal (pose or matrix) with a NaN or an infinity in it is RSIK_STATE_INVALID_INPUT.
// One compare per entry (|x| < inf is false for both), the lanes' answers meet on the scalar side.
template <int N>
__device__ __forceinline__ bool all_finite(const double (&v)[N]) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < N; k++) ok = ok && (fabs(v[k]) < __builtin_inf());
    return ok;
}

// Python float modulo `a % (2*pi)` (result in [0, 2pi)).  CPython computes fmod(a, b) (exact) and adds b
// when the signs differ.  For the small quotients on this path a - q*2pi is exactly representable, so one
// fma reproduces fmod bit for bit; the two fix-ups only fire when a*(1/2pi) rounded across an integer.
__device__ __forceinline__ double pymod_2pi(double a) {
    double q = floor(a * 0.15915494309189535);
    double m = fma(-q, kTwoPi, a);
    if (RSIK_RARE(!(m >= 0 && m < kTwoPi))) {  // a / 2pi rounded across an integer (or a is not finite)
        if (m < 0) m += kTwoPi;
        if (m >= kTwoPi) m -= kTwoPi;
    }
    return m;
}
// The serial phases of the continuous mode (one wave alone on its SIMD) are written for that wave's issue rules, measured
// with scripts/probes/issue_probe.hip: every vector instruction costs the wave ~4.5 cycles whether or not it depends on
// the one before, a v_cmp feeding v_cndmask (through vcc or any SGPR pair) costs nothing extra — but a SCALAR instruction
// that reads a mask a vector compare wrote (s_or_b64 / s_and_b64 of two compare results) stalls ~16 cycles, and a branch
// on such a mask ~29.  So on those paths conditions are never combined as masks: selects are chained instead, and
// `opaque` keeps the compiler from folding the chain back into mask algebra.
__device__ __forceinline__ unsigned opaque(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double opaque(double x) { asm volatile("" : "+v"(x)); return x; }
// The fix-up of a modulo result: +2 pi below 0, -2 pi from 2 pi on, else +0.0 — selected as the two words of the ADDEND
// (the low words of +-2 pi are the same, the high words differ in the sign bit): two compares and four 32-bit selects
// (the two conditions exclude each other), then one addition; x + 0.0 is x.
__device__ __forceinline__ double wrap_addend_2pi(double m) {
    constexpr unsigned kLo = 0x54442d18u, kHi = 0x401921fbu;  // 2 pi = 0x401921fb54442d18
    const bool neg = m < 0, big = m >= kTwoPi;
    unsigned lo = opaque(neg ? kLo : 0u), hi = opaque(neg ? kHi : 0u);
    lo = big ? kLo : lo;
    hi = big ? (kHi | 0x80000000u) : hi;
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double pymod_2pi_straight(double a) {
    const double q = floor(a * 0.15915494309189535);
    const double m = fma(-q, kTwoPi, a);
    return m + wrap_addend_2pi(m);
}
// angle_diff (U:486-490) for |a - b| <= 2 pi, i.e. two angles of [-pi, pi]: x = (a - b) + pi lies in [-pi, 3 pi], where
// Python's `x % 2pi` is x, x + 2 pi (x < 0) or x - 2 pi (x >= 2 pi, exact) — no quotient, no floor: bit for bit the
// general form's value.
__device__ __forceinline__ double angle_diff_near(double a, double b) {
    const double x = (a - b) + kPi;
    return (x + wrap_addend_2pi(x)) - kPi;
}
__device__ __forceinline__ double angle_diff_straight(double a, double b) {
    double d = a - b;
    return pymod_2pi_straight(d + kPi) - kPi;
}
// U:486-490
__device__ __forceinline__ double angle_diff(double a, double b) {
    double d = a - b;
    return pymod_2pi(d + kPi) - kPi;
}
// U:468-474
// Interval ends on this path always lie in [-pi, pi] (atan2 results, +-pi for the whole circle, the host-wrapped
// interval limits), where `i0 % 2pi == i1 % 2pi` (U:469) holds exactly when i0 == i1 or the pair is {-pi, pi}
// (-pi + 2pi == pi exactly in binary64): no modulo needed.
__device__ __forceinline__ bool is_valid_angle(double angle, double i0, double i1) {
    if (i0 == i1 || (fabs(i0) == kPi && fabs(i1) == kPi)) return true;
    if (i0 < i1) return (i0 <= angle) && (angle <= i1);
    return (i0 <= angle) || (angle <= i1);
}

// Goal rotation R = Rz(yaw) Ry(pitch) Rx(roll)  (scipy from_euler("xyz"), extrinsic; S:420).
struct Rot {
    double m[9];  // row-major
    __device__ __forceinline__ V3 apply_add(V3 v, V3 t) const {  // R.v + t
        return {fma(m[0], v.x, fma(m[1], v.y, fma(m[2], v.z, t.x))), fma(m[3], v.x, fma(m[4], v.y, fma(m[5], v.z, t.y))),
                fma(m[6], v.x, fma(m[7], v.y, fma(m[8], v.z, t.z)))};
    }
    __device__ __forceinline__ V3 apply_d(V3 v, V3 t) const {  // R.v + t with unfused rounding
        return {m[0] * v.x + m[1] * v.y + m[2] * v.z + t.x, m[3] * v.x + m[4] * v.y + m[5] * v.z + t.y,
                m[6] * v.x + m[7] * v.y + m[8] * v.z + t.z};
    }
    __device__ __forceinline__ V3 col0() const { return {m[0], m[3], m[6]}; }
};
__device__ __forceinline__ Rot rot_from_euler(double roll, double pitch, double yaw) {
    const double ang[3] = {roll, pitch, yaw};
    double sn[3], cs[3];
    fast_sincos_n<3>(ang, sn, cs);  // three angles in lock step
    const double sa = sn[0], ca = cs[0], sb = sn[1], cb = cs[1], sc = sn[2], cc = cs[2];
    Rot r;
    const double ccsb = cc * sb, scsb = sc * sb;
    r.m[0] = cc * cb; r.m[1] = fma(ccsb, sa, -(sc * ca)); r.m[2] = fma(ccsb, ca, sc * sa);
    r.m[3] = sc * cb; r.m[4] = fma(scsb, sa, cc * ca);    r.m[5] = fma(scsb, ca, -(cc * sa));
    r.m[6] = -sb;     r.m[7] = cb * sa;                r.m[8] = cb * ca;
    return r;
}

// utils.get_euler_from_homogeneous_matrix (U:84-90): Rotation.from_matrix(R).as_euler("xyz").  Follows the two
// published algorithms SciPy uses, because for inputs that are not exactly orthonormal, and at gimbal lock, the
// ALGORITHM defines the answer: (0) a matrix whose Gramian M M^T is not the identity (np.isclose, atol 1e-12, default
// rtol 1e-5: 1e-12 off the diagonal) is replaced by the nearest rotation, U V^T of its SVD (orthogonal Procrustes,
// SciPy >= 1.12) — computed here as the polar factor by Newton's iteration X <- (X + X^-T) / 2, which converges
// quadratically to the same matrix; (1) matrix -> quaternion from the largest of (R00, R11, R22, trace) (Markley
// 2008), normalised; (2) quaternion -> extrinsic xyz angles after
// Bernardes & Viollet (2022): with a = w - y, b = x + z, c = y + w, d = z - x the middle angle is
// 2 atan2(|(c, d)|, |(a, b)|) - pi/2 and the outer angles are atan2(b, a) -+ atan2(d, c); within 1e-7 of gimbal lock
// the third angle is set to 0 and the first takes the whole rotation.  Not on the hot path.
// np.isclose(M M^T, I, atol=1e-12) with the default rtol = 1e-5: SciPy's test for "already a rotation"
__device__ __forceinline__ bool gram_is_identity(const double (&m)[9]) {
    // the six distinct entries of M M^T - I; the tolerance of np.isclose is 1e-12 + 1e-5 |I_ij|, i.e. 1e-12 off the
    // diagonal and 1e-5 (+1e-12) on it, four orders of magnitude above any rounding of the products, so the dot
    // products are fused and the two groups are reduced to one comparison each
    auto g = [&](int i, int j) { return fma(m[3 * i], m[3 * j], fma(m[3 * i + 1], m[3 * j + 1], m[3 * i + 2] * m[3 * j + 2])); };
    const double off = fmax(fmax(fabs(g(0, 1)), fabs(g(0, 2))), fabs(g(1, 2)));
    const double dia = fmax(fmax(fabs(g(0, 0) - 1.0), fabs(g(1, 1) - 1.0)), fabs(g(2, 2) - 1.0));
    return off <= 1e-12 && dia <= 1e-12 + 1e-5;
}
__device__ inline void nearest_rotation(double (&m)[9]) {
    for (int it = 0; it < 24; it++) {
        // cofactors: X^-T = cof(X) / det(X)
        const double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
        const double c3 = m[2] * m[7] - m[1] * m[8], c4 = m[0] * m[8] - m[2] * m[6], c5 = m[1] * m[6] - m[0] * m[7];
        const double c6 = m[1] * m[5] - m[2] * m[4], c7 = m[2] * m[3] - m[0] * m[5], c8 = m[0] * m[4] - m[1] * m[3];
        const double idet = 1.0 / (m[0] * c0 + m[1] * c1 + m[2] * c2);
        const double cof[9] = {c0, c1, c2, c3, c4, c5, c6, c7, c8};
        double change = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const double x = 0.5 * (m[k] + cof[k] * idet);
            change = fmax(change, fabs(x - m[k]));
            m[k] = x;
        }
        if (change < 1e-15) break;
    }
}
__device__ inline void euler_xyz_from_matrix(const double (&m_in)[9], double (&eul)[3]) {
    double m[9];
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = m_in[k];
    const bool orth = gram_is_identity(m);
    if (!orth) nearest_rotation(m);
    const double tr = m[0] + m[4] + m[8];
    // pivot = first maximum of (R00, R11, R22, trace), as numpy.argmax picks it
    int pivot = 0;
    double top = m[0];
    if (m[4] > top) { pivot = 1; top = m[4]; }
    if (m[8] > top) { pivot = 2; top = m[8]; }
    if (tr > top) pivot = 3;
    double qx, qy, qz, qw;
    if (pivot == 3) { qx = m[7] - m[5]; qy = m[2] - m[6]; qz = m[3] - m[1]; qw = 1 + tr; }
    else if (pivot == 0) { qx = 1 - tr + 2 * m[0]; qy = m[3] + m[1]; qz = m[6] + m[2]; qw = m[7] - m[5]; }
    else if (pivot == 1) { qy = 1 - tr + 2 * m[4]; qz = m[7] + m[5]; qx = m[1] + m[3]; qw = m[2] - m[6]; }
    else { qz = 1 - tr + 2 * m[8]; qx = m[2] + m[6]; qy = m[5] + m[7]; qw = m[3] - m[1]; }
    const double inv = 1.0 / sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    qx *= inv; qy *= inv; qz *= inv; qw *= inv;
    const double a = qw - qy, b = qx + qz, c = qy + qw, d = qz - qx;
    double mid = 2 * fast_atan2(sqrt(c * c + d * d), sqrt(a * a + b * b));
    const double half_sum = fast_atan2(b, a), half_diff = fast_atan2(d, c);
    double first, third;
    if (fabs(mid) <= 1e-7) { first = 2 * half_sum; third = 0.0; }
    else if (fabs(mid - kPi) <= 1e-7) { first = -2 * half_diff; third = 0.0; }
    else { first = half_sum - half_diff; third = half_sum + half_diff; }
    mid -= kPi / 2;
    eul[0] = first; eul[1] = mid; eul[2] = third;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (eul[k] < -kPi) eul[k] += kTwoPi;
        else if (eul[k] > kPi) eul[k] -= kTwoPi;
    }
}

// Columns of utils.rotation_matrix_from_vector(n) (U:59-81): the rotation taking e_x to u = n/|n|.
// c0 is only needed for the "which side of the wrist-limit plane" tests (frame_c0).
struct Frame {
    V3 c0, c1, c2;
};
// u must already be normalised (the reference divides by the norm first, U:66).  The two colinear special cases
// need |u_y|, |u_z| <~ 1e-8: one rarely-taken branch, kept out of the straight-line code.
// (one max + one compare on the common path: the isclose pair can only hold below 2e-8)
__device__ __forceinline__ bool frame_is_special(V3 u) {
    if (!RSIK_RARE(fmax(fabs(u.y), fabs(u.z)) < 2e-8)) return false;
    return np_isclose(0.0, u.y) && np_isclose(0.0, u.z);
}
__device__ __forceinline__ V3 frame_c0(V3 u) {
    V3 c0 = u;  // R00 = 1 - s^2 (1-c)/s^2 = c to rounding
    if (RSIK_RARE(frame_is_special(u))) {
        if (np_isclose(1.0, u.x)) c0 = {1, 0, 0};
        else if (np_isclose(1.0, -u.x)) c0 = {-1, 0, 0};
    }
    return c0;
}
__device__ __forceinline__ Frame frame_from_unit(V3 u) {
    Frame F;
    // Rodrigues I + K + K^2 (1-c)/s^2 with v = e_x x u = (0, -u_z, u_y)
    // (1 - c) / s^2 = 1 / (1 + c) for a unit vector; well conditioned for the half space u_x > -1 + 1e-8 left by the
    // special cases below
    double h = fast_rcp(1.0 + u.x);
    double yh = u.y * h, zh = u.z * h;
    double yzh = u.y * zh;
    F.c0 = u;
    F.c1 = {-u.y, fma(-u.y, yh, 1.0), -yzh};
    F.c2 = {-u.z, -yzh, fma(-u.z, zh, 1.0)};
    // [D] colinear special cases, numpy.isclose semantics with the tolerance scaled by the 2nd argument
    if (RSIK_RARE(frame_is_special(u))) {
        if (np_isclose(1.0, u.x)) { F.c0 = {1, 0, 0}; F.c1 = {0, 1, 0}; F.c2 = {0, 0, 1}; }
        else if (np_isclose(1.0, -u.x)) { F.c0 = {-1, 0, 0}; F.c1 = {0, 1, 0}; F.c2 = {0, 0, -1}; }
    }
    return F;
}

// Result of the is_reachable stage: what SymbolicIK leaves on `self` for get_joints (Q1).
struct Reach {
    int state;     // RSIK_STATE_*
    bool ok;
    double i0, i1; // theta interval
    double ct0, st0; // cos / sin of i0 (from the intersection point itself, no trigonometric call)
    double ct1, st1; // cos / sin of i1
    V3 pos;        // self.goal_pose[0]
    V3 w;          // self.wrist_position
    V3 c2;         // intersection circle centre
    double r2;     // radius
    V3 n2;         // circle normal (self.intersection_circle[2])
    V3 a1, a2;     // circle frame axes: elbow(theta) = c2 + r2 cos(theta) a1 + r2 sin(theta) a2
    int stage;     // how far self.* was updated: 0 nothing, 1 goal_pose + wrist_position, 2 + intersection_circle
};

template <class Acc>
__device__ __forceinline__ V3 cvec(const Acc& A, int off) {
    return {A(off), A(off + 1), A(off + 2)};
}

// What the path needs from the goal orientation R: R.(-tip_x, tip_y, tip_z) (wrist offset, S:418-425),
// R.(-tip_x, tip_y, 0) (the "tip" point of S:808-812 relative to the goal position) and R's first column (S:839).
// Nine doubles like R itself, but each is only live where it is used (the fused solve kernel parks the last two in
// LDS while the reachability stage runs).
struct Goal {
    V3 woff, toff, xg;
    V3 tw;  // toff - woff = R.(0, 0, -tip_z): the "tip" point of S:808-812 seen from the wrist
};
template <class Acc>
__device__ __forceinline__ Goal make_goal(const Acc& A, const Rot& Rg) {
    const V3 tl = cvec(A, RSIK_C_TIPL);
    Goal g;
    // unfused, left to right: wrist = (R.tl) + pos rounds exactly like the reference's 4x4 product (S:422-424)
    g.woff = {Rg.m[0] * tl.x + Rg.m[1] * tl.y + Rg.m[2] * tl.z, Rg.m[3] * tl.x + Rg.m[4] * tl.y + Rg.m[5] * tl.z,
              Rg.m[6] * tl.x + Rg.m[7] * tl.y + Rg.m[8] * tl.z};
    g.toff = {fma(Rg.m[0], tl.x, Rg.m[1] * tl.y), fma(Rg.m[3], tl.x, Rg.m[4] * tl.y), fma(Rg.m[6], tl.x, Rg.m[7] * tl.y)};
    g.xg = Rg.col0();
    g.tw = {-(Rg.m[2] * tl.z), -(Rg.m[5] * tl.z), -(Rg.m[8] * tl.z)};
    return g;
}
// The same three vectors straight from the Euler angles for a tip offset along the goal z axis only (tip_x = tip_y = 0:
// the reference's default arm and the Reachy 2 URDF): the wrist offset is R's third column times tip_z — bit for bit
// what make_goal computes then, since its other two products are exact zeros — the "tip" point is the goal position
// itself, and R's second column is never formed.  Chosen per launch by the host from the uploaded constants.
template <class Acc>
__device__ __forceinline__ Goal goal_from_euler_tipz(const Acc& A, double roll, double pitch, double yaw) {
    const double ang[3] = {roll, pitch, yaw};
    double sn[3], cs[3];
    fast_sincos_n<3>(ang, sn, cs);
    const double sa = sn[0], ca = cs[0], sb = sn[1], cb = cs[1], sc = sn[2], cc = cs[2];
    const double tz = A(RSIK_C_TIPL + 2);
    Goal g;
    g.woff = {fma(cc * sb, ca, sc * sa) * tz, fma(sc * sb, ca, -(cc * sa)) * tz, (cb * ca) * tz};
    g.toff = {0.0, 0.0, 0.0};
    g.xg = {cc * cb, sc * cb, -sb};
    g.tw = {-g.woff.x, -g.woff.y, -g.woff.z};
    return g;
}
// S:418-425 — wrist = T_torso_goal . (-tip_x, tip_y, tip_z, 1)
__device__ __forceinline__ V3 wrist_position(const V3& woff, V3 pos) { return woff + pos; }

// SymbolicIK.is_reachable (S:121-282) including is_pose_in_robot_reach (S:284-307),
// reduce_goal_pose_no_limits (S:337-349), get_intersection_circle (S:366-399),
// get_limitation_wrist_circle (S:401-416) and are_circles_linked (S:427-568).
// NO_LIMITS = true gives SymbolicIK.is_reachable_no_limits (S:85-119): never fails on reach, interval [-pi, pi].
// KEEP = false (fused kernels that only go on when r.ok): the geometry fields of a failed pose are left unwritten
// instead of being filled with what the reference leaves on `self` (saves the register copies at every early exit).
// reach_impl takes the NO_LIMITS choice as a value: a compile-time constant (BoolC, folded away) for the fused kernels,
// or a per-lane bool for the joint phase of the continuous-mode pipeline, which re-derives the circle of whichever
// variant the step's get_joints follows.  GEOM_ONLY: stop once the intersection circle is known (stage 2) — what
// get_joints needs — for a pose that is known to pass the reach tests before it.
template <bool V>
struct BoolC {
    __device__ __forceinline__ constexpr operator bool() const { return V; }
};
template <bool KEEP, bool GEOM_ONLY, class NL, class Acc>
__device__ Reach reach_impl(const Acc& A, V3 pos_in, const V3 woff, const NL no_limits_v) {
    const bool NO_LIMITS = no_limits_v;
    Reach r;
    r.ok = false;
    r.i0 = r.i1 = __builtin_nan("");
    const V3 s = cvec(A, RSIK_C_SHOULDER);
    const double u = A(RSIK_C_UPPER_ARM), f = A(RSIK_C_FOREARM);
    const double bl = A(RSIK_C_BACKWARD), pm = A(RSIK_C_PROJ_MARGIN), upf = A(RSIK_C_UPF);

    // [D] S:284-307
    V3 gp = pos_in;
    V3 dv = gp - s;
    // |dv| > max_arm_length decided on the squared length against the exact threshold (RSIK_C_MAX_LEN_SQ): the
    // square root is only paid by poses that are projected back
    const double ss0 = dot_d(dv, dv);
    int st = RSIK_STATE_REACHABLE;
    if (RSIK_RARE(ss0 > A(RSIK_C_MAX_LEN_SQ))) {
        double nd = sqrt_cr(ss0) + pm;
        gp = madd(dv * fast_rcp(nd), A(RSIK_C_MAX_LEN), s);
        st = RSIK_STATE_POSE_OUT_OF_REACH;
    }
    if (RSIK_RARE(gp.x < bl)) {
        gp.x = bl;
        st = RSIK_STATE_BACKWARD_POSE;
    }
    r.state = st;
    r.stage = 0;
    if (KEEP) { r.pos = gp; r.w = gp; r.c2 = gp; r.n2 = gp; r.a1 = gp; r.a2 = gp; r.r2 = 0.0; r.ct0 = 1.0; r.st0 = 0.0; r.ct1 = 1.0; r.st1 = 0.0; }
    if (!NO_LIMITS && st != RSIK_STATE_REACHABLE) return r;
    r.stage = 1;

    RSIK_MARK("reach_wrist");
    V3 w = wrist_position(woff, gp);
    // [D] S:146-153 / S:94-98
    if (RSIK_RARE(w.x < bl)) {
        double diff = bl - w.x;
        gp.x = gp.x + diff;
        if (NO_LIMITS) w = wrist_position(woff, gp);
        else w.x = w.x + diff;
    }
    V3 P = w - s;
    double dsw, inv_d;  // get_intersection_circle recomputes the same norm (S:373); only the rare branches change it
    sqrt_rsqrt(dot_d(P, P), dsw, inv_d);
    double d = dsw;
    V3 self_pos = gp;  // what ends up in self.goal_pose (differs from the local only in the NO_LIMITS far case, Q4)
    if (NO_LIMITS) {
        if (RSIK_RARE(dsw > upf)) {  // S:102-105: self.wrist_position moved onto the sphere, self.goal_pose shifted
            double nd = fabs(dsw) + pm;
            V3 nw = madd((w - s) * fast_rcp(nd), upf, s);
            self_pos = gp + (nw - w);
            w = nw;
            P = w - s;
            sqrt_rsqrt(dot_d(P, P), d, inv_d);
        }
    } else {
        if (RSIK_RARE(dsw > upf)) {  // [D] S:157-161
            r.state = RSIK_STATE_WRIST_OUT_OF_RANGE;
            if (KEEP) { r.pos = gp; r.w = w; }
            return r;
        }
    }
    if (RSIK_RARE(dsw < A(RSIK_C_MIN_DIST))) {  // [D] S:166-171 / S:107-112
        // the wrist is pushed out radially to the minimum distance: new wrist = s + P k, k = d_min / (|P| + margin)
        // (this branch runs in nearly every wave: 8 % of the reachable poses of a random batch sit at the elbow limit.
        // The reference moves the goal by new_wrist - wrist and recomputes the wrist from it (S:168-170): goal and wrist
        // both move by P (k - 1), the new P is P k and |P| = |P_old| k up to rounding — no second square root, one
        // reciprocal: 1 / k = (|P| + margin) / d_min.)
        const double t = fabs(dsw) + pm;
        const double k = A(RSIK_C_MIN_DIST) * fast_rcp(t);
        const double km1 = k - 1.0;
        gp = madd(P, km1, gp);
        w = madd(P, km1, w);
        self_pos = gp;
        P = P * k;
        d = dsw * k;
        inv_d = inv_d * (t * A(RSIK_C_INV_MIN_DIST));
    }

    RSIK_MARK("reach_circle");
    // S:366-399 intersection circle of the shoulder sphere (radius u) and the wrist sphere (radius f)
    if (RSIK_RARE(d > upf)) {  // [D] S:374
        r.state = RSIK_STATE_SHOULD_NOT_HAPPEN;
        if (KEEP) { r.pos = self_pos; r.w = w; }
        return r;
    }
    r.pos = self_pos;
    r.w = w;
    V3 n2 = P * inv_d;
    double r2, ir2;  // radius and its reciprocal (0 for the degenerate circle)
    double kk;
    V3 c2;
    {
        double d2 = d * d, k = d2 - f * f + u * u;
        // [D] the radicand is exactly 0 for a fully extended arm (Q23) and must not become -1e-18 through an fma
        double rad = 4 * d2 * (u * u) - k * k;
        // radius = sqrt(rad) / (2 d) and its reciprocal from ONE reciprocal square root (sqrt(rad) = rad / sqrt(rad) to
        // 1 - 2 ulp; the radius is not a decision value)
        const double irad = (rad != 0.0) ? rsqrt_fast(rad) : 0.0;
        const double hid = 0.5 * inv_d;
        r2 = hid * (rad * irad);
        ir2 = (d + d) * irad;
        kk = k * hid;  // distance of the circle's centre from the shoulder along n2
        c2 = s + n2 * kk;
    }
    RSIK_MARK("reach_frame");
    Frame F2 = frame_from_unit(n2);
    r.c2 = c2; r.r2 = r2; r.n2 = n2; r.a1 = F2.c1; r.a2 = F2.c2;
    r.stage = 2;
    if (NO_LIMITS || GEOM_ONLY) {
        r.ok = true; r.state = RSIK_STATE_REACHABLE; r.i0 = -kPi; r.i1 = kPi;
        r.ct0 = -1.0; r.st0 = -1.2246467991473532e-16;  // cos(-pi), sin(-pi) as np.cos/np.sin return them
        r.ct1 = -1.0; r.st1 = 1.2246467991473532e-16;
        return r;
    }

    RSIK_MARK("reach_limit_side");
    // S:401-416 wrist-limit circle (cone of half-angle wrist_limit around the hand axis).
    // |w - goal| = |tip| by construction, so the unit normal costs one multiply by a constant.
    // wrist - goal is the wrist offset R.tip_local itself (every shift above moved both points together).
    V3 N1 = woff * A(RSIK_C_INV_GRIP);
    double r1 = A(RSIK_C_WRIST_R);

    // S:427-509 are_circles_linked, wrist-centred coordinates: p1 = c1 - w = N1 * axial offset
    V3 p1 = N1 * A(RSIK_C_WRIST_AX), p2 = c2 - w;
    const V3 f1 = frame_c0(N1);
    // [D] x of T_limitation_torso . p = c0.p + (-c0).p1
    const double tlx = -dot_d(f1, p1);
    const double side_val = dot_d(f1, p2) + tlx;
    bool side_ok = side_val > 0;
    r.state = side_ok ? RSIK_STATE_REACHABLE : RSIK_STATE_LIMITED_BY_WRIST;
    r.ok = side_ok;
    // the circles do not cross: whole circle [-pi, pi] or nothing, decided by the side (S:487-509).  Only the early
    // exits pay for these values.
    auto whole_or_nothing = [&]() {
        r.i0 = side_ok ? -kPi : __builtin_nan("");
        r.i1 = side_ok ? kPi : __builtin_nan("");
        r.ct0 = -1.0; r.st0 = -1.2246467991473532e-16;  // cos(-pi), sin(-pi) as np.cos/np.sin return them
        r.ct1 = -1.0; r.st1 = 1.2246467991473532e-16;
    };

    RSIK_MARK("reach_line");
    const V3 N2 = n2;  // already unit (the reference renormalises: a 1-ulp no-op)
    // Closed form of S:427-568 for the generic case.  A point of the elbow circle, e(theta) - w = p2 + r2 (a1 cos + a2 sin),
    // is on the allowed side of the wrist-limit plane iff N1.(e - w) > N1.p1 (the test the reference applies to its
    // mid-angle point, S:564), i.e. iff  A' cos(theta) + B' sin(theta) > D'  with A' = N1.a1, B' = N1.a2,
    // D' = (N1.p1 - N1.p2) / r2 = -side_val / r2: one arc, centred at phi = atan2(B', A'), of half-width
    // alpha = acos(D' / R'), R' = |(A', B')| = |N1 x N2|.  The interval is [phi - alpha, phi + alpha]; both ends are formed
    // as unit vectors (angle addition) and go through the table atan2, which also delivers them in (-pi, pi] so that
    // interval[0] > interval[1] means wrap-around exactly as in the reference.  No plane-plane line, no circle-line
    // intersection, no mid-point test: ~70 instead of ~135 fp64 operations.
    // The reference's own arithmetic (below) still decides everything that is decided by rounding: planes parallel to
    // 1e-6 (R'^2 < 1e-12), tangency (|R'^2 - D'^2| < 1e-8), the degenerate circle (r2 = 0) and the neighbourhood of its
    // isclose(t1, t0) early exit (Q7; t0 |N1 x N2| = N2.b, t1 |N1 x N2| = N1.b = side_val).
    {
        const double Ap = dot(N1, F2.c1), Bp = dot(N1, F2.c2);
        const double R2 = fma(Ap, Ap, Bp * Bp);
        const double Dp = -side_val * ir2;
        const double disc = fma(-Dp, Dp, R2);
        // N2.p2 = N2.(c2 - w) = kk - d  (c2 - s = kk n2, w - s = d n2); this value only feeds the guard below
        const double n2b = (kk - d) - A(RSIK_C_WRIST_AX) * dot(N1, N2);
        const bool exact = (ir2 == 0.0) || (R2 < 1e-12) || (fabs(disc) < 1e-8) ||
                           (fabs(side_val - n2b) <= 2e-8 + 2e-5 * fabs(n2b));
        if (!RSIK_RARE(exact)) {
            if (disc < 0) { whole_or_nothing(); return r; }
            r.ok = true;
            r.state = RSIK_STATE_REACHABLE;
            const double iR = rsqrt_fast(R2);
            const double cphi = Ap * iR, sphi = Bp * iR;
            const double cal = Dp * iR, sal = (disc * rsqrt_fast(disc)) * iR;
            const double c0 = fma(cphi, cal, sphi * sal), s0 = fma(sphi, cal, -(cphi * sal));
            const double c1 = fma(cphi, cal, -(sphi * sal)), s1 = fma(sphi, cal, cphi * sal);
            const double ss[2] = {s0, s1}, cc[2] = {c0, c1};
            double aa[2];
            unit_atan2_n<2>(A.utab, ss, cc, aa);
            r.i0 = aa[0]; r.i1 = aa[1];
            r.ct0 = c0; r.st0 = s0;
            r.ct1 = c1; r.st1 = s1;
            return r;
        }
    }
    const double mg = A(RSIK_C_NORMAL_MARGIN);
    // S:588-606 + S:570-586: line of intersection of the two planes.  The reference solves
    // [v1, -v2] t = p2 - p1 by least squares; since v1, v2, (p2-p1 minus its v-part) are coplanar the
    // minimiser is the exact intersection: t0 = N2.b / (N2.v1), t1 = N1.b / -(N1.v2), both denominators = |N1 x N2|.
    V3 cr = cross(N1, N2);
    const double crcr = dot(cr, cr);
    // [D] S:475-483 parallel planes: all three |N2 -+ N1| components below the margin.  That forces
    // |N1 x N2|^2 <= 3 margin^2, so the six-compare test is only evaluated below that bound.
    if (RSIK_RARE(crcr < 4.0 * (mg * mg) + 1e-30)) {
        bool par = (fabs(N2.x - N1.x) < mg && fabs(N2.y - N1.y) < mg && fabs(N2.z - N1.z) < mg) ||
                   (fabs(N2.x + N1.x) < mg && fabs(N2.y + N1.y) < mg && fabs(N2.z + N1.z) < mg);
        if (par) { whole_or_nothing(); return r; }
    }
    const double inv_nv = rsqrt_fast(crcr);
    V3 v = cr * inv_nv;
    V3 v1 = cross(v, N1);
    V3 b = p2 - p1;
    double t0 = dot(N2, b) * inv_nv;
    double t1 = dot(N1, b) * inv_nv;
    if (RSIK_RARE(np_isclose(t1, t0))) { whole_or_nothing(); return r; }  // [D] S:582-583 (Q7)
    V3 q = madd(v1, t0, p1);

    RSIK_MARK("reach_circle_line");
    // S:608-645 circle 1 (centre p1, radius r1) with the line (q, v); S:511-568 angles of the intersection points
    // in the circle-2 frame.  q - p1 = t0 v1 with v, v1 orthonormal, so the reference's quadratic
    // a t^2 + b t + c (a = |v|^2, b = 2 v.(q - p1), c = |q - p1|^2 - r1^2) is t^2 = r1^2 - t0^2 up to rounding.
    // [D] Its discriminant decides reachable / tangent / limited: within 1e-9 r1^2 of zero the reference's own
    // arithmetic is evaluated (rare branch), elsewhere both agree on the sign and the roots to O(1e-16).
    V3 a1 = F2.c1, a2 = F2.c2;
    const double r1sq = r1 * r1;
    const double disc4 = fma(-t0, t0, r1sq);
    double ly1, lz1, ly2, lz2;
    double by, bz;  // foot of the chord between the two points (their mean), in the unit circle-2 frame
    if (RSIK_RARE(fabs(disc4) < 1e-9 * r1sq)) {
        V3 wv = q - p1;
        const double qa = v.x * v.x + v.y * v.y + v.z * v.z;
        const double qb = 2 * (v.x * wv.x + v.y * wv.y + v.z * wv.z);
        const double qc = (wv.x * wv.x + wv.y * wv.y + wv.z * wv.z) - r1 * r1;
        const double disc = qb * qb - 4 * qa * qc;
        if (disc < 0) { whole_or_nothing(); return r; }  // [D]
        const double oy = -dot(a1, p2), oz = -dot(a2, p2);  // translation of T_intersection_torso
        r.ok = true;
        r.state = RSIK_STATE_REACHABLE;
        const double inv_2qa = fma(-0.5, qa, 1.0);  // 1 / (2 qa) for qa = |v|^2 = 1 + O(1e-16)
        if (disc == 0) {  // [D] tangent: interval [a, a] (Q8)
            double t = -qb * inv_2qa;
            V3 p = madd(v, t, q);
            double ly = dot(a1, p) + oy, lz = dot(a2, p) + oz;
            double ang = fast_atan2(lz, ly);
            double il = rsqrt_fast(ly * ly + lz * lz);
            r.i0 = ang; r.i1 = ang; r.ct0 = ly * il; r.st0 = lz * il; r.ct1 = r.ct0; r.st1 = r.st0;
            return r;
        }
        double sq = sqrt_cr(disc);
        double ta = (-qb + sq) * inv_2qa, tb = (-qb - sq) * inv_2qa;
        V3 pa = madd(v, ta, q), pb = madd(v, tb, q);
        ly1 = (dot(a1, pa) + oy) * ir2; lz1 = (dot(a2, pa) + oz) * ir2;
        ly2 = (dot(a1, pb) + oy) * ir2; lz2 = (dot(a2, pb) + oz) * ir2;
        by = 0.5 * (ly1 + ly2); bz = 0.5 * (lz1 + lz2);
    } else {
        if (disc4 < 0) { whole_or_nothing(); return r; }
        r.ok = true;
        r.state = RSIK_STATE_REACHABLE;
        // points q +- h v (the + root first, S:640-644), seen from the circle-2 centre and scaled to the unit circle:
        // both lie on circle 2 (the two circles share the wrist sphere), so (ly, lz) are unit vectors
        const double hr = (disc4 * rsqrt_fast(disc4)) * ir2;
        const V3 dq = q - p2;
        by = dot(a1, dq) * ir2; bz = dot(a2, dq) * ir2;
        const double vy = dot(a1, v) * hr, vz = dot(a2, v) * hr;
        ly1 = by + vy; lz1 = bz + vz;
        ly2 = by - vy; lz2 = bz - vz;
    }
    RSIK_MARK("reach_atan2x2");
    double ang1, ang2;
    {
        const double ss[2] = {lz1, lz2}, cc[2] = {ly1, ly2};
        double aa[2];
        unit_atan2_n<2>(A.utab, ss, cc, aa);
        ang1 = aa[0]; ang2 = aa[1];
    }
    RSIK_MARK("reach_mid_select");
    // S:548-566: the sorted pair [lo, hi] is the interval when the mid-angle point lies on the allowed side of the
    // wrist-limit plane, [hi, lo] otherwise.  The mid angle is symmetric in the two points, so no sort is needed:
    // interval[0] belongs to point 1 exactly when (inside != (ang2 < ang1)).
    // The point at the mean angle needs no sin/cos: the two points are unit vectors b +- v' with b their chord foot, so
    // the bisector of the arc between them is b / |b|, and the mean of two angles in (-pi, pi] is the bisector of the arc
    // that does not cross +-pi, i.e. -b / |b| when the angles are more than pi apart.  (|b| ~ 0: the points are
    // diametrically opposite and the direction is the mean angle's own, rare branch.)
    double sm, cm;
    const double bb = fma(by, by, bz * bz);
    if (RSIK_RARE(bb < 1e-12)) {
        fast_sincos((ang1 + ang2) / 2, &sm, &cm);
    } else {
        double ib = rsqrt_fast(bb);
        ib = (fabs(ang1 - ang2) > kPi) ? -ib : ib;
        cm = by * ib; sm = bz * ib;
    }
    // [D] S:564: x of T_limitation_torso . (p2 + r2 (a1 cos + a2 sin)) > 0; the p2 part is the side test's value
    const double fa1 = dot(f1, a1), fa2 = dot(f1, a2);
    const bool inside = fma(r2, fma(cm, fa1, sm * fa2), side_val) > 0;
    const bool first = inside != (ang2 < ang1);
    r.i0 = first ? ang1 : ang2;
    r.i1 = first ? ang2 : ang1;
    r.ct0 = first ? ly1 : ly2;
    r.st0 = first ? lz1 : lz2;
    r.ct1 = first ? ly2 : ly1;
    r.st1 = first ? lz2 : lz1;
    return r;
}

template <bool NO_LIMITS, bool KEEP = true, class Acc>
__device__ __forceinline__ Reach reach_g(const Acc& A, V3 pos_in, const V3 woff) {
    return reach_impl<KEEP, false>(A, pos_in, woff, BoolC<NO_LIMITS>{});
}

template <bool NO_LIMITS, bool KEEP = true, class Acc>
__device__ __forceinline__ Reach reach(const Acc& A, V3 pos_in, const Rot& Rg) {
    return reach_g<NO_LIMITS, KEEP>(A, pos_in, make_goal(A, Rg).woff);
}

// S:684-695
__device__ __forceinline__ V3 elbow_on_circle(const Reach& r, double ct, double st) {
    double y = r.r2 * ct, z = r.r2 * st;
    return madd(r.a1, y, madd(r.a2, z, r.c2));
}

// [D] S:708-713 / U:459-464: elbow above the singularity plane
// EXACT = false (the fused kernels): one fused multiply-add against the host-side constant RSIK_C_PLANE_K.  The
// projection this test triggers is continuous across the plane (a point on the plane is its own projection), so the
// rounding of the test is immaterial there; the stored-state path keeps the reference's operation order.
template <bool EXACT = true, class Acc>
__device__ __forceinline__ bool above_singularity_plane(const Acc& A, V3 e) {
    if constexpr (EXACT) return e.z > (e.x - A(RSIK_C_ES)) * A(RSIK_C_SING_COEFF) + A(RSIK_C_ES + 2) - A(RSIK_C_SING_OFFSET);
    else return e.z > fma(e.x, A(RSIK_C_SING_COEFF), A(RSIK_C_PLANE_K));
}
// U:443-465 (effective predicate, Q10).  PLANE = false: the launch's host code has shown that the singularity-plane half
// holds for every point a shoulder-centred sphere of radius u can reach (the non-DVT offset, Q18), so only the
// elbow-side half is evaluated.
template <bool PLANE = true, class Acc>
__device__ __forceinline__ bool is_elbow_ok(const Acc& A, V3 e) {
    bool ok = e.y * A(RSIK_C_SIDE) < -0.2;
    if constexpr (PLANE) ok = ok && (e.z < (e.x - A(RSIK_C_ES)) * A(RSIK_C_SING_COEFF) + A(RSIK_C_ES + 2) - A(RSIK_C_SING_OFFSET));
    return ok;
}

struct JointsOut {
    double j[7];
    V3 elbow;
    bool projected;
    bool sing;  // an exact singularity fell back to previous_joints[0] / [2] (S:751-753, 782-784)
    // unit vectors of the wrist angles, for ControlIK.safety_checks without re-evaluating sin/cos
    double c4, s4, c5, s5, c6, s6;
};

// SymbolicIK.get_joints (S:697-863).  Mutates r.pos / r.w like the reference mutates self (Q1).
//
// Derivation (see DESIGN.md "joint chain"): with q = M_shoulder_torso.(e - s),
//   G = Rz(-sr) Ry(-sp) has rows  q/|q|, (-q_y c, rho/|q|, -q_y s)/... , (-s, 0, c)   with (c, s) = (q_x, q_z)/rho
// so no trigonometric function of a computed joint angle is ever evaluated; the same holds for the
// elbow (H) and wrist (K) frames.  Exact-zero singularities fall back to previous_joints (S:751-753, 782-784).
// FRESH = true: the state comes straight from reach() on the same pose (the fused kernels), so |wrist - elbow| is
// the forearm length by construction; FRESH = false (stored solver state, possibly moved by an earlier projection,
// Q1) measures it.
template <bool FRESH, bool TIPZ = false, class Acc>
__device__ JointsOut joints_from_theta_g(const Acc& A, Reach& r, const Goal& G, double ct, double st, const double* prev) {
    RSIK_MARK("joints_elbow");
    JointsOut o;
    V3 e = elbow_on_circle(r, ct, st);
    o.projected = false;
    if (RSIK_RARE(above_singularity_plane<!FRESH>(A, e))) {  // S:708-718 -> make_elbow_projection S:647-682
        // project the elbow onto the plane and snap it to the circle (centre pc, in the plane) cut out of the shoulder sphere:
        // the in-plane vector from pc is measured directly (the plane point of S:657 is only needed for the distance, and
        // pc - P_limits is orthogonal to the normal)
        const V3 v3 = cvec(A, RSIK_C_PLANE_N), pc = cvec(A, RSIK_C_PROJ_CENTER);
        const V3 d1 = e - pc;
        const V3 V = madd(v3, -dot(d1, v3), d1);
        const V3 ne = madd(V, A(RSIK_C_PROJ_RADIUS) * rsqrt_fast(dot(V, V)), pc);
        const V3 shift = ne - e;
        e = ne;
        // S:718 recomputes the wrist from the moved goal; it is the old wrist moved by the same vector (to rounding).
        // FRESH: the moved goal position itself is not consumed again (the tip is taken relative to the wrist)
        if constexpr (FRESH) {
            r.w = r.w + shift;
        } else {
            r.pos = r.pos + shift;
            r.w = wrist_position(G.woff, r.pos);
        }
        o.projected = true;
    }
    RSIK_MARK("joints_shoulder");
    o.elbow = e;
    const double u = A(RSIK_C_UPPER_ARM), f = A(RSIK_C_FOREARM);
    // shoulder frame: x = M_shoulder_torso . p + P_shoulder_torso (S:728-741)
    auto to_shoulder = [&](V3 p) -> V3 {
        return {fma(A(RSIK_C_MST + 0), p.x, fma(A(RSIK_C_MST + 1), p.y, fma(A(RSIK_C_MST + 2), p.z, A(RSIK_C_TSH + 0)))),
                fma(A(RSIK_C_MST + 3), p.x, fma(A(RSIK_C_MST + 4), p.y, fma(A(RSIK_C_MST + 5), p.z, A(RSIK_C_TSH + 1)))),
                fma(A(RSIK_C_MST + 6), p.x, fma(A(RSIK_C_MST + 7), p.y, fma(A(RSIK_C_MST + 8), p.z, A(RSIK_C_TSH + 2))))};
    };
    V3 q = to_shoulder(e);
    // shoulder pitch / roll (S:751-766)
    // The seven joint angles are pure outputs (no rotation below is built from an angle): they are evaluated together
    // at the end from the normalised direction vectors.
    double cphi, sphi, rho;
    const bool sing_sp = RSIK_RARE(q.x == 0 && q.z == 0);
    if (sing_sp) {  // [D] exact singularity: keep the previous pitch
        double s_, c_;
        fast_sincos(prev[0], &s_, &c_);
        cphi = c_; sphi = -s_; rho = 0.0;
    } else {
        const double rho2 = fma(q.x, q.x, q.z * q.z);
        const double irho = rsqrt_fast(rho2);
        rho = rho2 * irho;
        cphi = q.x * irho; sphi = q.z * irho;
    }
    const double iL = A(RSIK_C_INV_U);  // |e - shoulder| = upper arm length by construction
    double cr = rho * iL, srs = q.y * iL;
    // G = Rz(-sr) Ry(-sp): rows g0, g1, g2
    V3 g0 = {cr * cphi, srs, cr * sphi};
    V3 g1 = {-srs * cphi, cr, -srs * sphi};
    V3 g2 = {-sphi, 0.0, cphi};
    // (g2 and h1 have a structural zero: their products are written out, a literal 0.0 operand is not folded away)
    auto rot_g = [&](V3 a) -> V3 { return {dot(g0, a), dot(g1, a), fma(g2.x, a.x, g2.z * a.z)}; };
    auto to_elbow = [&](V3 p) -> V3 {  // T_elbow_torso (S:776-777)
        V3 a = rot_g(to_shoulder(p));
        a.x -= u;
        return a;
    };
    RSIK_MARK("joints_elbow_angles");
    // elbow yaw / pitch (S:780-797)
    V3 pw = to_elbow(r.w);
    double sigma, ca, sa;
    const bool sing_ey = RSIK_RARE(pw.y == 0 && pw.z == 0);
    if (sing_ey) {  // [D] exact singularity
        fast_sincos(prev[2], &sa, &ca);
        sigma = 0.0;
    } else {
        const double sig2 = fma(pw.y, pw.y, pw.z * pw.z);
        const double isig = rsqrt_fast(sig2);
        sigma = sig2 * isig;
        ca = pw.z * isig; sa = pw.y * isig;
    }
    const double ilam = FRESH ? A(RSIK_C_INV_F) : rsqrt_fast(fma(sigma, sigma, pw.x * pw.x));
    double cchi = pw.x * ilam, schi = sigma * ilam;
    // H = Ry(-ep) Rx(ey)
    V3 h0 = {cchi, schi * sa, schi * ca};
    V3 h1 = {0.0, ca, -sa};
    V3 h2 = {-schi, cchi * sa, cchi * ca};
    auto rot_h = [&](V3 a) -> V3 { return {dot(h0, a), fma(h1.y, a.y, h1.z * a.z), dot(h2, a)}; };
    RSIK_MARK("joints_wrist");
    // wrist roll / pitch (S:808-826): the "tip" point of S:808-812 in the wrist frame
    V3 t;
    if constexpr (FRESH) {
        // the wrist is the origin of the wrist frame (T_wrist_torso . wrist = 0 by construction of G and H), so only the
        // three rotations act on tip' - wrist = R.(0, 0, -tip_z) (Goal::tw; a projection above moved both points alike)
        const V3 vs = {dot(cvec(A, RSIK_C_MST + 0), G.tw), dot(cvec(A, RSIK_C_MST + 3), G.tw), dot(cvec(A, RSIK_C_MST + 6), G.tw)};
        t = rot_h(rot_g(vs));
    } else {
        V3 a = to_elbow(TIPZ ? r.pos : (G.toff + r.pos));  // T_wrist_torso (S:805-806)
        t = rot_h(a);
        t.x -= f;
    }
    double tau, cw, sw, wr_zero = 0.0;
    const bool tau_zero = RSIK_RARE(t.x == 0 && t.y == 0);
    if (tau_zero) {
        wr_zero = kPi - fast_atan2(t.y, -t.x);  // +-0 arguments: 0 or pi like the C library
        double w0 = wr_zero > kPi ? wr_zero - kTwoPi : wr_zero;
        fast_sincos(w0, &sw, &cw);
        tau = 0.0;
    } else {
        const double tau2 = fma(t.x, t.x, t.y * t.y);
        const double itau = rsqrt_fast(tau2);
        tau = tau2 * itau;
        cw = t.x * itau; sw = t.y * itau;
    }
    const double imu = FRESH ? A(RSIK_C_INV_TIPZ) : rsqrt_fast(fma(tau, tau, t.z * t.z));  // |tip' - wrist| = |tip_z|
    double cp = tau * imu, spp = t.z * imu;
    // K = Ry(wp) Rz(-wr); only rows 1, 2 are needed for the yaw
    V3 k1 = {-sw, cw, 0.0};
    V3 k2 = {-spp * cw, -spp * sw, cp};
    RSIK_MARK("joints_yaw");
    // wrist yaw (S:839-848): direction of the goal frame's x axis seen from the tip frame
    V3 xg = G.xg;
    V3 xs = {dot(cvec(A, RSIK_C_MST + 0), xg), dot(cvec(A, RSIK_C_MST + 3), xg), dot(cvec(A, RSIK_C_MST + 6), xg)};
    V3 xw = rot_h(rot_g(xs));
    double gy = fma(k1.x, xw.x, k1.y * xw.y), gz = dot(k2, xw);
    RSIK_MARK("joints_atan2x7");
    // All seven angles are directions of normalised vectors: unit_atan2_n (no division), seven in lock step.
    // (gy, gz) is the goal x axis seen in the plane normal to the tip axis.  With a tip offset along the goal z axis
    // (TIPZ) the tip axis IS the goal z axis, the goal x axis is orthogonal to it and (gy, gz) is a unit vector already.
    double c6 = gz, s6 = gy;
    if (!TIPZ) {
        const double ign = rsqrt_fast(fma(gy, gy, gz * gz));
        c6 = gz * ign; s6 = gy * ign;
    }
    double at[7];
    {
        const double us[7] = {sphi, srs, ca, schi, sw, spp, s6};
        const double uc[7] = {cphi, cr, -sa, cchi, cw, cp, c6};
        unit_atan2_n<7>(A.utab, us, uc, at);
    }
    RSIK_MARK("joints_out");
    // wrist roll (S:813-816): pi - atan2(t_y, -t_x) wrapped into (-pi, pi] is atan2(t_y, t_x)
    o.j[0] = -at[0];
    o.j[1] = at[1];
    o.j[2] = -kPi / 2 + at[2];
    o.j[3] = fmin(fmax(-at[3], -A(RSIK_C_ELBOW_LIMIT)), A(RSIK_C_ELBOW_LIMIT));  // S:853-861
    o.j[4] = at[4];
    o.j[5] = -at[5];
    o.j[6] = at[6];
    if (RSIK_RARE(sing_sp || sing_ey || tau_zero)) {  // exact singularities keep the previous / C-library values
        if (sing_sp) o.j[0] = prev[0];
        if (sing_ey) o.j[2] = prev[2];
        if (tau_zero) o.j[4] = wr_zero > kPi ? wr_zero - kTwoPi : wr_zero;
    }
    o.sing = sing_sp || sing_ey;
    o.c4 = cw; o.s4 = sw; o.c5 = cp; o.s5 = -spp;
    o.c6 = c6; o.s6 = s6;
    return o;
}

template <bool FRESH, class Acc>
__device__ __forceinline__ JointsOut joints_from_theta(const Acc& A, Reach& r, const Rot& Rg, double ct, double st,
                                                       const double* prev) {
    return joints_from_theta_g<FRESH>(A, r, make_goal(A, Rg), ct, st, prev);
}

// Forward kinematics of the arm (the chain get_joints inverts, S:728-848):
//   T_torso_tip = T(s) Ms Ry(j0) Rz(j1) Tx(u) Rx(-j2) Ry(j3) Tx(f) Rz(j4) Ry(j5),  Ms = M_shoulder_torso^T,
//   goal axes in the tip frame: z_goal = -x_tip, x_goal = (0, sin j6, cos j6);  goal = wrist - R_goal . tip_local.
// A self-contained check on the device: FK(IK(pose)) == pose.
struct FkOut {
    V3 pos;
    double R[9];  // goal rotation, row-major
};
template <class Acc>
__device__ FkOut forward_kinematics(const Acc& A, const double (&j)[7]) {
    double sn[7], cs[7];
    {
        const double a[4] = {j[0], j[1], j[2], j[3]}, b[3] = {j[4], j[5], j[6]};
        double s1[4], c1[4], s2[3], c2[3];
        fast_sincos_n<4>(a, s1, c1);
        fast_sincos_n<3>(b, s2, c2);
#pragma unroll
        for (int k = 0; k < 4; k++) { sn[k] = s1[k]; cs[k] = c1[k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { sn[4 + k] = s2[k]; cs[4 + k] = c2[k]; }
    }
    auto mul = [](const double (&X)[9], const double (&Y)[9], double (&Z)[9]) {
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Z[3 * r + c] = fma(X[3 * r], Y[c], fma(X[3 * r + 1], Y[3 + c], X[3 * r + 2] * Y[6 + c]));
    };
    // Gt = Ry(j0) Rz(j1), Ht = Rx(-j2) Ry(j3), Kt = Rz(j4) Ry(j5)
    const double c0 = cs[0], s0 = sn[0], c1 = cs[1], s1 = sn[1], c2 = cs[2], s2 = sn[2], c3 = cs[3], s3 = sn[3];
    const double c4 = cs[4], s4 = sn[4], c5 = cs[5], s5 = sn[5];
    const double Gt[9] = {c0 * c1, -c0 * s1, s0, s1, c1, 0.0, -s0 * c1, s0 * s1, c0};
    // Rx(-j2) = [[1,0,0],[0,c2,s2],[0,-s2,c2]];  Ry(j3) = [[c3,0,s3],[0,1,0],[-s3,0,c3]]
    const double H[9] = {c3, 0.0, s3, -s2 * s3, c2, s2 * c3, -c2 * s3, -s2, c2 * c3};
    // Rz(j4) = [[c4,-s4,0],[s4,c4,0],[0,0,1]];  Ry(j5) = [[c5,0,s5],[0,1,0],[-s5,0,c5]]
    const double Kt[9] = {c4 * c5, -s4, c4 * s5, s4 * c5, c4, s4 * s5, -s5, 0.0, c5};
    double Ms[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Ms[3 * r + c] = A(RSIK_C_MST + 3 * c + r);
    double MG[9], MGH[9], Rt[9];
    mul(Ms, Gt, MG);
    mul(MG, H, MGH);
    mul(MGH, Kt, Rt);
    const double u = A(RSIK_C_UPPER_ARM), f = A(RSIK_C_FOREARM);
    // wrist = s + MG . (u e_x + H . f e_x) = s + u MG[:,0] + f MGH[:,0]
    const V3 sh = cvec(A, RSIK_C_SHOULDER);
    const V3 w = {fma(u, MG[0], fma(f, MGH[0], sh.x)), fma(u, MG[3], fma(f, MGH[3], sh.y)), fma(u, MG[6], fma(f, MGH[6], sh.z))};
    // goal axes: x = Rt . (0, s6, c6), z = -Rt[:,0], y = z x x
    const double s6 = sn[6], c6 = cs[6];
    const V3 xg = {fma(Rt[1], s6, Rt[2] * c6), fma(Rt[4], s6, Rt[5] * c6), fma(Rt[7], s6, Rt[8] * c6)};
    const V3 zg = {-Rt[0], -Rt[3], -Rt[6]};
    const V3 yg = cross(zg, xg);
    FkOut o;
    o.R[0] = xg.x; o.R[1] = yg.x; o.R[2] = zg.x;
    o.R[3] = xg.y; o.R[4] = yg.y; o.R[5] = zg.y;
    o.R[6] = xg.z; o.R[7] = yg.z; o.R[8] = zg.z;
    const V3 tl = cvec(A, RSIK_C_TIPL);
    o.pos = {w.x - fma(o.R[0], tl.x, fma(o.R[1], tl.y, o.R[2] * tl.z)), w.y - fma(o.R[3], tl.x, fma(o.R[4], tl.y, o.R[5] * tl.z)),
             w.z - fma(o.R[6], tl.x, fma(o.R[7], tl.y, o.R[8] * tl.z))};
    return o;
}

// U:93-112 limit_theta_to_interval (previous_theta is normalised by the reference but never used, Q12)
__device__ __forceinline__ double limit_theta_to_interval(double theta, double l0, double l1) {
    theta = pymod_2pi(theta);
    theta = (theta > kPi) ? theta - kTwoPi : theta;
    // straight-line: both distances are formed whether or not they are needed (the callers are latency-bound or
    // mix both outcomes in every wave)
    const bool valid = is_valid_angle(theta, l0, l1);
    const double posDiff = angle_diff(theta, l1);
    const double negDiff = angle_diff(theta, l0);
    const double snapped = (fabs(posDiff) < fabs(negDiff)) ? l1 : l0;
    return valid ? theta : snapped;
}

// limit_theta_to_interval for the serial theta phase (no branches, see pymod_2pi_straight)
// l1v: l1 again, for the caller that keeps a copy in a vector register across its loop (a select needs one vector operand)
__device__ __forceinline__ double limit_theta_to_interval_straight(double theta, double l0, double l1, double l1v) {
    theta = pymod_2pi_straight(theta);
    theta = (theta > kPi) ? theta - kTwoPi : theta;
    // is_valid_angle without its short circuits: the limits are launch constants, so `whole` and `wrap` are scalar values
    // and the test is three mask operations instead of three scalar branches per step
    const bool whole = (l0 == l1) | ((fabs(l0) == kPi) & (fabs(l1) == kPi));
    const bool wrap = !(l0 < l1);
    const bool ge = l0 <= theta, le = theta <= l1;
    const bool valid = whole | (wrap ? (ge | le) : (ge & le));
    // theta is in (-pi, pi] here and the interval limits are in [-pi, pi] (control_limits wraps them): the short form
    const double posDiff = angle_diff_near(theta, l1);
    const double negDiff = angle_diff_near(theta, l0);
    const double snapped = (fabs(posDiff) < fabs(negDiff)) ? l1v : l0;
    return valid ? theta : snapped;
}
__device__ __forceinline__ double limit_theta_to_interval_straight(double theta, double l0, double l1) {
    return limit_theta_to_interval_straight(theta, l0, l1, l1);
}

// limit_theta_to_interval's wrap alone (U:93-97): `theta % 2pi`, then `- 2pi` above pi.  Its results are fixed points of
// itself (m - 2 pi is exact for m in (pi, 2 pi], so adding 2 pi gives m back).
__device__ __forceinline__ double wrap_theta_to_pi(double theta) {
    theta = pymod_2pi_straight(theta);
    return (theta > kPi) ? theta - kTwoPi : theta;
}

// The recurrence on previous_theta for the theta phase of the trajectory pipeline — rate limiter (U:252-264 / U:115-127),
// then limit_theta_to_interval (U:93-112) — specialised by the KIND of the control interval [l0, l1] (theta_snap_plan on
// the host decides it and finds `tdag`), ~40 vector instructions and no scalar one (see `opaque`):
//   kSnapInner: l0 < l1, and of the gap's two ends l1 is the nearer one exactly for theta in (l1, tdag)
//   kSnapWrap:  l0 > l1 (the interval contains +-pi), the gap is (l1, l0), l1 is the nearer end exactly below tdag
// `tdag` replaces the reference's comparison |angle_diff(theta, l1)| < |angle_diff(theta, l0)| (U:105-111), which is
// monotonic in theta across the gap: the host finds the one double at which it flips, by bisection with the reference's
// own arithmetic, and checks the equivalence on a sample of the gap (else the generic form runs).
//   g:  the step's goal as the prepare phase left it (NaN = "stay": goal = previous_theta), gw = wrap_theta_to_pi(g)
//   prev in [-pi, pi] (any result of limit_theta_to_interval), 0 <= dmax < pi;  l0, l1, tdag, dmax in vector registers
constexpr int kSnapGeneric = 0, kSnapInner = 1, kSnapWrap = 2;
__device__ __forceinline__ unsigned hi_word(double x) { return (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32); }
__device__ __forceinline__ unsigned lo_word(double x) { return (unsigned)__builtin_bit_cast(unsigned long long, x); }
__device__ __forceinline__ double from_words(unsigned lo, unsigned hi) {
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// The compares and selects of the step are written as instructions: on gfx950 a vector instruction may read a mask
// (vcc or a scalar register pair) two issue slots after the compare that wrote it at the earliest; the compiler keeps
// every compare next to its select and pays an s_nop for each (9 of ~50 slots a step), while here the compares are
// issued early and other work fills the slots.  Within a block, and from one block to the next, every mask is read at
// least two instructions after its compare (the blocks are `volatile`: they keep their order, and the compiler can
// only add instructions between them).  Everything else of the step is plain arithmetic and stays with the compiler.
template <int KIND>
__device__ __forceinline__ double continuous_next_theta_lean(double g, double gw, double prev, double dmax, double l0, double l1,
                                                             double tdag) {
    static_assert(KIND == kSnapInner || KIND == kSnapWrap, "the generic interval goes through continuous_next_theta_goal");
    // angle_diff(goal, previous_theta) (U:486-490) enters the step through |ad| < d_theta_max and through its sign only.
    // The modulo's fix-ups (pymod_2pi: +-2 pi when the quotient was rounded across an integer) move ad by a whole turn from
    // below -pi or from pi on: they never change the first test (|ad| >= pi either way) and flip the sign — so the raw
    // result is taken and the sign flipped where they would have acted.  NaN for "stay" (the unordered compare).
    double adr, q;
    unsigned long long m_stay, m_below, m_above, m_within;
    asm volatile(
        "v_add_f64 %[x], %[g], -%[prev]\n\t"
        "v_cmp_u_f64 %[ms], %[g], %[g]\n\t"
        "v_add_f64 %[x], %[x], %[pi]\n\t"
        "v_mul_f64 %[q], %[x], %[inv]\n\t"
        "v_floor_f64 %[q], %[q]\n\t"
        "v_fma_f64 %[x], -%[q], %[two], %[x]\n\t"
        "v_add_f64 %[x], %[x], -%[pi]\n\t"
        "v_cmp_gt_f64 %[m1], -%[pi], %[x]\n\t"
        "v_cmp_le_f64 %[m2], %[pi], %[x]\n\t"
        "v_cmp_lt_f64 %[mw], |%[x]|, %[dmax]"
        : [x] "=&v"(adr), [q] "=&v"(q), [ms] "=&s"(m_stay), [m1] "=&s"(m_below), [m2] "=&s"(m_above), [mw] "=&s"(m_within)
        : [g] "v"(g), [prev] "v"(prev), [pi] "s"(kPi), [inv] "s"(0.15915494309189535), [two] "s"(kTwoPi), [dmax] "v"(dmax));
    // previous_theta + sign * d_theta_max (sign = ad / |ad| is +-1 exactly), "stay" adds +0.0: then theta = previous_theta,
    // which still goes through limit_theta_to_interval like any other
    unsigned flip, ahi, alo;
    asm volatile(
        "v_cndmask_b32 %[f], 0, %[sign], %[m1]\n\t"
        "v_cndmask_b32 %[f], %[f], %[sign], %[m2]\n\t"
        "v_xor_b32 %[f], %[f], %[xhi]\n\t"
        "v_bfi_b32 %[ahi], %[absmask], %[dhi], %[f]\n\t"
        "v_cndmask_b32 %[ahi], %[ahi], 0, %[ms]\n\t"
        "v_cndmask_b32 %[alo], %[dlo], 0, %[ms]"
        : [f] "=&v"(flip), [ahi] "=&v"(ahi), [alo] "=&v"(alo)
        : [sign] "v"(0x80000000u), [m1] "s"(m_below), [m2] "s"(m_above), [ms] "s"(m_stay), [xhi] "v"(hi_word(adr)),
          [absmask] "s"(0x7fffffffu), [dhi] "v"(hi_word(dmax)), [dlo] "v"(lo_word(dmax)));
    const double tr = prev + from_words(alo, ahi);
    // U:93-97 for tr in (-2 pi, 2 pi): `% 2pi` is tr itself or tr + 2 pi (rounded, as Python does), then `- 2pi` above pi;
    // both addends' words are masked with a sign (of tr, of pi - m; neither is ever -0.0)
    constexpr unsigned kLo = 0x54442d18u, kHi = 0x401921fbu;  // 2 pi = 0x401921fb54442d18
    const unsigned neg = (unsigned)((int)hi_word(tr) >> 31);
    const double m = tr + from_words(neg & kLo, neg & kHi);
    const unsigned over = (unsigned)((int)hi_word(kPi - m) >> 31);
    const double r = m + from_words(over & kLo, over & (kHi | 0x80000000u));
    // theta = goal where the rate limit allows it (its wrap was done by the prepare phase), then U:98-112: inside the
    // interval theta stays, else the nearer end — below tdag that is l1 (or theta is inside)
    unsigned tlo, thi;
    asm volatile(
        "v_cndmask_b32 %[tlo], %[rlo], %[glo], %[mw]\n\t"
        "v_cndmask_b32 %[thi], %[rhi], %[ghi], %[mw]"
        : [tlo] "=&v"(tlo), [thi] "=&v"(thi)
        : [rlo] "v"(lo_word(r)), [rhi] "v"(hi_word(r)), [glo] "v"(lo_word(gw)), [ghi] "v"(hi_word(gw)), [mw] "s"(m_within));
    const double t1 = from_words(tlo, thi);
    double low, high;
    unsigned long long m_low;
    if constexpr (KIND == kSnapInner) {
        // below tdag: [l0, l1] untouched, (l1, tdag) -> l1, below l0 -> l0; from tdag on: l0
        asm volatile(
            "v_cmp_lt_f64 %[mk], %[t], %[tdag]\n\t"
            "v_max_f64 %[lo], %[t], %[l0]\n\t"
            "v_min_f64 %[lo], %[lo], %[l1]"
            : [mk] "=&s"(m_low), [lo] "=&v"(low)
            : [t] "v"(t1), [tdag] "v"(tdag), [l0] "v"(l0), [l1] "v"(l1));
        high = l0;
    } else {
        // the gap is (l1, l0): below tdag min(theta, l1), from tdag on max(theta, l0)
        asm volatile(
            "v_cmp_lt_f64 %[mk], %[t], %[tdag]\n\t"
            "v_min_f64 %[lo], %[t], %[l1]\n\t"
            "v_max_f64 %[hi], %[t], %[l0]"
            : [mk] "=&s"(m_low), [lo] "=&v"(low), [hi] "=&v"(high)
            : [t] "v"(t1), [tdag] "v"(tdag), [l0] "v"(l0), [l1] "v"(l1));
    }
    unsigned olo, ohi;
    asm volatile(
        "v_cndmask_b32 %[olo], %[hlo], %[llo], %[mk]\n\t"
        "v_cndmask_b32 %[ohi], %[hhi], %[lhi], %[mk]"
        : [olo] "=&v"(olo), [ohi] "=&v"(ohi)
        : [hlo] "v"(lo_word(high)), [hhi] "v"(hi_word(high)), [llo] "v"(lo_word(low)), [lhi] "v"(hi_word(low)), [mk] "s"(m_low));
    return from_words(olo, ohi);
}

// ControlIK.safety_checks (C:464-497), in two halves:
//   limit_wrist_cone — utils.limit_orbita3d_joints_wrist (U:508-532): wrist triple as intrinsic XYZ -> ZYZ (alpha, beta,
//     gamma), clamp beta to the Orbita3D cone, back to XYZ (scipy gimbal conventions: beta within 1e-7 of 0 or pi =>
//     gamma := 0).  A function of this step's joints only.
//   multiturn_checks — utils.allow_multiturn (U:493-505) + utils.multiturn_safety_check (U:535-568): needs
//     previous_sol, i.e. the previous step's result.  Returns the cause bits RSIK_EMERGENCY_* of the limits that tripped
//     (the reference appends one message per tripped joint to ControlIK.emergency_state).
// The wrist angles enter as unit vectors (c, s) so no sin/cos is evaluated for joints that came out of atan2.
__device__ __forceinline__ void limit_wrist_cone(UnitAtanTab utab, double (&j)[7], double ca, double sa, double cb, double sb,
                                                 double cc, double sc, double cos_max, double sin_max) {
    // W = Rx(a) Ry(b) Rz(c); its third column (W02, W12, W22) = (sin beta cos alpha, sin beta sin alpha, cos beta) is
    // the wrist axis, so everything about beta in [0, pi] is decided on its cosine / sine, no angle is formed:
    //   beta <= 1e-7 or pi - beta <= 1e-7 (SciPy's gimbal cases)  <=>  sin beta <= sin(1e-7), sign of cos beta
    //   beta > max_angle (the only clamp that can act, beta >= 0)   <=>  cos beta < cos(max_angle)
    double W00 = cb * cc, W02 = sb;
    double W10 = ca * sc + sa * sb * cc, W12 = -sa * cb;
    double W20 = sa * sc - ca * sb * cc, W21 = sa * cc + ca * sb * sc, W22 = ca * cb;
    const double sb2 = W02 * W02 + W12 * W12;
    double cal, sal, cga, sga;  // cos/sin of alpha, gamma
    double cbe = W22, sbe;
    if (RSIK_RARE(sb2 <= 9.999999999999965e-15)) {  // sin(1e-7)^2: gamma := 0, alpha takes the whole z rotation
        const double ih = rsqrt_fast(W00 * W00 + W10 * W10);
        const double sg = (W22 > 0) ? 1.0 : -1.0;
        cal = sg * W00 * ih; sal = sg * W10 * ih; cga = 1.0; sga = 0.0;
        sbe = (sb2 > 0.0) ? sqrt(sb2) : 0.0;
    } else {
        const double isb = rsqrt_fast(sb2);
        sbe = sb2 * isb;
        cal = W02 * isb; sal = W12 * isb;
        const double ih = rsqrt_fast(W20 * W20 + W21 * W21);
        cga = -W20 * ih; sga = W21 * ih;
    }
    if (cbe < cos_max) { cbe = cos_max; sbe = sin_max; }
    // W' = Rz(alpha) Ry(beta') Rz(gamma); intrinsic XYZ angles of W': roll = atan2(-V12, V22), pitch = asin(V02),
    // yaw = atan2(-V01, V00).  (V12, V22) and (V01, V00) both have length cos(pitch) >= cos(max_angle), so one reciprocal
    // square root makes all three direction vectors unit and the table atan2 applies.
    double V02 = cal * sbe, V12 = sal * sbe, V22 = cbe;
    double V01 = -cal * cbe * sga - sal * cga, V00 = cal * cbe * cga - sal * sga;
    const double cp2 = fma(-V02, V02, 1.0);
    const double icp = rsqrt_fast(cp2);
    const double ss[3] = {-V12 * icp, V02, -V01 * icp};
    const double cs[3] = {V22 * icp, cp2 * icp, V00 * icp};
    double aa[3];
    unit_atan2_n<3>(utab, ss, cs, aa);
    j[4] = aa[0]; j[5] = aa[1]; j[6] = aa[2];
}
// one joint: utils.allow_multiturn (U:493-505): prev + angle_diff(j, prev); the wrap only acts when they are > pi apart
__device__ __forceinline__ double allow_multiturn_one(double j, double prev) {
    double t = (j - prev) + kPi;
    if (RSIK_RARE(!(t >= 0.0 && t < kTwoPi))) t = pymod_2pi(t);
    return prev + (t - kPi);
}
__device__ __forceinline__ double allow_multiturn_one_straight(double j, double prev) {
    return prev + (pymod_2pi_straight((j - prev) + kPi) - kPi);
}
// one joint of utils.multiturn_safety_check (U:535-568): joints 0, 2, 6 are clamped to +-6 pi; true if it tripped
__device__ __forceinline__ bool multiturn_limit_one(double& j) {
    const double lim = 6 * kPi;
    bool hit = false;
    if (j > lim) { j = lim; hit = true; }
    if (j < -lim) { j = -lim; hit = true; }
    return hit;
}
__device__ __forceinline__ int multiturn_checks(double (&j)[7], const double* prev) {
#pragma unroll
    for (int k = 0; k < 7; k++) j[k] = allow_multiturn_one(j[k], prev[k]);
    int cause = 0;
    if (multiturn_limit_one(j[0])) cause |= RSIK_EMERGENCY_SHOULDER_PITCH;
    if (multiturn_limit_one(j[2])) cause |= RSIK_EMERGENCY_ELBOW_YAW;
    if (multiturn_limit_one(j[6])) cause |= RSIK_EMERGENCY_WRIST_YAW;
    return cause;
}
__device__ __forceinline__ int safety_checks(UnitAtanTab utab, double (&j)[7], double ca, double sa, double cb, double sb,
                                             double cc, double sc, const double* prev, double max_angle, double cos_max,
                                             double sin_max) {
    (void)max_angle;
    limit_wrist_cone(utab, j, ca, sa, cb, sb, cc, sc, cos_max, sin_max);
    return multiturn_checks(j, prev);
}

// The grid part of utils.get_best_discrete_theta (U:372-396), one pose per lane, walking every grid point.
template <bool PLANE = true, class Acc>
__device__ bool best_discrete_theta_grid(const Acc& A, const Reach& r, double a, double step, double b, int nb, double pref,
                                         double& theta_out) {
    bool found = false;
    double best = 0.0, best_d = __builtin_inf();
    for (int k = 0; k < nb; k++) {
        double th = (k == nb - 1) ? b : ((double)k * step + a);
        double st, ct;
        fast_sincos(th, &st, &ct);
        if (is_elbow_ok<PLANE>(A, elbow_on_circle(r, ct, st))) {
            double dist = fabs(angle_diff(th, pref));
            if (dist < best_d) { best_d = dist; best = th; found = true; }
        }
    }
    theta_out = best;
    return found;
}

// utils.get_best_theta_to_current_joints (U:267-331) with a flat 7-joint target (C:322-324): ternary search over the
// circle; every evaluation is a state-mutating get_joints call exactly like the reference (Q1).
// PAIR: two lanes per trajectory (an even lane and its odd neighbour, `half` = 0 / 1) evaluate the two mid points of an
// iteration side by side and swap the distances.  Only valid where get_joints cannot move the solver's state (no elbow
// projection possible: the caller has checked that the singularity plane is out of the elbow's reach), so that the two
// evaluations of the reference's sequential loop do not depend on each other; the numbers compared are the same.
template <bool PAIR = false, class Acc>
__device__ double best_theta_to_current_joints(const Acc& A, Reach& r, const Rot& Rg, const double* cur, double pref, int half = 0) {
    const double zeros[7] = {0, 0, 0, 0, 0, 0, 0};
    auto dist_at = [&](double th) -> double {
        double st, ct;
        fast_sincos(th, &st, &ct);
        JointsOut o = joints_from_theta<false>(A, r, Rg, ct, st, zeros);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++) { double d = angle_diff(o.j[k], cur[k]); acc += d * d; }
        return sqrt(acc);
    };
    double low = -kPi, high = kPi;
    if (A(RSIK_C_SIDE) < 0) { low = 0; high = kTwoPi; }
    const double tolerance = 0.01;
    if (dist_at(pref) < tolerance) return pref;
    while ((high - low) > tolerance) {
        double mid1 = low + (high - low) / 3;
        double mid2 = high - (high - low) / 3;
        double f1, f2;
        if constexpr (PAIR) {
            const double mine = dist_at(half ? mid2 : mid1);
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, mine);
            const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)bits, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
            const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(bits >> 32), 0xB1, 0xf, 0xf, true);
            const double other = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
            f1 = half ? other : mine;
            f2 = half ? mine : other;
        } else {
            f1 = dist_at(mid1);
            f2 = dist_at(mid2);
        }
        if (f1 < f2) high = mid2; else low = mid1;
    }
    double best = (low + high) / 2;
    if constexpr (!PAIR) (void)dist_at(best);  // U:324 (the call's side effect on the solver state; PAIR: there is none)
    return best;
}

// utils.get_best_discrete_theta's grid search (U:366-396) without walking the whole grid.
//
// Both halves of is_elbow_ok (U:443-465) are of the form  A cos(theta) + B sin(theta) < D  on the elbow circle, i.e.
// each FAILS on one closed arc [phi - alpha, phi + alpha] (R = |(A, B)|, phi = atan2(B, A), alpha = acos(D / R)).
// The grid points that pass are therefore a few index runs delimited by those arc ends and the grid ends, and the
// first-strict-minimum of |angle_diff(theta_k, preferred)| (U:381-388) over a run sits at a run end or next to the
// preferred angle.  This search only runs after the preferred-theta shortcut (U:357-364) has failed, i.e. the
// preferred angle is either outside the interval (no grid point around it: the nearest ones are the grid ends) or inside
// one of the failing arcs (the nearest passing points are the ones just outside that arc's ends).  So the minimum is
// among: the two grid ends and, per arc, the first grid point above its upper end and the last one below its lower end
// — 4 (PLANE = false) or 6 candidates, independent of nb_search_points.  (The grid points just INSIDE an arc end fail by
// construction; if an arc holds no grid point at all, the outside neighbour of one end is the inside neighbour of the
// other.)  Each candidate is then judged with the reference's own predicate and distance (same theta_k = linspace value,
// same is_elbow_ok), so the arcs only propose.  `fast_ok` = false hands the pose to the exhaustive wave-cooperative sweep:
// a degenerate step, an anchor within 1e-10 rad / 1e-6 steps of a grid point (its rounding could move it across), an arc
// that nearly vanishes (its ends are ill-conditioned).  `pref_free`: the caller saw one of the two situations in which
// the shortcut fails although the preferred angle can have passing grid points around it — the whole-circle grid of
// U:366-368 chosen for an interval that does not contain the preferred angle, or a preferred angle outside [-pi, pi]
// (U:468-474 compares it unwrapped; the left arm's mirrored -pi - preferred_theta, C:252) — and the two grid points that
// bracket the preferred angle are judged as well (a wave-uniform branch no wave of a typical launch takes).
// ends_cs = cos/sin of the two grid ends (known from the interval's intersection points, or constants for the whole
// circle): they are not evaluated again.
template <bool PLANE = true, class Acc>
__device__ bool grid_theta_candidates(const Acc& A, const Reach& r, double a, double step, double b, int nb, double pref,
                                      double ca, double sa, double cb, double sb, bool pref_free, double& theta_out,
                                      bool& fast_ok) {
    fast_ok = step > 1e-9;
    const double side = A(RSIK_C_SIDE), sc = A(RSIK_C_SING_COEFF);
    // constraint 1: side * e_y < -0.2;  constraint 2: e_z - sc * e_x < es_z - so - sc * es_x
    const double A1 = side * r.r2 * r.a1.y, B1 = side * r.r2 * r.a2.y, D1 = -0.2 - side * r.c2.y;
    const double R1s = fma(A1, A1, B1 * B1);
    const double q1r = fma(-D1, D1, R1s);
    const bool v1 = q1r > 0.0;  // the constraint really changes sign on the circle
    if (fabs(q1r) < 1e-6 * R1s) fast_ok = false;
    constexpr int NC = PLANE ? 4 : 2;  // candidates besides the grid ends
    double ang[NC];
    bool v2 = false;
    {
        // phi = direction of (A, B), alpha = acos(D / R) = direction of (D, sqrt(R^2 - D^2)): both vectors have length R,
        // so one reciprocal square root per constraint makes them unit and the table atan2 applies.  A constraint that
        // does not change sign on the circle (v false) gets harmless stand-ins; its angles are not used.
        const double Rs1 = v1 ? R1s : 1.0;
        const double i1 = rsqrt_fast(Rs1);
        const double q1 = v1 ? q1r : 1.0;
        const double h1 = v1 ? q1 * rsqrt_fast(q1) : 0.0;
        if constexpr (PLANE) {
            const double A2 = r.r2 * fma(-sc, r.a1.x, r.a1.z), B2 = r.r2 * fma(-sc, r.a2.x, r.a2.z);
            const double D2 = (A(RSIK_C_ES + 2) - A(RSIK_C_SING_OFFSET) - sc * A(RSIK_C_ES)) - fma(-sc, r.c2.x, r.c2.z);
            const double R2s = fma(A2, A2, B2 * B2);
            const double q2r = fma(-D2, D2, R2s);
            v2 = q2r > 0.0;
            if (fabs(q2r) < 1e-6 * R2s) fast_ok = false;
            const double Rs2 = v2 ? R2s : 1.0;
            const double i2 = rsqrt_fast(Rs2);
            const double q2 = v2 ? q2r : 1.0;
            const double h2 = v2 ? q2 * rsqrt_fast(q2) : 0.0;
            const double yy[4] = {(v1 ? B1 : 0.0) * i1, h1 * i1, (v2 ? B2 : 0.0) * i2, h2 * i2};
            const double xx[4] = {(v1 ? A1 : 1.0) * i1, (v1 ? D1 : 1.0) * i1, (v2 ? A2 : 1.0) * i2, (v2 ? D2 : 1.0) * i2};
            double at[4];
            unit_atan2_n<4>(A.utab, yy, xx, at);  // phi_1, alpha_1, phi_2, alpha_2
            ang[0] = at[0] + at[1]; ang[1] = at[0] - at[1];
            ang[2] = at[2] + at[3]; ang[3] = at[2] - at[3];
        } else {
            const double yy[2] = {(v1 ? B1 : 0.0) * i1, h1 * i1};
            const double xx[2] = {(v1 ? A1 : 1.0) * i1, (v1 ? D1 : 1.0) * i1};
            double at[2];
            unit_atan2_n<2>(A.utab, yy, xx, at);
            ang[0] = at[0] + at[1]; ang[1] = at[0] - at[1];
        }
    }
    const double inv_step = fast_rcp(step);
    const double eps = fmax(1e-6, 1e-10 * inv_step);  // in grid steps
    const int last = nb - 1;
    double best_d = __builtin_inf();
    int best_k = 0x7fffffff;
    double best_th = 0.0;
    auto judge = [&](int k, double th, double sn, double cs) {
        if (is_elbow_ok<PLANE>(A, elbow_on_circle(r, cs, sn))) {
            double dist = fabs(angle_diff(th, pref));
            if (dist < best_d || (dist == best_d && k < best_k)) { best_d = dist; best_k = k; best_th = th; }
        }
    };
    judge(0, a, sa, ca);      // the two grid ends
    judge(last, b, sb, cb);
    // arc j: ang[2 j] = its upper end (candidate: the first grid point above), ang[2 j + 1] = its lower end (the last
    // grid point below).  An end beyond the grid clamps to the last point, i.e. to a candidate that is already there.
    int kc[NC];
    double thc[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) {
        const bool arc_valid = (j < 2) ? v1 : v2;
        double pos = pymod_2pi(ang[j] - a) * inv_step;   // real-valued grid index of the arc end, >= 0
        pos = (pos < 2.0e9) ? pos : 0.0;                 // also catches NaN
        const int k0 = (int)pos;
        const double frac = pos - (double)k0;
        if (arc_valid && (frac < eps || frac > 1.0 - eps)) fast_ok = false;
        const int kk = k0 + ((j & 1) ? 0 : 1);
        kc[j] = arc_valid ? (kk > last ? last : kk) : 0;  // an arc without ends proposes the grid's first point again
        thc[j] = (kc[j] == last) ? b : ((double)kc[j] * step + a);  // np.linspace (Q11)
    }
    {
        double sn[NC], cs[NC];
        fast_sincos_n<NC>(thc, sn, cs);
#pragma unroll
        for (int j = 0; j < NC; j++) judge(kc[j], thc[j], sn[j], cs[j]);
    }
    if (RSIK_RARE(__any(pref_free))) {
        double pos = pymod_2pi(pref - a) * inv_step;
        pos = (pos < 2.0e9) ? pos : 0.0;
        const int k0 = (int)pos;
        const double frac = pos - (double)k0;
        if (pref_free && (frac < eps || frac > 1.0 - eps)) fast_ok = false;
        const int kb[2] = {k0 > last ? last : k0, k0 + 1 > last ? last : k0 + 1};
        const double thb[2] = {(kb[0] == last) ? b : ((double)kb[0] * step + a), (kb[1] == last) ? b : ((double)kb[1] * step + a)};
        double sn[2], cs[2];
        fast_sincos_n<2>(thb, sn, cs);
        if (pref_free) { judge(kb[0], thb[0], sn[0], cs[0]); judge(kb[1], thb[1], sn[1], cs[1]); }
    }
    theta_out = best_th;
    return best_k != 0x7fffffff;
}

// utils.get_best_discrete_theta (U:334-396), one pose per lane, as the continuous mode calls it for every control step
// (10 points, C:350-361): the preferred-theta shortcut, then the arc-end candidates of grid_theta_candidates instead of
// a walk over the grid; the few poses that search hands back (fast_ok false) walk it.
// pref_cs / pref_sn: cos / sin of the (launch-constant) preferred angle, host libm.
template <bool PLANE, class Acc>
__device__ bool best_discrete_theta_lane(const Acc& A, const Reach& r, int nb, double pref, double pref_cs, double pref_sn,
                                         double& theta_out) {
    const bool valid = is_valid_angle(pref, r.i0, r.i1);
    if (valid && is_elbow_ok<PLANE>(A, elbow_on_circle(r, pref_cs, pref_sn))) { theta_out = pref; return true; }  // U:357-364
    double a, b, ca, sa, cb, sb;
    bool pref_free = !valid && fabs(pref) > kPi;
    if (fabs(fabs(r.i0) + fabs(r.i1) - kTwoPi) < 0.00001) {  // U:366-368
        a = kPi / 2; b = kPi / 2 + kTwoPi;
        ca = 6.123233995736766e-17; sa = 1.0; cb = 3.061616997868383e-16; sb = 1.0;  // np.cos / np.sin of pi/2, 5pi/2
        pref_free = pref_free || !valid;
    } else {
        a = r.i0; b = (r.i0 < r.i1) ? r.i1 : r.i1 + kTwoPi;
        ca = r.ct0; sa = r.st0; cb = r.ct1; sb = r.st1;
    }
    const double step = (b - a) / (double)(nb - 1);
    bool fast_ok;
    double th;
    bool found = grid_theta_candidates<PLANE>(A, r, a, step, b, nb, pref, ca, sa, cb, sb, pref_free, th, fast_ok);
    if (RSIK_RARE(!fast_ok)) found = best_discrete_theta_grid<PLANE>(A, r, a, step, b, nb, pref, th);
    theta_out = th;
    return found;
}

}  // namespace rsik
