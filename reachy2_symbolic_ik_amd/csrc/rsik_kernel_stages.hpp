// rsik_kernel_stages.hpp — rsik_stage: the stages of SymbolicIK.is_reachable as entry points of their own
// (one translation unit: included by rsik_lib.hip, in this order, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// The reference exposes the stages of is_reachable as public methods that take their operands as ARGUMENTS (plus
// self.wrist_position, which a caller may assign): is_pose_in_robot_reach S:284-307, get_wrist_position S:418-425,
// get_limitation_wrist_circle S:401-416, get_intersection_circle S:366-399, are_circles_linked S:427-509 (with
// get_interval_from_intersection S:511-568), points_of_nearest_approach S:588-606 (intersection_point S:570-586),
// intersection_circle_line_3d_vd S:608-645, utils.rotation_matrix_from_vector U:59-81 — its own harness times them one by one
// (src/benchmark/ik_benchmarks.py:36-130).  The fused kernels never form these intermediates (rsik_device.hpp: reach_impl
// works on the wrist-centred, closed-form version); this kernel does, stage by stage, on whatever operands it is given — with the
// reference's own sequence of operations, since a caller can hand it circles no pose produces.  One row per lane, row-major in
// and out (the scalar drop-in calls it with n = 1; nothing here is on the batched solve path).
// ------------------------------------------------------------------------------------------
struct StageArgs {
    int64_t n;
    int op;
    const double* in;   // [n][in_stride]
    double* out;        // [n][out_stride]
    int in_stride, out_stride;
    ArmC arms[2];       // (slot 1 unused: stage_tables' layout)
};

__device__ __forceinline__ V3 cross_d(V3 a, V3 b) {  // np.cross: products and differences, nothing fused
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 div_d(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }

// U:59-81, row-major
__device__ inline void rotation_from_vector_ref(V3 vect, double (&R)[9]) {
    const V3 u = div_d(vect, sqrt(dot_d(vect, vect)));
    if (np_isclose(1.0, u.x) && np_isclose(0.0, u.y) && np_isclose(0.0, u.z)) {
        R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
        return;
    }
    if (np_isclose(1.0, -u.x) && np_isclose(0.0, -u.y) && np_isclose(0.0, -u.z)) {
        R[0] = -1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = -1;
        return;
    }
    const V3 v = {0.0, -u.z, u.y};  // (1, 0, 0) x u
    const double c = u.x, s = sqrt(dot_d(v, v));
    const double K[9] = {0.0, -v.z, v.y, v.z, 0.0, -v.x, -v.y, v.x, 0.0};
    const double h = (1 - c) / (s * s);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const double kk = K[3 * r] * K[q] + K[3 * r + 1] * K[3 + q] + K[3 * r + 2] * K[6 + q];
            R[3 * r + q] = ((r == q) ? 1.0 : 0.0) + K[3 * r + q] + kk * h;
        }
}

// S:588-606 with S:570-586: the line of intersection of the two circles' planes.  The reference's least-squares solve
// (np.linalg.lstsq, an SVD) of the 3 x 2 system [v1, -v2] t = p2 - p1, here by a thin QR factorisation — Gram-Schmidt with one
// re-orthogonalisation of the second column — which, like the SVD, works at the system's own condition number: the normal equations
// (rounds 5's form) square it, and for planes a few 1e-6 apart — just outside normal_vector_margin — lost four more digits of q than the
// reference does (the advisor's round-5 finding; G18 holds such pairs).
// false: intersection_point returned [] (np.all(np.isclose(params, params[0])), Q7).
__device__ inline bool nearest_approach_ref(V3 p1, V3 n1, V3 p2, V3 n2, V3& q, V3& v) {
    const V3 c = cross_d(n1, n2);
    v = div_d(c, sqrt(dot_d(c, c)));
    const V3 v1 = cross_d(v, n1), v2 = cross_d(v, n2);
    const V3 b = p2 - p1;
    const V3 a2 = {-v2.x, -v2.y, -v2.z};
    const double r11 = sqrt(dot_d(v1, v1));
    const V3 q1 = div_d(v1, r11);
    double r12 = dot_d(q1, a2);
    V3 w = a2 - q1 * r12;
    const double fix = dot_d(q1, w);  // (second pass: what the first left of q1 in w)
    w = w - q1 * fix;
    r12 += fix;
    const double r22 = sqrt(dot_d(w, w));
    const V3 q2 = div_d(w, r22);
    const double t1 = dot_d(q2, b) / r22;
    const double t0 = (dot_d(q1, b) - r12 * t1) / r11;
    q = {v1.x * t0 + p1.x, v1.y * t0 + p1.y, v1.z * t0 + p1.z};
    return !(np_isclose(t0, t0) && np_isclose(t1, t0));
}

// ---- the policy layer's helpers as the reference's utils module exposes them (U:93-112, 334-396, 443-589): scalar functions of
// explicit arguments that callers of the reference import (src/example/test_ik.py:16-21, test_go_to.py:10-13).  The fused control
// kernels carry specialised forms of the same arithmetic (rsik_device.hpp); these are the general ones — any interval, any limits.
// U:468-474, for any interval (the fused kernels' form assumes ends in [-pi, pi])
__device__ inline bool is_valid_angle_ref(double angle, double i0, double i1) {
    if (pymod_2pi(i0) == pymod_2pi(i1)) return true;
    if (i0 < i1) return (i0 <= angle) && (angle <= i1);
    return (i0 <= angle) || (angle <= i1);
}
// U:93-112 (previous_theta is normalised there and never used, Q12).  `inside`: "theta in interval" / "theta not in interval"
__device__ inline double limit_theta_to_interval_ref(double theta, double i0, double i1, bool& inside) {
    theta = pymod_2pi(theta);
    if (theta > kPi) theta -= kTwoPi;
    inside = is_valid_angle_ref(theta, i0, i1);
    if (inside) return theta;
    const double posDiff = angle_diff(theta, i1), negDiff = angle_diff(theta, i0);
    return (fabs(posDiff) < fabs(negDiff)) ? i1 : i0;
}
// U:443-465 on explicit arguments
__device__ inline bool is_elbow_ok_ref(V3 e, double side, double so, double coeff, V3 esp) {
    bool ok = e.y * side < -0.2;
    ok = ok && (e.z < (e.x - esp.x) * coeff + esp.z - so);
    return ok;
}
// S:684-695 on an explicit circle (centre, radius, normal)
__device__ inline V3 elbow_position_ref(V3 centre, double radius, V3 normal, double theta) {
    double R[9];
    rotation_from_vector_ref(normal, R);
    const double y = radius * cos(theta), z = radius * sin(theta);
    return {R[0] * 0.0 + R[1] * y + R[2] * z + centre.x, R[3] * 0.0 + R[4] * y + R[5] * z + centre.y, R[6] * 0.0 + R[7] * y + R[8] * z + centre.z};
}
// U:334-396.  Returns found; `worked`: the preferred theta itself passed (the reference's early return)
__device__ inline bool best_discrete_theta_ref(double previous_theta, double i0, double i1, int nb, double pref, double side, double so, double coeff,
                                               V3 esp, V3 centre, double radius, V3 normal, double& theta, bool& worked) {
    worked = false;
    if (is_valid_angle_ref(pref, i0, i1) && is_elbow_ok_ref(elbow_position_ref(centre, radius, normal, pref), side, so, coeff, esp)) {
        theta = pref;
        worked = true;
        return true;
    }
    double a, b;
    if (fabs(fabs(i0) + fabs(i1) - 2 * kPi) < 0.00001) { a = kPi / 2; b = kPi / 2 + 2 * kPi; }
    else if (i0 < i1) { a = i0; b = i1; }
    else { a = i0; b = i1 + 2 * kPi; }
    // np.linspace(a, b, nb): k * step + a with step = (b - a) / (nb - 1), the last point b itself (one point: a)
    const double step = nb > 1 ? (b - a) / (double)(nb - 1) : 0.0;
    bool found = false;
    double best_d = __builtin_inf();
    for (int k = 0; k < nb; k++) {
        const double th = (k == nb - 1 && nb > 1) ? b : ((double)k * step + a);
        if (is_elbow_ok_ref(elbow_position_ref(centre, radius, normal, th), side, so, coeff, esp)) {
            const double d = fabs(angle_diff(th, pref));
            if (d < best_d) { best_d = d; theta = th; found = true; }
        }
    }
    if (!found) theta = previous_theta;
    return found;
}

// S:608-645: 0, 1 or 2 points (the + root first)
__device__ inline int circle_line_ref(V3 center, double radius, V3 dir, V3 point, V3& pa, V3& pb) {
    const V3 w = point - center;
    const double a = dot_d(dir, dir), b = 2 * dot_d(dir, w), c = dot_d(w, w) - radius * radius;
    const double disc = b * b - 4 * a * c;
    pa = pb = {__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
    if (disc < 0) return 0;
    if (disc == 0) {
        const double t = -b / (2 * a);
        pa = {point.x + t * dir.x, point.y + t * dir.y, point.z + t * dir.z};
        return 1;
    }
    const double sq = sqrt(disc);
    const double t1 = (-b + sq) / (2 * a), t2 = (-b - sq) / (2 * a);
    pa = {point.x + t1 * dir.x, point.y + t1 * dir.y, point.z + t1 * dir.z};
    pb = {point.x + t2 * dir.x, point.y + t2 * dir.y, point.z + t2 * dir.z};
    return 2;
}

// S:427-568 on explicit circles (centre, radius, normal) and the wrist they are seen from.  Returns how many numbers the
// reference's array holds: 0 ([]: the circles do not cross and the elbow circle lies on the forbidden side) or 2 (the interval).
__device__ inline int circles_linked_ref(double normal_margin, V3 wrist, V3 c2, double r2, V3 n2, V3 c1, double r1, V3 n1, double& i0, double& i1) {
    i0 = i1 = __builtin_nan("");
    const V3 p1 = c1 - wrist, p2 = c2 - wrist;
    double R2[9], R1[9];
    rotation_from_vector_ref(n2, R2);
    rotation_from_vector_ref(n1, R1);
    auto colT = [](const double (&R)[9], int k, V3 a) { return R[k] * a.x + R[3 + k] * a.y + R[6 + k] * a.z; };  // row k of R^T times a
    // T_limitation_torso . p: R1^T p - R1^T p1 (S:456-462); its x decides the side
    auto lim_x = [&](V3 p) { return colT(R1, 0, p) + (-colT(R1, 0, p1)); };
    const bool side = lim_x(p2) > 0;
    auto whole_or_nothing = [&]() -> int {
        if (!side) return 0;
        i0 = -kPi; i1 = kPi;
        return 2;
    };
    V3 N1 = n1, N2 = n2;
    if (N1.x != 0 || N1.y != 0 || N1.z != 0) N1 = div_d(N1, sqrt(dot_d(N1, N1)));
    if (N2.x != 0 || N2.y != 0 || N2.z != 0) N2 = div_d(N2, sqrt(dot_d(N2, N2)));
    const double mg = normal_margin;
    if ((fabs(N2.x - N1.x) < mg && fabs(N2.y - N1.y) < mg && fabs(N2.z - N1.z) < mg) ||
        (fabs(N2.x + N1.x) < mg && fabs(N2.y + N1.y) < mg && fabs(N2.z + N1.z) < mg))
        return whole_or_nothing();
    V3 q, v;
    if (!nearest_approach_ref(p1, N1, p2, N2, q, v)) return whole_or_nothing();
    V3 pa, pb;
    const int np_ = circle_line_ref(p1, r1, v, q, pa, pb);
    if (np_ == 0) return whole_or_nothing();
    // S:511-568: angles of the points in the intersection circle's frame, T_intersection_torso = (R2^T, -R2^T p2)
    auto angle_of = [&](V3 p) {
        const double y = colT(R2, 1, p) + (-colT(R2, 1, p2)), z = colT(R2, 2, p) + (-colT(R2, 2, p2));
        return atan2(z, y);
    };
    if (np_ == 1) {
        i0 = i1 = angle_of(pa);
        return 2;
    }
    double a1 = angle_of(pa), a2 = angle_of(pb);
    if (a2 < a1) { const double t = a1; a1 = a2; a2 = t; }
    const double at = (a1 + a2) / 2;
    const double ly = cos(at) * r2, lz = sin(at) * r2;
    // T_torso_intersection . (0, ly, lz): R2 (0, ly, lz) + p2
    const V3 tp = {R2[1] * ly + R2[2] * lz + p2.x, R2[4] * ly + R2[5] * lz + p2.y, R2[7] * ly + R2[8] * lz + p2.z};
    if (lim_x(tp) > 0) { i0 = a1; i1 = a2; }
    else { i0 = a2; i1 = a1; }
    return 2;
}

__global__ __launch_bounds__(kBlock) void stage_kernel(const StageArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    __shared__ SharedTables lds_tab;
        stage_tables<false, (int)offsetof(StageArgs, arms) + (int)sizeof(ArmC)>(lds_tab, K.arms);
    if (i >= K.n) return;
    const Acc<false> A = make_acc<false>(K.arms, false, lds_tab);
    const double* x = K.in + i * K.in_stride;
    double* o = K.out + i * K.out_stride;
    const V3 s = cvec(A, RSIK_C_SHOULDER);
    const double u = A(RSIK_C_UPPER_ARM), f = A(RSIK_C_FOREARM);
    auto put = [&](int at, V3 a) { o[at] = a.x; o[at + 1] = a.y; o[at + 2] = a.z; };
    switch (K.op) {
    case RSIK_STAGE_POSE_IN_REACH: {  // S:284-307.  in: position 3, euler 3; out: in reach 0/1, position 3, state code
        V3 gp = {x[0], x[1], x[2]};
        const V3 dv = gp - s;
        const double d = sqrt(dot_d(dv, dv));
        int st = RSIK_STATE_EMPTY;
        if (d > A(RSIK_C_MAX_LEN)) {
            const V3 dir = div_d(dv, d + A(RSIK_C_PROJ_MARGIN));
            gp = {s.x + dir.x * A(RSIK_C_MAX_LEN), s.y + dir.y * A(RSIK_C_MAX_LEN), s.z + dir.z * A(RSIK_C_MAX_LEN)};
            st = RSIK_STATE_POSE_OUT_OF_REACH;
        }
        if (gp.x < A(RSIK_C_BACKWARD)) { gp.x = A(RSIK_C_BACKWARD); st = RSIK_STATE_BACKWARD_POSE; }
        o[0] = st == RSIK_STATE_EMPTY ? 1.0 : 0.0;
        put(1, gp);
        o[4] = (double)st;
        break;
    }
    case RSIK_STAGE_WRIST_POSITION: {  // S:418-425.  in: position 3, euler 3; out: wrist 3
        const Rot Rg = rot_from_euler(x[3], x[4], x[5]);
        put(0, wrist_position(make_goal(A, Rg).woff, V3{x[0], x[1], x[2]}));
        break;
    }
    case RSIK_STAGE_LIMITATION_CIRCLE: {  // S:401-416.  in: wrist 3, goal position 3; out: centre 3, radius, normal 3 (not normalised, as the reference)
        const V3 w = {x[0], x[1], x[2]};
        const V3 nv = w - V3{x[3], x[4], x[5]};
        const V3 vec = div_d(nv, sqrt(dot_d(nv, nv))) * A(RSIK_C_WRIST_AX);
        put(0, w + vec);
        o[3] = A(RSIK_C_WRIST_R);
        put(4, nv);
        break;
    }
    case RSIK_STAGE_INTERSECTION_CIRCLE: {  // S:366-399.  in: wrist 3; out: found 0/1, centre 3, radius, normal 3
        const V3 P = V3{x[0], x[1], x[2]} - s;
        const double d = sqrt(P.x * P.x + P.y * P.y + P.z * P.z);
        if (d > u + f) {
            o[0] = 0.0;
            for (int k = 1; k < 8; k++) o[k] = __builtin_nan("");
            break;
        }
        const V3 dir = div_d(P, d);  // M_torso_intersection . e_x (the Euler pair of S:378-385 spells this direction)
        const double kq = d * d - f * f + u * u;
        o[0] = 1.0;
        put(1, s + dir * (kq / (2 * d)));
        o[4] = 1 / (2 * d) * sqrt(4 * (d * d) * (u * u) - kq * kq);
        put(5, dir);
        break;
    }
    case RSIK_STAGE_CIRCLES_LINKED: {  // S:427-568.  in: wrist 3, intersection circle (centre 3, radius, normal 3), limitation circle (same); out: count, interval 2
        double i0, i1;
        const int cnt = circles_linked_ref(A(RSIK_C_NORMAL_MARGIN), V3{x[0], x[1], x[2]}, V3{x[3], x[4], x[5]}, x[6], V3{x[7], x[8], x[9]},
                                           V3{x[10], x[11], x[12]}, x[13], V3{x[14], x[15], x[16]}, i0, i1);
        o[0] = (double)cnt; o[1] = i0; o[2] = i1;
        break;
    }
    case RSIK_STAGE_NEAREST_APPROACH: {  // S:588-606.  in: p1 3, normal1 3, p2 3, normal2 3; out: q found 0/1, q 3, v 3
        V3 q, v;
        const bool ok = nearest_approach_ref(V3{x[0], x[1], x[2]}, V3{x[3], x[4], x[5]}, V3{x[6], x[7], x[8]}, V3{x[9], x[10], x[11]}, q, v);
        o[0] = ok ? 1.0 : 0.0;
        put(1, q);
        put(4, v);
        break;
    }
    case RSIK_STAGE_CIRCLE_LINE: {  // S:608-645.  in: centre 3, radius, direction 3, point on line 3; out: count, point 3, point 3
        V3 pa, pb;
        o[0] = (double)circle_line_ref(V3{x[0], x[1], x[2]}, x[3], V3{x[4], x[5], x[6]}, V3{x[7], x[8], x[9]}, pa, pb);
        put(1, pa);
        put(4, pb);
        break;
    }
    case RSIK_STAGE_ROTATION_FROM_VECTOR: {  // U:59-81.  in: vector 3; out: 3 x 3 row-major
        double R[9];
        rotation_from_vector_ref(V3{x[0], x[1], x[2]}, R);
        for (int k = 0; k < 9; k++) o[k] = R[k];
        break;
    }
    case RSIK_STAGE_ANGLE_DIFF:  // U:486-490.  in: a, b; out: the difference in [-pi, pi)
        o[0] = angle_diff(x[0], x[1]);
        break;
    case RSIK_STAGE_IS_VALID_ANGLE:  // U:468-474.  in: angle, interval 2; out: 0/1
        o[0] = is_valid_angle_ref(x[0], x[1], x[2]) ? 1.0 : 0.0;
        break;
    case RSIK_STAGE_LIMIT_THETA_TO_INTERVAL: {  // U:93-112.  in: theta, previous_theta, interval 2; out: theta, in interval 0/1
        bool inside;
        o[0] = limit_theta_to_interval_ref(x[0], x[2], x[3], inside);
        o[1] = inside ? 1.0 : 0.0;
        break;
    }
    case RSIK_STAGE_IS_ELBOW_OK:  // U:443-465.  in: elbow 3, side (+1 r / -1 l), singularity_offset, singularity_limit_coeff, elbow_singularity_position 3; out: 0/1
        o[0] = is_elbow_ok_ref(V3{x[0], x[1], x[2]}, x[3], x[4], x[5], V3{x[6], x[7], x[8]}) ? 1.0 : 0.0;
        break;
    case RSIK_STAGE_ALLOW_MULTITURN:  // U:493-505.  in: new joints 7, previous joints 7; out: joints 7
        for (int k = 0; k < 7; k++) o[k] = x[7 + k] + angle_diff(x[k], x[7 + k]);
        break;
    case RSIK_STAGE_LIMIT_ORBITA3D_JOINTS: {  // U:508-519.  in: roll, pitch, yaw (intrinsic XYZ), max angle; out: the three angles inside the cone
        double j[7] = {0, 0, 0, 0, x[0], x[1], x[2]};
        limit_wrist_cone(A.utab, j, cos(x[0]), sin(x[0]), cos(x[1]), sin(x[1]), cos(x[2]), sin(x[2]), cos(x[3]), sin(x[3]));
        o[0] = j[4]; o[1] = j[5]; o[2] = j[6];
        break;
    }
    case RSIK_STAGE_MULTITURN_SAFETY_CHECK: {  // U:535-568.  in: joints 7, shoulder pitch / elbow yaw / wrist yaw limits; out: joints 7, RSIK_EMERGENCY_* bits
        int cause = 0;
        for (int k = 0; k < 7; k++) o[k] = x[k];
        const int which[3] = {0, 2, 6}, bit[3] = {RSIK_EMERGENCY_SHOULDER_PITCH, RSIK_EMERGENCY_ELBOW_YAW, RSIK_EMERGENCY_WRIST_YAW};
        for (int q = 0; q < 3; q++) {
            const double lim = x[7 + q];
            if (o[which[q]] > lim) { o[which[q]] = lim; cause |= bit[q]; }
            if (o[which[q]] < -lim) { o[which[q]] = -lim; cause |= bit[q]; }
        }
        o[7] = (double)cause;
        break;
    }
    case RSIK_STAGE_CONTINUITY_CHECK: {  // U:571-589.  in: joints 7, previous joints 7, max angular change 7; out: joints 7 (the previous ones when not continuous), stop 0/1
        bool disc = false;
        for (int k = 0; k < 7; k++) disc = disc || (fabs(angle_diff(x[k], x[7 + k])) > x[14 + k]);
        for (int k = 0; k < 7; k++) o[k] = disc ? x[7 + k] : x[k];
        o[7] = disc ? 1.0 : 0.0;
        break;
    }
    case RSIK_STAGE_BEST_DISCRETE_THETA: {  // U:334-396.  in: previous_theta, interval 2, nb_search_points, preferred_theta, side, singularity_offset,
        // singularity_limit_coeff, elbow_singularity_position 3, the intersection circle get_elbow_position reads (centre 3, radius, normal 3);
        // out: found 0/1, theta, "preferred_theta worked" 0/1
        double in[18];
        for (int k = 0; k < 18; k++) in[k] = x[k];
        double th = in[0];
        bool worked = false;
        // (the grid size is data: a NaN, a negative number or an absurd one must not become the trip count of a loop on the device)
        const int nb = (in[3] >= 0.0 && in[3] <= 1048576.0) ? (int)in[3] : 0;
        const bool found = best_discrete_theta_ref(in[0], in[1], in[2], nb, in[4], in[5], in[6], in[7], V3{in[8], in[9], in[10]}, V3{in[11], in[12], in[13]},
                                                   in[14], V3{in[15], in[16], in[17]}, th, worked);
        o[0] = found ? 1.0 : 0.0; o[1] = th; o[2] = worked ? 1.0 : 0.0;
        break;
    }
    default:
        break;
    }
}

}  // namespace rsik
