// rsik_kernel_continuous.hpp — rsik_control_continuous_step: the state-carrying step kernel and the (re)initialisation kernel
// (one translation unit: included by rsik_lib.hip, in this order, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// ControlIK continuous mode (C:276-407).
// Per-trajectory state lives in a caller-owned SoA array state[RSIK_CONT_STATE_ROWS][n]:
//   row 0 previous_theta, rows 1-7 previous_sol, row 8 init, row 9 emergency_stop, row 10 has_previous_sol,
//   row 11 cause bits and rows 12-18 rejected joints of the step that tripped the emergency stop.
// The reference's wall-clock timeout (C:296-304) becomes the per-trajectory `timed_out` byte.
// ------------------------------------------------------------------------------------------
struct ContinuousArgs {
    int64_t n;
    const double* in[12];
    const double* cur_pose[12];   // current_pose of a (re)initialising trajectory, NULL columns => goal matrix itself
    const uint8_t* arm;
    const uint8_t* timed_out;     // NULL => nobody timed out
    int euler_roundtrip;          // RSIK_OPT_EULER_ROUNDTRIP
    int first_timed_out;          // non-zero: every trajectory (re)initialises
    double pref_arg[2];           // preferred_theta argument per arm slot (mirrored for l)
    double pref_self[2];          // ControlIK.preferred_theta[name] per arm slot
    double pref_self_cs[2], pref_self_sn[2];  // its cosine / sine (host libm, once per launch)
    double lim[2][2];
    double d_theta_max;
    const double* current_joints; // [n,7] or NULL => previous_sol
    double max_angle, cos_max, sin_max;
    double* st;                   // state SoA
    double* joints;
    uint8_t* reachable;
    uint8_t* state;
    unsigned* started_word;       // cont_init_kernel, launch by launch: started_seq is written here when the kernel's last workgroup has
    unsigned started_seq;         // been placed (a prepare kernel of the same run is held on it, rsik_control_continuous_run), or NULL
    ArmC arms[2];
};

// C:296-325: (re)initialisation of a trajectory whose caller timed out: previous_sol := current_joints and
// previous_theta := the theta of the current pose closest to them (utils.get_best_theta_to_current_joints).
// `only_init`: the launch does nothing else (rsik_control_continuous_run's first phase).
template <bool PAIR = false, class Acc>
__device__ __forceinline__ void continuous_reinit(const Acc& A, const ContinuousArgs& K, int64_t ii, double pref,
                                                  double& prev_theta, double (&prev_sol)[7], int half = 0) {
    if (K.current_joints) {
#pragma unroll
        for (int k = 0; k < 7; k++) prev_sol[k] = K.current_joints[ii * 7 + k];
    }
    Rot Rc;
    V3 cpos;
    load_m12(K.cur_pose[0] ? K.cur_pose : K.in, ii, Rc, cpos, K.euler_roundtrip);
    Reach rc = reach<true>(A, cpos, Rc);
    prev_theta = best_theta_to_current_joints<PAIR>(A, rc, Rc, prev_sol, pref, half);
}

// U:571-589 continuity_check with the thresholds of C:398
__device__ __forceinline__ bool joints_discontinuous(const double (&jv)[7], const double (&prev)[7]) {
    bool disc = false;
#pragma unroll
    for (int k = 0; k < 7; k++) disc = disc || (fabs(angle_diff(jv[k], prev[k])) > (k < 4 ? 0.5 : 1.0));
    return disc;
}

// The state-independent front half of a control step (C:327-388 up to the rate limiter): is_reachable, and then
// either the 10-point search for the target theta (get_best_continuous_theta2 -> get_best_discrete_theta, U:220-264)
// or, for an unreachable goal, is_reachable_no_limits.  `r` is left holding the geometry get_joints will use.
struct ThetaTarget {
    bool ok_limits;   // is_reachable succeeded
    bool found;       // ... and the grid search found an elbow-ok theta
    double theta;     // the search's theta (found only)
    int code;         // state code the step reports
};
// FALLBACK_GEOMETRY = false (the pipeline's prepare phase): the unreachable side's is_reachable_no_limits is left to
// the phase that needs its circle.
template <bool PLANE, bool FALLBACK_GEOMETRY = true, class Acc>
__device__ __forceinline__ ThetaTarget continuous_target(const Acc& A, V3 pos, const V3 woff, double pref_self, double pref_cs,
                                                         double pref_sn, Reach& r) {
    ThetaTarget T;
    r = reach_g<false, false>(A, pos, woff);
    T.ok_limits = r.ok;
    T.found = false;
    T.theta = 0.0;
    T.code = RSIK_STATE_EMPTY;
    if (r.ok) {
        T.found = best_discrete_theta_lane<PLANE>(A, r, 10, pref_self, pref_cs, pref_sn, T.theta);
        if (!T.found) T.code = RSIK_STATE_LIMITED_BY_SHOULDER;
    } else {
        T.code = r.state;
        if constexpr (FALLBACK_GEOMETRY) r = reach_g<true>(A, pos, woff);
    }
    return T;
}
// The recurrence on previous_theta: rate limiter of get_best_continuous_theta2 (U:252-264) / tend_to_preferred_theta
// (U:115-127), then limit_theta_to_interval (U:93-112).
// dmax_v / l1v: the same values again, for the caller that keeps copies in vector registers across its loop.
__device__ __forceinline__ double continuous_next_theta_goal(double goal, double prev_theta, double d_theta_max, double l0,
                                                            double l1, double dmax_v, double l1v) {
    // sign * d_theta_max with sign = ad / |ad| (U:260, U:126) is copysign(d_theta_max, ad), bit for bit: the quotient
    // of a non-zero finite number by its own magnitude is exactly +-1.
    const double ad = angle_diff_straight(goal, prev_theta);
    const double theta = (fabs(ad) < d_theta_max) ? goal : (prev_theta + copysign(dmax_v, ad));
    return limit_theta_to_interval_straight(theta, l0, l1, l1v);
}
__device__ __forceinline__ double continuous_next_theta_goal(double goal, double prev_theta, double d_theta_max, double l0,
                                                            double l1) {
    return continuous_next_theta_goal(goal, prev_theta, d_theta_max, l0, l1, d_theta_max, l1);
}
__device__ __forceinline__ double continuous_next_theta(bool ok_limits, bool found, double target, double pref_arg,
                                                        double prev_theta, double d_theta_max, double l0, double l1) {
    // One straight line for the three cases (this is the serial part of a trajectory: a lone wave pays every dependent
    // instruction in full).  Reachable and found: tend to the search's theta (U:252-264); reachable, nothing found:
    // stay (goal = previous_theta, whose angle_diff is 0); unreachable: tend to the preferred theta (U:115-127).
    const double goal = ok_limits ? (found ? target : prev_theta) : pref_arg;
    return continuous_next_theta_goal(goal, prev_theta, d_theta_max, l0, l1);
}
// One launch = one control step of n independent trajectories (rsik_control_continuous_step): everything fused, the
// trajectory state makes one round trip through HBM.
template <bool MIXED, bool PLANE>
__global__ __launch_bounds__(kBlock) void control_continuous_kernel(const ContinuousArgs K) {
    __shared__ double lds_out[kBlock / 64][64 * 7];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t wave_base = (int64_t)blockIdx.x * kBlock + wave * 64;
    const bool live = i < K.n;
    const int64_t ii = live ? i : (K.n - 1);
    const int64_t n = K.n;

    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(ContinuousArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[ii] != 0) : false, lds_tab);
    const int slot = MIXED ? (A.isl ? 1 : 0) : 0;

    double prev_theta = K.st[0 * n + ii];
    double prev_sol[7];
#pragma unroll
    for (int k = 0; k < 7; k++) prev_sol[k] = K.st[(1 + k) * n + ii];
    bool init = K.st[8 * n + ii] != 0.0;
    bool emergency = K.st[9 * n + ii] != 0.0;
    bool has_prev = K.st[10 * n + ii] != 0.0;

    double jv[7], rejected[7];
    int st_code = RSIK_STATE_EMPTY;
    int cause = 0;
    bool ok = false;
    if (emergency) {  // C:205-210
#pragma unroll
        for (int k = 0; k < 7; k++) jv[k] = prev_sol[k];
        st_code = RSIK_STATE_EMERGENCY;
    } else {
        Rot Rg;
        V3 pos;
        double m[12];
#pragma unroll
        for (int k = 0; k < 12; k++) m[k] = K.in[k][ii];
        const bool invalid = !all_finite(m);
        goal_from_m12(m, Rg, pos, K.euler_roundtrip);
        if (K.first_timed_out || (K.timed_out && K.timed_out[ii])) { has_prev = false; init = true; }  // C:298-304
        if (!has_prev) {  // C:306-325
            has_prev = true;
            continuous_reinit(A, K, ii, K.pref_arg[slot], prev_theta, prev_sol);
        }
        const Goal G = make_goal(A, Rg);
        Reach r;
        const ThetaTarget T = continuous_target<PLANE>(A, pos, G.woff, K.pref_self[slot], K.pref_self_cs[slot], K.pref_self_sn[slot], r);
        if (RSIK_RARE(invalid)) {
            // rsik.h "Rows that are not numbers": no joints; previous_sol, init and the latch stay as they are; previous_theta goes
            // through the step of a search that found nothing (U:252-264 with goal = previous_theta, then U:93-112)
#pragma unroll
            for (int k = 0; k < 7; k++) jv[k] = __builtin_nan("");
            st_code = RSIK_STATE_INVALID_INPUT;
            prev_theta = continuous_next_theta(true, false, 0.0, K.pref_arg[slot], prev_theta, K.d_theta_max, K.lim[slot][0], K.lim[slot][1]);
        } else if (RSIK_RARE(!T.ok_limits && !r.ok)) {
            // C:385-387: is_reachable_no_limits came back false (only a solver whose projection_margin lets the pulled-back
            // wrist land beyond u + f can do that, S:343-345) and the reference raises RuntimeError — before it touches
            // previous_theta, previous_sol or init.  Reported as data: NaN joints, RSIK_STATE_NOT_REACHABLE_NO_LIMITS.
#pragma unroll
            for (int k = 0; k < 7; k++) jv[k] = __builtin_nan("");
            st_code = RSIK_STATE_NOT_REACHABLE_NO_LIMITS;
        } else {
        ok = T.ok_limits && T.found;
        st_code = T.code;
        const double theta = continuous_next_theta(T.ok_limits, T.found, T.theta, K.pref_arg[slot], prev_theta, K.d_theta_max,
                                                   K.lim[slot][0], K.lim[slot][1]);
        prev_theta = theta;
        double sn, cs;
        fast_sincos(theta, &sn, &cs);
        JointsOut o = joints_from_theta_g<true>(A, r, G, cs, sn, prev_sol);
#pragma unroll
        for (int k = 0; k < 7; k++) jv[k] = o.j[k];
        cause = safety_checks(A.utab, jv, o.c4, o.s4, o.c5, o.s5, o.c6, o.s6, prev_sol, K.max_angle, K.cos_max, K.sin_max);
        emergency = cause != 0;
        if (!init && joints_discontinuous(jv, prev_sol)) {  // U:571-589 continuity_check, thresholds C:398
            cause |= RSIK_EMERGENCY_CONTINUITY;
            emergency = true;
#pragma unroll
            for (int k = 0; k < 7; k++) { rejected[k] = jv[k]; jv[k] = prev_sol[k]; }
        }
        init = false;
        if (!emergency) {
#pragma unroll
            for (int k = 0; k < 7; k++) prev_sol[k] = jv[k];
        }
        }
    }
    store_rows<7>(K.joints, wave_base, K.n, lane, lds_out[wave], jv);
    if (live) {
        if (K.reachable) K.reachable[i] = ok ? 1 : 0;
        if (K.state) K.state[i] = (uint8_t)st_code;
        K.st[0 * n + i] = prev_theta;
#pragma unroll
        for (int k = 0; k < 7; k++) K.st[(1 + k) * n + i] = prev_sol[k];
        K.st[8 * n + i] = init ? 1.0 : 0.0;
        K.st[9 * n + i] = emergency ? 1.0 : 0.0;
        K.st[10 * n + i] = has_prev ? 1.0 : 0.0;
        if (cause != 0) {
            K.st[11 * n + i] = (double)cause;
            if (cause & RSIK_EMERGENCY_CONTINUITY) {
#pragma unroll
                for (int k = 0; k < 7; k++) K.st[(12 + k) * n + i] = rejected[k];
            }
        }
    }
}

// C:296-325 for the trajectories of a batch that (re)initialise: previous_sol, previous_theta, init.
// PAIR: two lanes per trajectory share the start-up search (best_theta_to_current_joints<PAIR>): half its latency, which
// is on the critical path of a run.
template <bool MIXED, bool PAIR>
__global__ __launch_bounds__(kBlock) void cont_init_kernel(const ContinuousArgs K) {
    __builtin_amdgcn_s_setprio(3);  // a few lone waves on the critical path, beside the chip-filling prepare phase
    if (K.started_word != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)  // (the last workgroup runs: all of them have been placed)
        __hip_atomic_store(K.started_word, K.started_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i = PAIR ? (gid >> 1) : gid;
    const int half = PAIR ? (int)(gid & 1) : 0;
    const bool live = i < K.n;
    const int64_t ii = live ? i : (K.n - 1);
    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(ContinuousArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[ii] != 0) : false, lds_tab);
    const int slot = MIXED ? (A.isl ? 1 : 0) : 0;
    const int64_t n = K.n;
    if (!live || K.st[9 * n + i] != 0.0) return;  // emergency latched: nothing is touched (C:205-210)
    const bool timed_out = K.first_timed_out || (K.timed_out && K.timed_out[i]);
    if (!timed_out && K.st[10 * n + i] != 0.0) return;
    double prev_theta = K.st[0 * n + i];
    double prev_sol[7];
#pragma unroll
    for (int k = 0; k < 7; k++) prev_sol[k] = K.st[(1 + k) * n + i];
    continuous_reinit<PAIR>(A, K, i, K.pref_arg[slot], prev_theta, prev_sol, half);
    if (half != 0) return;  // (both lanes of a pair hold the same result)
    K.st[0 * n + i] = prev_theta;
#pragma unroll
    for (int k = 0; k < 7; k++) K.st[(1 + k) * n + i] = prev_sol[k];
    K.st[8 * n + i] = 1.0;
    K.st[10 * n + i] = 1.0;
}

}  // namespace rsik
