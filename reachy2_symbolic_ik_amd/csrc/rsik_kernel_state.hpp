// rsik_kernel_state.hpp — solver-state kernels of the scalar drop-in API, forward kinematics, the math test hook, the clock monitor
// (one translation unit: included by rsik_lib.hip, in this order, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// Solver-state kernels: the scalar drop-in API (SymbolicIK objects keep `self.goal_pose`,
// `self.wrist_position`, `self.intersection_circle` between is_reachable() and the returned closure, Q1).
// State row layout (RSIK_SOLVER_STATE_STRIDE doubles):
//   0-2 goal position, 3-5 goal euler, 6-8 wrist, 9-11 circle centre, 12 radius, 13-15 circle normal,
//   16-18 elbow position of the last get_joints, 19 projection-fired flag, 20-21 interval, 22 reachable, 23 state code of
//   the last is_reachable, 24-30 joints of the last get_joints, 31 reserved.
// ------------------------------------------------------------------------------------------
struct StateArgs {
    int64_t n;
    const double* in[6];
    const uint8_t* arm;
    int no_limits;
    double* solver_state;
    const double* theta;
    const double* prev;  // [n,7] device or NULL
    double* joints;
    double* interval;
    double* elbow;
    uint8_t* reachable;
    uint8_t* state;
    ArmC arms[2];
};

template <bool MIXED>
__global__ __launch_bounds__(kBlock) void reach_state_kernel(const StateArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(StateArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    if (i >= K.n) return;
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[i] != 0) : false, lds_tab);
    V3 pos = {K.in[0][i], K.in[1][i], K.in[2][i]};
    double e0 = K.in[3][i], e1 = K.in[4][i], e2 = K.in[5][i];
    const double in6[6] = {pos.x, pos.y, pos.z, e0, e1, e2};
    const bool invalid = !all_finite(in6);
    Rot Rg = rot_from_euler(e0, e1, e2);
    Reach r = K.no_limits ? reach<true>(A, pos, Rg) : reach<false>(A, pos, Rg);
    if (RSIK_RARE(invalid)) {  // rsik.h "Rows that are not numbers": the solver object stays as it was
        r.ok = false; r.state = RSIK_STATE_INVALID_INPUT; r.stage = 0; r.i0 = r.i1 = __builtin_nan("");
    }
    double* S = K.solver_state + i * RSIK_SOLVER_STATE_STRIDE;
    if (r.stage >= 1) {
        S[0] = r.pos.x; S[1] = r.pos.y; S[2] = r.pos.z; S[3] = e0; S[4] = e1; S[5] = e2;
        S[6] = r.w.x; S[7] = r.w.y; S[8] = r.w.z;
    }
    if (r.stage >= 2) {
        S[9] = r.c2.x; S[10] = r.c2.y; S[11] = r.c2.z; S[12] = r.r2;
        S[13] = r.n2.x; S[14] = r.n2.y; S[15] = r.n2.z;
    }
    // the call's results also go into the row, so a scalar caller needs ONE download per call
    S[20] = r.i0; S[21] = r.i1; S[22] = r.ok ? 1.0 : 0.0; S[23] = (double)r.state;
    if (K.interval) { K.interval[2 * i] = r.i0; K.interval[2 * i + 1] = r.i1; }
    if (K.reachable) K.reachable[i] = r.ok ? 1 : 0;
    if (K.state) K.state[i] = (uint8_t)r.state;
}

__device__ __forceinline__ Reach reach_from_state(const double* S) {
    Reach r;
    r.ok = true; r.state = RSIK_STATE_REACHABLE; r.stage = 2; r.i0 = -kPi; r.i1 = kPi;
    r.pos = {S[0], S[1], S[2]};
    r.w = {S[6], S[7], S[8]};
    r.c2 = {S[9], S[10], S[11]};
    r.r2 = S[12];
    r.n2 = {S[13], S[14], S[15]};
    Frame F = frame_from_unit(normalized(r.n2));  // S:686: get_elbow_position rebuilds the frame from the stored normal
    r.a1 = F.c1; r.a2 = F.c2;
    return r;
}

template <bool MIXED>
__global__ __launch_bounds__(kBlock) void joints_state_kernel(const StateArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(StateArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    if (i >= K.n) return;
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[i] != 0) : false, lds_tab);
    double* S = K.solver_state + i * RSIK_SOLVER_STATE_STRIDE;
    Reach r = reach_from_state(S);
    Rot Rg = rot_from_euler(S[3], S[4], S[5]);
    double prev[7];
#pragma unroll
    for (int k = 0; k < 7; k++) prev[k] = K.prev ? K.prev[i * 7 + k] : 0.0;
    double st, ct;
    fast_sincos(K.theta[i], &st, &ct);
    JointsOut o = joints_from_theta<false>(A, r, Rg, ct, st, prev);
    if (K.joints) {
#pragma unroll
        for (int k = 0; k < 7; k++) K.joints[i * 7 + k] = o.j[k];
    }
#pragma unroll
    for (int k = 0; k < 7; k++) S[24 + k] = o.j[k];
    S[0] = r.pos.x; S[1] = r.pos.y; S[2] = r.pos.z;
    S[6] = r.w.x; S[7] = r.w.y; S[8] = r.w.z;
    S[16] = o.elbow.x; S[17] = o.elbow.y; S[18] = o.elbow.z;
    S[19] = o.projected ? 1.0 : 0.0;
    if (K.elbow) { K.elbow[3 * i] = o.elbow.x; K.elbow[3 * i + 1] = o.elbow.y; K.elbow[3 * i + 2] = o.elbow.z; }
}

__global__ __launch_bounds__(kBlock) void elbow_state_kernel(const StateArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    stage_sincos_tab();
    __syncthreads();
    if (i >= K.n) return;
    const double* S = K.solver_state + i * RSIK_SOLVER_STATE_STRIDE;
    Reach r = reach_from_state(S);
    double st, ct;
    fast_sincos(K.theta[i], &st, &ct);
    V3 e = elbow_on_circle(r, ct, st);
    K.elbow[3 * i] = e.x; K.elbow[3 * i + 1] = e.y; K.elbow[3 * i + 2] = e.z;
}

// Forward kinematics and the FK(IK(pose)) residual (a checker-free correctness monitor on the device).
struct FkArgs {
    int64_t n;
    const double* joints;   // [n,7]
    const uint8_t* arm;
    int goal_kind;          // residual only: RSIK_GOAL_POSE6 (pose_soa[6]) or RSIK_GOAL_M12 (m12_soa[12])
    const double* goal[12];
    double* pos;            // [n,3] or NULL
    double* rot;            // [n,9] row-major or NULL
    double* err;            // [n,2]: |position error| (m), rotation error (rad) or NULL
    ArmC arms[2];
};

template <bool MIXED>
__global__ __launch_bounds__(kBlock) void fk_kernel(const FkArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    __shared__ SharedTables lds_tab;
    stage_tables<MIXED>(lds_tab, K.arms);
    if (i >= K.n) return;
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[i] != 0) : false, lds_tab);
    double j[7];
#pragma unroll
    for (int k = 0; k < 7; k++) j[k] = K.joints[i * 7 + k];
    const FkOut o = forward_kinematics(A, j);
    if (K.pos) { K.pos[3 * i] = o.pos.x; K.pos[3 * i + 1] = o.pos.y; K.pos[3 * i + 2] = o.pos.z; }
    if (K.rot) {
#pragma unroll
        for (int k = 0; k < 9; k++) K.rot[9 * i + k] = o.R[k];
    }
    if (K.err) {
        V3 gp;
        Rot Rg;
        if (K.goal_kind == RSIK_GOAL_M12) {
#pragma unroll
            for (int k = 0; k < 9; k++) Rg.m[k] = K.goal[k][i];
            gp = {K.goal[9][i], K.goal[10][i], K.goal[11][i]};
        } else {
            gp = {K.goal[0][i], K.goal[1][i], K.goal[2][i]};
            Rg = rot_from_euler(K.goal[3][i], K.goal[4][i], K.goal[5][i]);
        }
        const V3 d = o.pos - gp;
        double fro = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) { const double e = o.R[k] - Rg.m[k]; fro = fma(e, e, fro); }
        // |R1 - R2|_F = 2 sqrt(2) sin(angle / 2): the small-angle value sqrt(fro / 2) is what a monitor needs
        K.err[2 * i] = sqrt(dot(d, d));
        K.err[2 * i + 1] = sqrt(0.5 * fro);
    }
}

// Unit-test hook for rsik_math.hpp (rsik_debug_math): op 0 rcp, 1 sqrt_cr, 2 rsqrt, 3 atan2(a,b), 4 sincos(a), 5 a % 2pi, 6 fp64 FMA issue-rate calibration,
// 7 unit_atan2(s = a, c = b) of a unit vector
__global__ void debug_math_kernel(int op, int64_t n, const double* a, const double* b, double* o0, double* o1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ double utab[3][kUnitAtanRows];
    stage_sincos_tab();
    stage_unit_atan_tab(utab);
    __syncthreads();
    if (i >= n) return;
    double x = a[i], r0 = 0.0, r1 = 0.0;
    switch (op) {
        case 7: {  // the hot path's atan2: direction angle of the UNIT vector (c, s) = (b, a)
            const double ss[1] = {x}, cc[1] = {b[i]};
            double o[1];
            unit_atan2_n<1>((UnitAtanTab)&utab[0][0], ss, cc, o);
            r0 = o[0];
            break;
        }
        case 0: r0 = fast_rcp(x); break;
        case 1: sqrt_rsqrt(x, r0, r1); r1 = sqrt_cr(x); break;
        case 2: r0 = rsqrt_fast(x); break;
        case 3: r0 = fast_atan2(x, b[i]); break;
        case 4: fast_sincos(x, &r0, &r1); break;
        case 5: r0 = pymod_2pi(x); r1 = angle_diff(x, b[i]); break;
        case 6: {  // fp64 VALU calibration (scripts/valu_peak.py): 8 independent chains x 2048 dependent v_fma_f64
            double c[8];
#pragma unroll
            for (int k = 0; k < 8; k++) c[k] = x + k;
            const double m = b[i];
#pragma unroll 1
            for (int it = 0; it < 2048; ++it) {
#pragma unroll
                for (int k = 0; k < 8; k++) c[k] = fma(c[k], m, x);
            }
            r0 = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
            break;
        }
        default: break;
    }
    o0[i] = r0;
    if (o1) o1[i] = r1;
}

// rsik_debug_math op 8: clock monitor.  Each wave of the launch records the shader-clock counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime), sleeps until `ticks[0]` 100 MHz ticks have passed and records both again:
// core clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.  Launched on a side stream while the kernel under study
// runs on the main one it reads the clock the chip holds UNDER THAT LOAD without a single stamp in a product kernel
// (MI355X_MICROARCH.md, DVFS give-back (6)).  The wait is bounded twice: by the tick count (clamped to 5 s) and by an
// iteration budget, so every wave exits.
__global__ void clock_monitor_kernel(const double* ticks, int64_t n_waves, double* core_ticks, double* real_ticks) {
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n_waves) return;
    double want = ticks[0];
    want = want < 0.0 ? 0.0 : (want > 5.0e8 ? 5.0e8 : want);
    const uint64_t dur = (uint64_t)want;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
    uint64_t r = r0;
    for (int guard = 0; guard < 4000000 && r - r0 < dur; ++guard) {
        __builtin_amdgcn_s_sleep(127);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        core_ticks[w] = (double)(c1 - c0);
        real_ticks[w] = (double)(r - r0);
    }
}

}  // namespace rsik
