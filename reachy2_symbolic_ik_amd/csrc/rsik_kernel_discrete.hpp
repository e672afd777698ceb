// rsik_kernel_discrete.hpp — rsik_control_discrete: the theta-grid sweep and control_discrete_kernel
// (one translation unit: included by rsik_lib.hip, in this order, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// ControlIK discrete mode (C:162-274, C:409-497)
// ------------------------------------------------------------------------------------------
struct DiscreteArgs {
    int64_t n;
    const double* in[12];
    const uint8_t* arm;
    int nb;
    int log2p;            // sweep sub-group width P = 1 << log2p  (P = pow2ceil(min(nb, 64)))
    int sweep_mode;       // 0 auto, 1 always the exhaustive wave-cooperative sweep, 2 always the per-lane search
    int euler_roundtrip;  // RSIK_OPT_EULER_ROUNDTRIP
    int stagger;          // non-zero: a single-round launch, waves lower their issue priority as they advance (RSIK_DISC_PRIO)
    double pref[2];       // preferred theta per arm slot (already mirrored for l, C:252)
    double pref_cs[2], pref_sn[2];  // its cosine / sine (host libm, once per launch)
    double lim[2][2];     // interval_limit per arm slot (C:225-250)
    double prev_sol[2][7];
    double prev_cs[2][3], prev_sn[2][3];  // cos / sin of previous_sol[4..6] (host libm): the wrist of a pose that falls back to it
    const double* current_joints;
    double max_angle, cos_max, sin_max;
    double* joints;
    uint8_t* reachable;
    uint8_t* state;
    uint8_t* emergency;
    ArmC arms[2];
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-wave LDS slab of the discrete kernel, kDiscRows rows of 64 doubles (one per lane), so that four workgroups fit in
// a CU's 160 KB (the staged circle geometry used to sit in its own 13-row array: 47.6 KB per workgroup, 3 per CU):
//   rows 0-5   the two goal vectors the joint stage reads again (parked during the theta search);
//              after the search the first 7 * 64 doubles are the [64][7] output staging area
//   rows 6-16  circle geometry for the wave-cooperative sweep: c2 (3), r2 a1 (3), r2 a2 (3), grid ends a, b
//              (the step is (b - a) / (nb - 1), recomputed by its readers); the sweep's result for pose p overwrites
//              a[p], which only p's own sub-group reads, and only before it posts the result
constexpr int kDiscRows = 17;
constexpr int kGeoRow0 = 6;

// utils.get_best_discrete_theta (U:334-396) for the poses of one wave that need the grid.
// Lane-per-pose has already staged the circle geometry of its pose in LDS; here the wave walks the set
// bits of `mask` and gives every selected pose a P-lane sub-group: lane k evaluates theta_k, elbow-ok
// (U:443-465) and |angle_diff(theta_k, preferred)|, a segmented xor-butterfly keeps the lexicographic
// minimum of (distance, k) = the reference's "first strict minimum" (U:381-388), the sub-group leader
// posts the winner.  nb > 64 is handled by extra rounds of the same lanes.
template <bool MIXED, bool PLANE>
__device__ __forceinline__ void sweep_theta_grid(const DiscreteArgs& K, uint64_t mask, int lane, bool my_isl,
                                                 SharedTables& lds_tab, const double (*geo)[64], double* res) {
    const int P = 1 << K.log2p;
    const int G = 64 >> K.log2p;
    const int sub = lane >> K.log2p;
    const int k0 = lane & (P - 1);
    const int rounds = (K.nb + 63) >> 6;
    const double inf = __builtin_inf();
    while (mask) {
        int p = -1;
        uint64_t m = mask;
        for (int g = 0; g < G; g++) {
            if (m) {
                int bit = __builtin_ctzll(m);
                if (g == sub) p = bit;
                m &= m - 1;
            }
        }
        mask = m;
        double best_d = inf;
        int best_k = 0x7fffffff;
        double ga = 0, gstep = 0, gb = 0;
        int src = p < 0 ? lane : p;
        bool isl = MIXED ? (__shfl((int)my_isl, src) != 0) : false;
        const Acc<MIXED> A = make_acc<MIXED>(K.arms, isl, lds_tab);
        const int slot = MIXED ? (isl ? 1 : 0) : 0;
        if (p >= 0) {
            V3 c2 = {geo[0][p], geo[1][p], geo[2][p]};
            V3 a1 = {geo[3][p], geo[4][p], geo[5][p]};  // r2 a1
            V3 a2 = {geo[6][p], geo[7][p], geo[8][p]};  // r2 a2
            ga = geo[9][p]; gb = geo[10][p];
            gstep = (gb - ga) / (double)(K.nb - 1);
            const double pref = K.pref[slot];
            for (int rd = 0; rd < rounds; rd++) {
                int k = k0 + (rd << 6);
                if (k < K.nb) {
                    double th = (k == K.nb - 1) ? gb : ((double)k * gstep + ga);  // np.linspace (Q11)
                    double st, ct;
                    fast_sincos(th, &st, &ct);
                    V3 e = {a1.x * ct + a2.x * st + c2.x, a1.y * ct + a2.y * st + c2.y, a1.z * ct + a2.z * st + c2.z};
                    if (is_elbow_ok<PLANE>(A, e)) {
                        double dist = fabs(angle_diff(th, pref));
                        if (dist < best_d) { best_d = dist; best_k = k; }
                    }
                }
            }
        }
        for (int off = P >> 1; off >= 1; off >>= 1) {
            double od = __shfl_xor(best_d, off);
            int ok = __shfl_xor(best_k, off);
            if (od < best_d || (od == best_d && ok < best_k)) { best_d = od; best_k = ok; }
        }
        if (p >= 0 && k0 == 0) {
            double th = __builtin_nan("");
            if (best_k != 0x7fffffff) th = (best_k == K.nb - 1) ? gb : ((double)best_k * gstep + ga);
            res[p] = th;
        }
    }
}

#ifndef RSIK_DISC_ATTR
#define RSIK_DISC_ATTR
#endif
#ifndef RSIK_DISC_MIN_WAVES
#define RSIK_DISC_MIN_WAVES 3  // 168 VGPR (28 B scratch) beats 182 VGPR at 2 waves/SIMD: 37.6 vs 39.0 us on config 3
#endif
// PLANE = false: the singularity-plane half of is_elbow_ok can never fail for these arms (decided on the host).
// Issue priority by progress.  A launch of up to one workgroup per resident slot (262 144 matrices on MI355X: BASELINE
// config 3) is a single round: the four waves of a SIMD start together, the hardware serves the oldest first, and from the
// moment the first of them finishes the SIMD runs with fewer and fewer waves to hide its latencies behind
// (profiles/r03/timeline/discrete_*: first waves done at 9 us, last at 15.4).  So in such a launch (DiscreteArgs.stagger,
// set by the host) a wave LOWERS its priority as it advances — reach 3, search 2, joints 1, safety 0 — and the SIMD
// always serves the wave that is furthest behind: all four stay in flight to the end.  15.8 -> 14.5 us at 262 144 matrices,
// 11.2 -> 10.4 at 131 072; a launch of several rounds wants the opposite (its old waves should finish and make room:
// +4 % at 524 288, +15 % at 1 M with the priorities on), hence the switch.
#define RSIK_DISC_PRIO(p) do { if (K.stagger) __builtin_amdgcn_s_setprio(p); } while (0)
#ifndef RSIK_DISC_BLOCK
#define RSIK_DISC_BLOCK 256
#endif
constexpr int kDiscBlock = RSIK_DISC_BLOCK;  // threads per workgroup of the discrete kernel
template <bool MIXED, bool PLANE>
__global__ __launch_bounds__(kDiscBlock, RSIK_DISC_MIN_WAVES) RSIK_DISC_ATTR void control_discrete_kernel(const DiscreteArgs K) {
    __shared__ double lds_slab[kDiscBlock / 64][kDiscRows][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kDiscBlock + threadIdx.x;
    const int64_t wave_base = (int64_t)blockIdx.x * kDiscBlock + wave * 64;
    const bool live = i < K.n;
    const int64_t ii = live ? i : (K.n - 1);

#ifdef RSIK_TIMELINE_PROBE
    uint64_t probe_t[7];  // (6: joints done, safety_checks not yet — the stage timers' extra stamp)
    probe_t[0] = __builtin_amdgcn_s_memrealtime();
#define RSIK_DISC_PROBE(k) do { __builtin_amdgcn_sched_barrier(0); probe_t[k] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RSIK_DISC_PROBE(k) do { } while (0)
#endif
    // the twelve goal-matrix loads (and the arm byte) are issued before the table staging so that their latency overlaps it
    double m12[12];
#pragma unroll
    for (int k = 0; k < 12; k++) m12[k] = K.in[k][ii];
    const bool lane_isl = MIXED ? (K.arm[ii] != 0) : false;
        __shared__ SharedTables lds_tab;
    stage_tables<MIXED, (int)offsetof(DiscreteArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC)), kDiscBlock>(lds_tab, K.arms);
#ifdef RSIK_TIMELINE_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (probe build only: the head ends when the twelve columns are in)
#endif
    RSIK_DISC_PROBE(1);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, lane_isl, lds_tab);
    const int slot = MIXED ? (A.isl ? 1 : 0) : 0;

    // C:212-217: M -> pose (see goal_from_m12)
    const bool invalid = !all_finite(m12);  // rsik.h "Rows that are not numbers" (judged while the twelve values are at hand)
    Rot Rg;
    V3 pos;
    goal_from_m12(m12, Rg, pos, K.euler_roundtrip);

    // Of the goal orientation the solver only needs three vectors (Goal); the two that are read again by the joint
    // stage wait in the output staging slab while the theta search runs (registers are the scarce resource here)
    Goal G = make_goal(A, Rg);
    {
        const double pk[6] = {G.tw.x, G.tw.y, G.tw.z, G.xg.x, G.xg.y, G.xg.z};
#pragma unroll
        for (int k = 0; k < 6; k++) lds_slab[wave][k][lane] = pk[k];
    }
    RSIK_MARK("disc_reach");
    RSIK_DISC_PRIO(3);
    Reach r = reach_g<false, false>(A, pos, G.woff);
    if (RSIK_RARE(invalid)) { r.ok = false; r.state = RSIK_STATE_INVALID_INPUT; }
    RSIK_MARK("disc_shortcut");
    const double pref = K.pref[slot];
    bool found = false;
    double theta = 0.0;
    bool need = false;
    bool pref_valid = false;
    if (r.ok) {  // U:357-364 preferred-theta shortcut
        pref_valid = is_valid_angle(pref, r.i0, r.i1);
        if (pref_valid) {
            const double st = K.pref_sn[slot], ct = K.pref_cs[slot];  // launch-uniform: not evaluated per lane
            if (is_elbow_ok<PLANE>(A, elbow_on_circle(r, ct, st))) { found = true; theta = pref; }
        }
        need = !found;
    }
    double ca = 1.0, sa = 0.0, cb = 1.0, sb = 0.0;  // cos / sin of the grid's end points
    // grid points may pass on both sides of the preferred angle although the shortcut failed (see grid_theta_candidates)
    bool pref_free = need && !pref_valid && fabs(pref) > kPi;
    if (need) {  // U:366-375 grid end points
        double a, b;
        if (fabs(fabs(r.i0) + fabs(r.i1) - kTwoPi) < 0.00001) {
            a = kPi / 2; b = kPi / 2 + kTwoPi;
            ca = 6.123233995736766e-17; sa = 1.0; cb = 3.061616997868383e-16; sb = 1.0;  // np.cos / np.sin of pi/2, 5pi/2
            pref_free = pref_free || !pref_valid;
        } else {
            a = r.i0; b = (r.i0 < r.i1) ? r.i1 : r.i1 + kTwoPi;
            ca = r.ct0; sa = r.st0; cb = r.ct1; sb = r.st1;  // the interval's own intersection points (reach_g)
        }
        double (*g)[64] = &lds_slab[wave][kGeoRow0];
        g[0][lane] = r.c2.x; g[1][lane] = r.c2.y; g[2][lane] = r.c2.z;
        g[3][lane] = r.r2 * r.a1.x; g[4][lane] = r.r2 * r.a1.y; g[5][lane] = r.r2 * r.a1.z;
        g[6][lane] = r.r2 * r.a2.x; g[7][lane] = r.r2 * r.a2.y; g[8][lane] = r.r2 * r.a2.z;
        g[9][lane] = a; g[10][lane] = b;
    }
    // Two ways to search the grid, chosen per wave (wave-uniform): when only a few lanes need it, the exhaustive
    // wave-cooperative sweep (cost ~ number of such poses); when most lanes need it, every lane searches its own
    // pose serially — the whole grid if it is short, else the 4 / 6 arc-end candidates (grid_theta_candidates).
    RSIK_MARK("disc_grid");
    RSIK_DISC_PRIO(2);
    RSIK_DISC_PROBE(2);
    const uint64_t need_mask = __ballot(need);
    const int cnt = __popcll(need_mask);
    const bool walk = K.nb <= 4;  // a grid this short is cheaper to walk than to analyse
    const int serial_cost = walk ? K.nb * 60 : (PLANE ? 340 : 230);
    const int coop_rounds = ((cnt + (64 >> K.log2p) - 1) >> (6 - K.log2p)) * ((K.nb + 63) >> 6);
    bool dense = serial_cost < coop_rounds * 150;
    if (K.sweep_mode == 1) dense = false;
    if (K.sweep_mode == 2) dense = true;
    bool coop = need;
    double th_serial = 0.0;
    bool found_serial = false;
    if (dense && need) {
        const double ga = lds_slab[wave][kGeoRow0 + 9][lane], gb = lds_slab[wave][kGeoRow0 + 10][lane];
        const double gstep = (gb - ga) / (double)(K.nb - 1);
        if (walk) {
            found_serial = best_discrete_theta_grid<PLANE>(A, r, ga, gstep, gb, K.nb, pref, th_serial);
            coop = false;
        } else {
            bool fast_ok;
            found_serial = grid_theta_candidates<PLANE>(A, r, ga, gstep, gb, K.nb, pref, ca, sa, cb, sb, pref_free, th_serial, fast_ok);
            coop = !fast_ok;
        }
    }
    RSIK_MARK("disc_sweep");
    const uint64_t mask = __ballot(coop);
    if (mask) {  // wave-uniform: most waves of a dense launch have nothing for the cooperative sweep
        wave_lds_sync();
        sweep_theta_grid<MIXED, PLANE>(K, mask, lane, A.isl, lds_tab, &lds_slab[wave][kGeoRow0], &lds_slab[wave][kGeoRow0 + 9][0]);
        wave_lds_sync();
    }
    int st_code = r.state;
    if (need) {
        double th = coop ? lds_slab[wave][kGeoRow0 + 9][lane] : (found_serial ? th_serial : __builtin_nan(""));
        if (th == th) { found = true; theta = th; }
        else st_code = RSIK_STATE_LIMITED_BY_SHOULDER;  // C:451-452
    }

    RSIK_MARK("disc_joints");
    RSIK_DISC_PRIO(1);
    RSIK_DISC_PROBE(3);
    const double* prev = K.prev_sol[slot];
    double jv[7];
    double c4, s4, c5, s5, c6, s6;
    if (found) {  // C:454-456
        theta = limit_theta_to_interval(theta, K.lim[slot][0], K.lim[slot][1]);
        double st, ct;
        fast_sincos(theta, &st, &ct);
        const double* pk = &lds_slab[wave][0][0];
        G.tw = {pk[0 * 64 + lane], pk[1 * 64 + lane], pk[2 * 64 + lane]};
        G.xg = {pk[3 * 64 + lane], pk[4 * 64 + lane], pk[5 * 64 + lane]};
        JointsOut o = joints_from_theta_g<true>(A, r, G, ct, st, prev);
#pragma unroll
        for (int k = 0; k < 7; k++) jv[k] = o.j[k];
        c4 = o.c4; s4 = o.s4; c5 = o.c5; s5 = o.s5; c6 = o.c6; s6 = o.s6;
    } else if (K.current_joints) {  // C:457-458
#pragma unroll
        for (int k = 0; k < 7; k++) jv[k] = K.current_joints[ii * 7 + k];
        const double w3[3] = {jv[4], jv[5], jv[6]};
        double sn3[3], cs3[3];
        fast_sincos_n<3>(w3, sn3, cs3);
        c4 = cs3[0]; s4 = sn3[0]; c5 = cs3[1]; s5 = sn3[1]; c6 = cs3[2]; s6 = sn3[2];
    } else {  // current_joints defaults to previous_sol (C:237-238): launch constants, their sin / cos come with the launch
#pragma unroll
        for (int k = 0; k < 7; k++) jv[k] = prev[k];
        c4 = K.prev_cs[slot][0]; s4 = K.prev_sn[slot][0]; c5 = K.prev_cs[slot][1]; s5 = K.prev_sn[slot][1];
        c6 = K.prev_cs[slot][2]; s6 = K.prev_sn[slot][2];
    }
    RSIK_MARK("disc_safety");
    RSIK_DISC_PROBE(6);
    RSIK_DISC_PRIO(0);
    int em = safety_checks(A.utab, jv, c4, s4, c5, s5, c6, s6, prev, K.max_angle, K.cos_max, K.sin_max);
    if (RSIK_RARE(invalid)) {  // no joints, no verdict on them (the reference has raised, C:215 / S:580)
#pragma unroll
        for (int k = 0; k < 7; k++) jv[k] = __builtin_nan("");
        em = 0;
    }
    RSIK_MARK("disc_store");
    RSIK_DISC_PROBE(4);
#ifdef RSIK_DISC_CLASS_PROBE
    // diagnostic build only (scripts/probes/disc_sorted_bound.py): the pose's path through the kernel instead of the emergency bits —
    // 16 the grid search was needed (the preferred-theta shortcut failed), 32 a theta was found, 64 it took the cooperative sweep
    em = (need ? 16 : 0) | (found ? 32 : 0) | (coop ? 64 : 0);
#endif
    // (written through: the launch is one round of waves, its end is a fifth of it — rsik_kernel_solve.hpp, kStoreThrough)
    store_rows<7, kStoreThrough>(K.joints, wave_base, K.n, lane, &lds_slab[wave][0][0], jv);
    if (live) {
        if (K.reachable) st_stream<kStoreThrough>(K.reachable + i, (uint8_t)(found ? 1 : 0));
        if (K.state) st_stream<kStoreThrough>(K.state + i, (uint8_t)st_code);
        if (K.emergency) st_stream<kStoreThrough>(K.emergency + i, (uint8_t)em);  // RSIK_EMERGENCY_* cause bits
    }
#ifdef RSIK_TIMELINE_PROBE
    // diagnostic build only (scripts/disc_timeline_probe.py): lane 0 of every wave overwrites its joints row with the six
    // 100 MHz stamps (start, inputs + tables in, reach + shortcut done, theta chosen, joints + safety done, stores
    // acknowledged) and the hardware id; lane 1 its row's first entry with the XCC id
    __builtin_amdgcn_s_waitcnt(0);
    RSIK_DISC_PROBE(5);
    if (live && lane == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        for (int k = 0; k < 6; k++) K.joints[i * 7 + k] = (double)probe_t[k];
        K.joints[i * 7 + 6] = (double)hw;
    }
    if (live && lane == 1) {
        K.joints[i * 7] = (double)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
        K.joints[i * 7 + 1] = (double)probe_t[6];
    }
#endif
}

}  // namespace rsik
