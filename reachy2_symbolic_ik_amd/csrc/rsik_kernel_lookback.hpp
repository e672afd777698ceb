// rsik_kernel_lookback.hpp — rsik_control_continuous_run: the joints phase that finishes its rows itself
// (one translation unit: included by rsik_lib.hip behind rsik_kernel_pipeline.hpp, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// The recurrence on previous_sol (allow_multiturn U:493-505, multiturn_safety_check U:535-568, continuity_check
// U:571-589, the emergency latch C:205-210, C:398-405) used to be two more kernels behind the joints phase — a sequential
// one that walked the chunks' first / last rows and one that added the whole turns it found — and two more hand-overs
// between dependent launches, 15-40 us each on this runtime.  Here a wave of the joints phase (a CHUNK: 8 consecutive
// steps of 8 neighbouring trajectories, lane = 8 * step + trajectory) learns the state at the end of the chunk before
// it from the wave that computes that chunk, while both are in flight — a decoupled look-back over chunks, per
// trajectory — and writes its rows once, final.
//
// Per (chunk, trajectory) four 64-bit words in the run's workspace (zeroed by the prepare phase of the chunk's block),
// each written whole by one agent-scope store, so that each is valid by itself and nothing needs a fence:
//   word 0      status + four signed bytes (joints 0, 2, 4, 6 — the ones whose raw angle has a branch cut, see
//               cont_joints_kernel):
//                 A   the chunk is QUIET (every step, the one from the chunk before included, stays within the continuity
//                     thresholds less a margin, no exact singularity): bytes = the whole turns its last step is away from
//                     the last step of the chunk before; bits 40-41: how far a limited joint (0, 2, 6) gets from the turn
//                     it came with, at any of its steps (0, 1, 2 = more)
//                 P   the chunk is decided: bytes = the whole turns between its last step's raw joints and previous_sol after
//                     it; bit 3: all within +-1 (limited joints) / +-90; bit 4: its rows have arrived in memory (only chunks
//                     that were walked wait for that before they publish)
//                 Pe  decided, rows in memory, but previous_sol after it is not "raw joints + whole turns" (a held, clamped
//                     or recomputed step): the next chunk reads the row itself
//                 E   final, the trajectory is latched (C:205-210): previous_sol is rows 1-7 of cont_state
//   words 1-3   the raw joints of the chunk's last step to 1e-4 rad (7 x 18 bits), bit 63 set, bit 56: not usable (an
//               exact singularity) — published as soon as they are computed
// A wave publishes words 1-3, reads those of the chunk before, decides QUIET or not, publishes A if so, then looks back:
// chunks c-1 ... c-8 at once (lane = 8 * distance + trajectory), further in steps of eight, adding up the A's until it finds
// a P.  Then previous_sol before its first step is "raw + whole turns" with known turns, its own rows are raw + 2 pi (turns
// + its in-chunk prefix), and it publishes P.  That holds as long as the +-6 pi clamps (U:535-568) cannot have bound on the
// way: the P within +-1 turn and at most one more turn of a limited joint since, own steps included (+-2 turns are 5 pi + the
// frame offset at most); otherwise the wave waits for a nearer P.  A chunk that is not quiet — or sits behind a Pe, or
// right behind a P two turns out — waits for the chunk before it to be final,
// reads that chunk's last row and walks its eight steps with the reference's own sequence of operations, eight lanes per
// trajectory (lane 8 * joint + trajectory), exactly what the sequential phase did for such chunks.
// Forward progress: a chunk's waves wait only for chunks with a lower workgroup index, and the hardware starts the
// workgroups of a launch in index order on each XCD: the lowest unfinished workgroup is always running and waits for nobody
// (a ticket counter instead of the index was measured: 16 000 atomic increments of one word cost 60 us per pass).  Every wait
// is bounded anyway (a trajectory whose wait runs out is latched with cause bit RSIK_EMERGENCY_INTERNAL).
// ------------------------------------------------------------------------------------------
// resident waves per SIMD the joints phase is compiled for (its register budget; the out-of-line redo path spills to fit)
#ifndef RSIK_LB_WAVES
#define RSIK_LB_WAVES 5
#endif
constexpr unsigned kLbA = 1, kLbP = 2, kLbPe = 3, kLbE = 4;
constexpr int kLbSpinLimit = 1 << 18;
constexpr double kLbQuantum = 1e-4, kLbMargin = 2e-4;

__device__ __forceinline__ unsigned long long lb_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double lb_load_f64(const double* p) {
    return __builtin_bit_cast(double, lb_load(reinterpret_cast<const unsigned long long*>(p)));
}
__device__ __forceinline__ void lb_store_f64(double* p, double v) {
    lb_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v));
}
__device__ __forceinline__ unsigned long long lb_word0(unsigned status, bool bit3, int t0, int t2, int t4, int t6) {
    return (unsigned long long)status | (bit3 ? 8ull : 0ull) | ((unsigned long long)(unsigned char)t0 << 8) |
           ((unsigned long long)(unsigned char)t2 << 16) | ((unsigned long long)(unsigned char)t4 << 24) |
           ((unsigned long long)(unsigned char)t6 << 32);
}
__device__ __forceinline__ int lb_byte(unsigned long long w, int k) { return (int)(signed char)((w >> (8 + 8 * k)) & 0xff); }
__device__ __forceinline__ unsigned long long lb_shfl64(unsigned long long v, int src_lane) {
    const int lo_ = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)v);
    const int hi_ = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)(v >> 32));
    return ((unsigned long long)(unsigned)hi_ << 32) | (unsigned)lo_;
}
__device__ __forceinline__ double lb_shfl_f64(double v, int src_lane) {
    return __builtin_bit_cast(double, lb_shfl64(__builtin_bit_cast(unsigned long long, v), src_lane));
}
// stores that another XCD may read while this kernel runs: written through (sc1)
__device__ __forceinline__ void st_row_f64_agent(__amdgpu_buffer_rsrc_t buf, unsigned lane, unsigned row, double v) {
#ifdef RSIK_LB_PLAIN_STORES
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(RowWords2, v), buf, lane, row, 0);
#else
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(RowWords2, v), buf, lane, row, 16);
#endif
}

// One step's get_joints out of line (the step kernel's own code), joint jj of it: with previous_sol = pv for a step that hit an
// exact singularity; or `as_first_pass`, the raw joints exactly as the chunk's own wave computed them.  See the calls.
template <bool MIXED>
__device__ __noinline__ double lb_redo_step(const ContRunArgs* Kp, SharedTables* tab, int64_t ts, int64_t ii, int jj, bool as_first_pass,
                                            double p0, double p1, double p2, double p3, double p4, double p5, double p6) {
    const ContRunArgs& K = *Kp;
    const double pv[7] = {p0, p1, p2, p3, p4, p5, p6};
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[ii] != 0) : false, *tab);
    const int f = K.flags[ts * K.n + ii];
    Reach r;
    Goal G;
    double mm[12];
    load_step_m12(K, ts, ii, mm);
    const bool plain = as_first_pass && (f & 8) == 0;
    if (plain) {
        mm[6] = fma(mm[1], mm[5], -(mm[2] * mm[4]));
        mm[7] = fma(mm[2], mm[3], -(mm[0] * mm[5]));
        mm[8] = fma(mm[0], mm[4], -(mm[1] * mm[3]));
    }
    step_geometry(A, mm, K.euler_roundtrip, (f & 1) == 0, r, G, plain);
    double jr[7];
    bool sing2;
    step_joints(A, K, r, G, RSIK_WS(K, ts, ii), pv, jr, sing2);
    double mine = jr[0];
#pragma unroll
    for (int k = 1; k < 7; k++) mine = (jj == k) ? jr[k] : mine;
    return mine;
}

#ifdef RSIK_LB_STATS
__device__ unsigned long long g_lb_stats[16];
#define RSIK_LB_COUNT(k, v) atomicAdd(&g_lb_stats[k], (unsigned long long)(v))
#else
#define RSIK_LB_COUNT(k, v)
#endif
template <bool MIXED>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RSIK_LB_WAVES, RSIK_LB_WAVES))) void cont_joints_lb_kernel(const ContRunArgs K) {
    RSIK_PIPE_STAMP(K, 2);
    static_assert(kJointChunk == 8, "lane = 8 * step + trajectory");
    __shared__ double lds_out[kBlock / 64][64 * 7];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tl = lane & 7, sl = lane >> 3;
    const int64_t n = K.n;
    // workgroups in chunk-major order: the chunk before is always a lower index
    const unsigned groups_wg = (unsigned)((n + 8 * (kBlock / 64) - 1) / (8 * (kBlock / 64)));
    const int64_t c = blockIdx.x / groups_wg;
    const int64_t grp = (int64_t)(blockIdx.x - (unsigned)c * groups_wg) * (kBlock / 64) + wave;  // this wave's group of 8 trajectories
    const int64_t i = grp * 8 + tl;
    const int64_t t = c * kJointChunk + sl;
    const int steps_left = (int)((K.T - c * kJointChunk) < kJointChunk ? (K.T - c * kJointChunk) : kJointChunk);
    const int last = steps_left - 1;  // the lane row of the chunk's last step
    const bool traj = i < n;          // this lane's trajectory exists
    const bool live = traj && sl <= last;
    const int64_t ii = traj ? i : (n - 1);
    const int64_t tt = t < K.T ? t : (K.T - 1);
    const int64_t cg = K.chunk0 + c;  // the chunk's index in the run
    const bool last_chunk = K.last_block && (c + 1) * kJointChunk >= K.T;
    double m[12];
    {
        const double* src = K.m12_steps + (K.t0 + tt) * 12 * n + ii;
#pragma unroll
        for (int k = 0; k < 6; k++) m[k] = src[k * n];
#pragma unroll
        for (int k = 9; k < 12; k++) m[k] = src[k * n];
    }
    const double theta = RSIK_WS(K, tt, ii);
    const int flag = K.flags[tt * n + ii];
    const bool special = (flag & 8) != 0;
    if (RSIK_RARE(special)) {
        const double* src = K.m12_steps + (K.t0 + tt) * 12 * n + ii;
#pragma unroll
        for (int k = 6; k < 9; k++) m[k] = src[k * n];
    } else {
        m[6] = fma(m[1], m[5], -(m[2] * m[4]));
        m[7] = fma(m[2], m[3], -(m[0] * m[5]));
        m[8] = fma(m[0], m[4], -(m[1] * m[3]));
    }
    const bool lane_isl = MIXED ? (K.arm[ii] != 0) : false;
    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(ContRunArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, lane_isl, lds_tab);
    double jv[7];
    bool sing;
    {
        Reach r;
        Goal G;
        step_geometry(A, m, K.euler_roundtrip, (flag & 1) == 0, r, G, !special);
        const double zeros[7] = {0, 0, 0, 0, 0, 0, 0};
        step_joints(A, K, r, G, theta, zeros, jv, sing);
    }
    const unsigned long long tmask = 0x0101010101010101ull << tl;  // the lanes of this lane's trajectory
    unsigned long long* const lb = K.lb;
    // ---- words 1-3: the raw joints of the chunk's last step, as soon as they exist
    if (traj && sl == last) {
        bool usable = !sing;
        unsigned long long q[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            usable = usable && (fabs(jv[k]) <= 13.0);  // (raw angles are atan2 values plus the arm's frame offsets)
            q[k] = (unsigned long long)((unsigned)(int)rint(jv[k] * (1.0 / kLbQuantum)) & 0x3ffffu);
        }
        const unsigned long long tag = (0x80ull | (usable ? 0ull : 1ull)) << 56;
        lb_store(lb + (cg * 4 + 1) * n + i, tag | q[0] | (q[1] << 18) | (q[2] << 36));
        lb_store(lb + (cg * 4 + 2) * n + i, tag | q[3] | (q[4] << 18) | (q[5] << 36));
        lb_store(lb + (cg * 4 + 3) * n + i, tag | q[6]);
    }
    // ---- steps relative to the step before, inside the chunk (lane - 8): whole turns of the four joints with a branch
    // cut as base-256 digits (8 + turn), the largest remaining step against the continuity thresholds (C:398)
    const int below = (sl == 0 ? lane : lane - 8) << 2;
    auto from_below = [&](double v) {
        const int lo_ = __builtin_amdgcn_ds_bpermute(below, (int)__double2loint(v));
        const int hi_ = __builtin_amdgcn_ds_bpermute(below, (int)__double2hiint(v));
        return __hiloint2double(hi_, lo_);
    };
    double worst_a = 0.0, worst_b = 0.0;
    double packed = 0.0;
#pragma unroll
    for (int k = 6; k >= 0; k--) {
        const double d = jv[k] - from_below(jv[k]);
        double x = d;
        if (k == 0 || k == 2 || k == 4 || k == 6) {
            const double r = rint(d * 0.15915494309189535);
            x = fma(-r, kTwoPi, d);
            packed = fma(packed, 256.0, 8.0 - r);
        }
        if (k < 4) worst_a = __builtin_fmax(worst_a, fabs(x));
        else worst_b = __builtin_fmax(worst_b, fabs(x));
    }
    const unsigned long long sing_mask = __ballot(sing && live);
    const bool sing_below = sl > 0 && ((sing_mask >> (lane - 8)) & 1ull) != 0;
    const bool ev_in = sing || sing_below || !(worst_a <= 0.5 - 1e-9) || !(worst_b <= 1.0 - 1e-9) || !(fabs(packed) < 4.0e9);
    const bool event_inside = (__ballot(ev_in && live) & tmask) != 0;
    unsigned word = (unsigned)packed;
#pragma unroll
    for (int step = 1; step < 8; step *= 2) {
        const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute((lane - 8 * step) << 2, (int)word);
        if (sl >= step) word += w;
    }
    int t_rel[4];  // whole turns of joints 0, 2, 4, 6 at this step relative to the last step of the chunk before (once the boundary is known)
    {
        const int bias = 8 * (sl + 1);
        t_rel[0] = (int)(word & 0xffu) - bias;
        t_rel[1] = (int)((word >> 8) & 0xffu) - bias;
        t_rel[2] = (int)((word >> 16) & 0xffu) - bias;
        t_rel[3] = (int)(word >> 24) - bias;
    }
    // ---- the chunk before: its words 1-3 (the step into this chunk) and, in the same round trips, the look-back
    bool have_boundary = cg == 0;   // (per trajectory, the same in its eight lanes)
    bool eventful = true;           // the run's first chunk is walked from cont_state
    int sums[4] = {0, 0, 0, 0};     // of the chunk: the last step's turns
    int extent = 0;                 // how far a LIMITED joint (0, 2, 6) gets from the turn it came with, at any step: 0, 1, or 2 = more
    unsigned long long before = kLbPe;  // word 0 of what the look-back ended on
    int T0[4] = {0, 0, 0, 0};       // whole turns of previous_sol before the first step relative to the raw joints of the step it belongs to
    bool timed_out = false;
#ifdef RSIK_LB_STATS
    int lb_spins_look = 0;
#endif
    if (cg > 0) {
        bool done = !traj;
        int64_t base = cg - 1;
        bool strict = false;          // only the chunk right before will do (or a latch anywhere)
        int run[4] = {0, 0, 0, 0};   // turns and extent of the quiet chunks already passed, window by window
        int run_extent = 0;
        const unsigned long long* raw_src = lb + ((cg - 1) * 4 + 1) * n + ii;
        int spins = 0;
        while (__any(!done)) {
            const bool want_raw = traj && sl == 0 && !have_boundary;
            unsigned long long r1 = 0, r2 = 0, r3 = 0;
            if (want_raw) {
                r1 = lb_load(raw_src);
                r2 = lb_load(raw_src + n);
                r3 = lb_load(raw_src + 2 * n);
            }
            const int64_t cc = base - sl;
            const unsigned long long w = (!done && cc >= 0) ? lb_load(lb + (cc * 4) * n + ii) : 0ull;
            const bool raw_in = want_raw && ((r1 & r2 & r3) >> 63) != 0;
            if (__any(raw_in)) {
                // the step from the chunk before into this one, by the lane of the chunk's first step
                bool quiet_in = false;
                int bd[4] = {0, 0, 0, 0};
                if (raw_in) {
                    const bool usable = (((r1 | r2 | r3) >> 56) & 1ull) == 0;
                    auto field = [](unsigned long long v, int at) { return (double)(((int)((unsigned)(v >> at) << 14)) >> 14); };  // 18 bits, signed
                    const double pq[7] = {field(r1, 0), field(r1, 18), field(r1, 36), field(r2, 0), field(r2, 18), field(r2, 36), field(r3, 0)};
                    bool ok = usable && !sing;
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        const double d = jv[k] - pq[k] * kLbQuantum;
                        double x = d;
                        if (k == 0 || k == 2 || k == 4 || k == 6) {
                            const double r = rint(d * 0.15915494309189535);
                            x = fma(-r, kTwoPi, d);
                            bd[k >> 1] = -(int)r;
                            ok = ok && fabs(r) <= 2.0;
                        }
                        ok = ok && (fabs(x) <= (k < 4 ? 0.5 : 1.0) - kLbMargin);
                    }
                    quiet_in = ok;
                }
                // first-step lane -> every lane of the trajectory
                const unsigned long long m_in = __ballot(raw_in);
                const bool mine_in = ((m_in >> tl) & 1ull) != 0;
                unsigned bw = ((unsigned)(bd[0] & 0xff)) | ((unsigned)(bd[1] & 0xff) << 8) | ((unsigned)(bd[2] & 0xff) << 16) | ((unsigned)(bd[3] & 0xff) << 24);
                bw = (unsigned)__builtin_amdgcn_ds_bpermute(tl << 2, (int)bw);
                const bool boundary_quiet = ((__ballot(quiet_in) >> tl) & 1ull) != 0;
                if (mine_in) {
#pragma unroll
                    for (int k = 0; k < 4; k++) t_rel[k] += (int)(signed char)((bw >> (8 * k)) & 0xff);
                }
                int s_[4];
#pragma unroll
                for (int k = 0; k < 4; k++) s_[k] = __builtin_amdgcn_ds_bpermute((8 * last + tl) << 2, t_rel[k]);
                const bool far = (__ballot(live && (__builtin_abs(t_rel[0]) > 1 || __builtin_abs(t_rel[1]) > 1 || __builtin_abs(t_rel[3]) > 1)) & tmask) != 0;
                const bool some = (__ballot(live && (t_rel[0] | t_rel[1] | t_rel[3]) != 0) & tmask) != 0;
                if (mine_in) {
#pragma unroll
                    for (int k = 0; k < 4; k++) sums[k] = s_[k];
                    extent = far ? 2 : (some ? 1 : 0);
                    const bool sums_fit = __builtin_abs(sums[0]) <= 7 && __builtin_abs(sums[1]) <= 7 && __builtin_abs(sums[2]) <= 7 && __builtin_abs(sums[3]) <= 7;
                    eventful = event_inside || !boundary_quiet || !sums_fit;
                    have_boundary = true;
                    strict = eventful;
                    run_extent = extent;
                    if (traj && sl == 7 && !eventful)
                        lb_store(lb + (cg * 4) * n + i, lb_word0(kLbA, false, sums[0], sums[1], sums[2], sums[3]) | ((unsigned long long)extent << 40));
                }
            }
            // the look-back window: the nearest decided chunk and the quiet ones before it
            const unsigned stt = (unsigned)(w & 7);
            const unsigned long long m_final = __ballot(stt >= kLbP), m_quiet = __ballot(stt == kLbA);
            const unsigned long long bits_final = (m_final >> tl) & 0x0101010101010101ull;
            const int ks = bits_final != 0 ? (__builtin_ctzll(bits_final) >> 3) : 8;  // distance - 1 of the nearest decided chunk in the window
            const unsigned long long need = (ks == 8 ? ~0ull : ((1ull << (8 * ks)) - 1ull)) & 0x0101010101010101ull;
            const bool all_quiet = ((m_quiet >> tl) & need) == need;
            const unsigned long long wt = lb_shfl64(w, 8 * (ks < 8 ? ks : 7) + tl);
            // the quiet chunks nearer than that one, added up: four turn sums (each + 8: digits of a base-256 number) and the extents
            unsigned long long acc = (sl < ks && stt == kLbA) ? ((((w >> 8) & 0xffffffffull) ^ 0x80808080ull) - 0x78787878ull) | (((w >> 40) & 3ull) << 32) : 0ull;
            acc += lb_shfl64(acc, lane ^ 8);
            acc += lb_shfl64(acc, lane ^ 16);
            acc += lb_shfl64(acc, lane ^ 32);
            if (!done && have_boundary) {
                const unsigned ts = (unsigned)(wt & 7);
                int win[4];
#pragma unroll
                for (int k = 0; k < 4; k++) win[k] = (int)((acc >> (8 * k)) & 0xff) - 8 * ks;
                const int win_extent = (int)((acc >> 32) & 0xff);
                if (ks < 8 && all_quiet) {
                    const bool direct = ks == 0 && base == cg - 1;
                    // (with the nearest decided chunk within +-1 turn and at most one more turn on the way, a limited joint stays
                    // within +-2 turns = 5 pi + the frame offset of the +-6 pi clamps: none of the chunks passed can have walked)
                    if (ts == kLbE || direct || (!strict && ts == kLbP && (wt & 8) != 0 && run_extent + win_extent <= 1)) {
                        before = wt;
#pragma unroll
                        for (int k = 0; k < 4; k++) T0[k] = lb_byte(wt, k) + run[k] + win[k];
                        done = true;
                    }
                } else if (ks == 8 && all_quiet && !strict) {
                    if (run_extent + win_extent <= 1) {
                        base -= 8;
                        run_extent += win_extent;
#pragma unroll
                        for (int k = 0; k < 4; k++) run[k] += win[k];
                    } else {
                        strict = true;  // too much turning on the way: wait for the chunk right before
                        base = cg - 1;
#pragma unroll
                        for (int k = 0; k < 4; k++) run[k] = 0;
                    }
                }
            }
            if (++spins > kLbSpinLimit) {
                timed_out = !done;
                break;
            }
            if (__any(!done)) __builtin_amdgcn_s_sleep(1);
        }
#ifdef RSIK_LB_STATS
        lb_spins_look = spins;
#endif
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) sums[k] = __builtin_amdgcn_ds_bpermute((8 * last + tl) << 2, t_rel[k]);
    }
    // ---- what this chunk is: 0 rows = raw + whole turns; 1 walked step by step; 2 latched before it began; 3 gave up
    int mode;
    const int64_t t_abs = K.t0 + t;
    const unsigned before_status = (unsigned)(before & 7);
    {
        bool walk = eventful || before_status == kLbPe;
        if (!walk && before_status == kLbP && (before & 8) == 0) {
            // right behind a chunk with a limited joint two turns out, or wrist roll ninety: the reference's clamps
            // (U:535-568) could bind — tested on the first step with the slack the other steps can use up, like the
            // sequential phase's shortcut did
            bool clear = true;
            if (sl == 0) {
                const double clear_of_limit = 6 * kPi - (kJointChunk - 1) * 1.0 - 1e-6;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double turns = (double)(T0[k] + t_rel[k]);
                    if (k != 2) clear = clear && fabs(fma(turns, kTwoPi, jv[2 * k])) <= clear_of_limit;
                    clear = clear && fabs(turns) <= 90.0;
                }
            }
            walk = ((__ballot(!clear && sl == 0) >> tl) & 1ull) != 0;
        }
        mode = before_status == kLbE ? 2 : (walk ? 1 : 0);
        if (cg == 0 && K.st[9 * n + ii] != 0.0) mode = 2;  // latched when the run began
        if (timed_out) mode = 3;
    }
    int t_end[4];
#pragma unroll
    for (int k = 0; k < 4; k++) t_end[k] = T0[k] + sums[k];
    auto publish = [&](unsigned status, bool rows_in_memory) {
        const bool within = __builtin_abs(t_end[0]) <= 1 && __builtin_abs(t_end[1]) <= 1 && __builtin_abs(t_end[3]) <= 1 && __builtin_abs(t_end[2]) <= 90;
        const bool fits = __builtin_abs(t_end[0]) <= 100 && __builtin_abs(t_end[1]) <= 100 && __builtin_abs(t_end[2]) <= 100 && __builtin_abs(t_end[3]) <= 100;
        const unsigned so = (status == kLbP && !fits) ? kLbPe : status;
        lb_store(lb + (cg * 4) * n + i, lb_word0(so, within, fits ? t_end[0] : 0, fits ? t_end[1] : 0, fits ? t_end[2] : 0, fits ? t_end[3] : 0) |
                                            (rows_in_memory ? 16ull : 0ull));
    };
    // a chunk that stands as raw + whole turns, or repeats the latched previous_sol, is decided now: the chunks behind it need
    // its turns, not its rows (a walked chunk behind it recomputes the one raw row it needs)
    if (traj && sl == 7 && (mode == 0 || mode == 2)) publish(mode == 0 ? kLbP : kLbE, false);
#ifdef RSIK_LB_STATS
    if (traj && sl == 7) {
        RSIK_LB_COUNT(0, 1); RSIK_LB_COUNT(1, mode == 0); RSIK_LB_COUNT(2, mode == 1); RSIK_LB_COUNT(3, mode >= 2);
        RSIK_LB_COUNT(4, eventful); RSIK_LB_COUNT(6, extent != 0);
        RSIK_LB_COUNT(7, before_status == kLbPe); RSIK_LB_COUNT(8, (before & 8) == 0); RSIK_LB_COUNT(9, event_inside);
    }
    if (lane == 0) { RSIK_LB_COUNT(10, 1); RSIK_LB_COUNT(12, lb_spins_look); }
#endif
    double* lw = lds_out[wave];
    if (mode == 0) {
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const bool turning = k == 0 || k == 2 || k == 4 || k == 6;
            lw[lane * 7 + k] = turning ? fma((double)(T0[k >> 1] + t_rel[k >> 1]), kTwoPi, jv[k]) : jv[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 7; k++) lw[lane * 7 + k] = mode == 1 ? jv[k] : __builtin_nan("");
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    // ---- modes 1 to 3: lane (j, trajectory) owns joint j of the trajectory (j = 7: its bookkeeping)
    const int j = sl, jj = j < 7 ? j : 6;
    const bool owner = traj && j < 7;
    unsigned status_out = kLbP;
    if (RSIK_RARE(__any(mode != 0))) {
        if (mode == 2) {
            // latched: every step returns previous_sol (C:205-210), state and flag say so
            const double prev = lb_load_f64(K.st + (1 + jj) * n + ii);
            if (owner)
                for (int s = 0; s <= last; s++) lw[(8 * s + tl) * 7 + j] = prev;
            if (live) {
                if (K.state) K.state[t_abs * n + i] = (uint8_t)RSIK_STATE_EMERGENCY;
                if (K.reachable) K.reachable[t_abs * n + i] = 0;
            }
            status_out = kLbE;
        } else if (mode == 3) {
            if (traj && j == 7) {
                lb_store_f64(K.st + 9 * n + i, 1.0);
                lb_store_f64(K.st + 11 * n + i, (double)RSIK_EMERGENCY_INTERNAL);
            }
            if (live) {
                if (K.state) K.state[t_abs * n + i] = (uint8_t)RSIK_STATE_EMERGENCY;
                if (K.reachable) K.reachable[t_abs * n + i] = 0;
            }
            status_out = kLbE;
        }
        if (__any(mode == 1)) {
            const bool mine = mode == 1;
            const ContRunArgs* const Kp = reinterpret_cast<const ContRunArgs*>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
            double prev = 0.0;
            bool init = false, emergency = false;
            // previous_sol before the chunk's first step: cont_state at the start of the run; the last row of the chunk before
            // where that has arrived in memory (it was walked, or an earlier launch wrote it); else raw joints + whole turns, the
            // raw joints recomputed as that chunk's wave computed them
            const bool from_row = cg > 0 && (c == 0 || (before & 16) != 0);
            if (mine) {
                if (cg == 0) {
                    prev = K.st[(1 + jj) * n + ii];
                    init = K.st[8 * n + ii] != 0.0;
                } else if (from_row) {
                    prev = lb_load_f64(K.joints + ((K.t0 + c * kJointChunk - 1) * n + ii) * 7 + jj);
                }
            }
            if (__any(mine && cg > 0 && !from_row)) {
                const double raw = lb_redo_step<MIXED>(Kp, &lds_tab, c * kJointChunk - 1, ii, jj, true, 0, 0, 0, 0, 0, 0, 0);
                const bool turning = jj == 0 || jj == 2 || jj == 4 || jj == 6;
                if (mine && cg > 0 && !from_row) prev = turning ? fma((double)T0[jj >> 1], kTwoPi, raw) : raw;
            }
            const double thr = jj < 4 ? 0.5 : 1.0;                                          // continuity thresholds, C:398
            const double lim = (jj == 0 || jj == 2 || jj == 6) ? 6 * kPi : __builtin_inf();  // multiturn limit of this lane's joint (U:535-568)
            const int hit_bit = jj == 0 ? RSIK_EMERGENCY_SHOULDER_PITCH : (jj == 2 ? RSIK_EMERGENCY_ELBOW_YAW : RSIK_EMERGENCY_WRIST_YAW);
            auto group_or = [&](int v) -> int {  // OR over the 8 lanes of a trajectory (stride 8), left in every one of them
                v |= __builtin_amdgcn_ds_bpermute((lane ^ 8) << 2, v);
                v |= __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, v);
                v |= __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, v);
                return v;
            };
            bool normal_end = false;
            double raw_last = 0.0;
#pragma unroll 1
            for (int s = 0; s <= last; s++) {
                const int slot = (8 * s + tl) * 7 + jj;
                double cur = lw[slot];
                const bool redo = mine && !emergency && ((sing_mask >> (8 * s + tl)) & 1ull) != 0;
                if (RSIK_RARE(__any(redo))) {
                    // exact singularity in get_joints (S:751-753, 782-784): the step is recomputed with the real previous_sol
                    // (every lane of the trajectory computes all seven joints and keeps its own) — out of line, through the
                    // kernel-argument segment, so that the second copy of get_joints does not set this kernel's register count
                    double pv[7];
#pragma unroll
                    for (int k = 0; k < 7; k++) pv[k] = lb_shfl_f64(prev, 8 * k + tl);
                    const double again = lb_redo_step<MIXED>(Kp, &lds_tab, c * kJointChunk + s, ii, jj, false, pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], pv[6]);
                    cur = redo ? again : cur;
                }
                raw_last = cur;
                const double turned = allow_multiturn_one_straight(cur, prev);        // U:493-505
                const double clamped = fmin(fmax(turned, -lim), lim);                 // U:535-568 (lim = inf for joints 1, 3, 4, 5)
                int code = (mine && clamped != turned && j < 7) ? hit_bit : 0;
                // U:571-589: |angle_diff(joint, previous)| against the joint's threshold, on the limited value like the reference
                code |= (mine && j < 7 && fabs(angle_diff_straight(clamped, prev)) > thr) ? 16 : 0;
                code = group_or(code);
                const bool disc = !init && (code & 16) != 0;
                const int cause = (code & 7) | (disc ? RSIK_EMERGENCY_CONTINUITY : 0);
                const double accepted = disc ? prev : clamped;
                const bool trips = cause != 0 && !emergency;
                const double result = emergency ? prev : accepted;                    // latched (C:205-210): previous_sol
                if (mine && owner) lw[slot] = result;
                if (RSIK_RARE(mine && (emergency || trips)) && traj) {
                    const int64_t ta = K.t0 + c * kJointChunk + s;
                    if (emergency) {
                        if (j == 7) {
                            if (K.state) K.state[ta * n + i] = (uint8_t)RSIK_STATE_EMERGENCY;
                            if (K.reachable) K.reachable[ta * n + i] = 0;
                        }
                    } else {
                        if (j == 7) {
                            K.st[11 * n + i] = (double)cause;
                            K.st[0 * n + i] = RSIK_WS(K, c * kJointChunk + s, i);  // previous_theta of the step that tripped (phase 2 ran ahead)
                            lb_store_f64(K.st + 9 * n + i, 1.0);
                        } else {
                            if (disc) K.st[(12 + j) * n + i] = clamped;             // the joints that failed the check
                            lb_store_f64(K.st + (1 + j) * n + i, prev);             // previous_sol for the latched chunks behind this one
                        }
                    }
                }
                normal_end = !emergency && !trips && (code & 7) == 0 && !disc && !redo;
                prev = (emergency || trips) ? prev : accepted;
                init = emergency ? init : false;
                emergency = emergency || trips;
            }
            if (mine) {
                // the whole turns between the last step's raw joint and previous_sol now (joints 0, 2, 4, 6 -> lanes 0, 2, 4, 6 of the trajectory)
                const double tq = rint((prev - raw_last) * 0.15915494309189535);
                const bool in_range = fabs(tq) <= 100.0;
                const int ti = in_range ? (int)tq : 0;
#pragma unroll
                for (int k = 0; k < 4; k++) t_end[k] = __builtin_amdgcn_ds_bpermute((16 * k + tl) << 2, ti);
                const bool ranges = (__ballot(mine && !in_range && owner && (j & 1) == 0) & tmask) == 0;
                status_out = emergency ? kLbE : ((normal_end && ranges) ? kLbP : kLbPe);
                if (cg == 0 && traj && j == 7) K.st[8 * n + i] = 0.0;  // (init is spent by the run's first step)
                if (last_chunk && owner && !emergency) K.st[(1 + j) * n + i] = prev;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the run's state after its last step (a latched trajectory keeps what the step that tripped left)
    if (last_chunk && traj && sl == last) {
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 7; k++) K.st[(1 + k) * n + i] = lw[lane * 7 + k];
        }
        if (status_out != kLbE) K.st[0 * n + i] = theta;  // previous_theta after the last step
    }
    // ---- rows out: the wave's 64 rows are 8 runs (one per step) of 8 x 7 consecutive doubles; 32-bit offsets from the chunk's
    // first row (a block's joints stay below 2 GB, see rsik_control_continuous_run); written through: a walked chunk behind
    // this one may read the last row from another XCD
    const int traj_left = (int)((n - grp * 8) < 8 ? (n - grp * 8) : 8);  // trajectories of this group that exist (<= 0 past the end)
    const __amdgpu_buffer_rsrc_t obuf = row_buffer(K.joints + ((K.t0 + c * kJointChunk) * n + grp * 8) * 7);
    const unsigned row_bytes = (unsigned)(n * 7 * sizeof(double));
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const int idx = k * 64 + lane;
        const int s_ = idx / 56, off = idx - s_ * 56;
        if (s_ < steps_left && off < traj_left * 7) st_row_f64_agent(obuf, (unsigned)s_ * row_bytes + (unsigned)off * 8u, 0, lw[idx]);
    }
    // ---- a walked chunk (or one that gave up) is decided once its rows, and the state rows above, have arrived
    if (RSIK_RARE(__any(mode == 1 || mode == 3))) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (traj && sl == 7 && (mode == 1 || mode == 3)) publish(status_out, true);
    }
}

}  // namespace rsik
