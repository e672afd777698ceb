// rsik_kernel_solve.hpp — rsik_solve: launch-argument blocks, constant access, table staging, row stores, solve_kernel, the goal-matrix conversion
// (one translation unit: included by rsik_lib.hip, in this order, inside nothing)
#pragma once

namespace rsik {

#ifndef RSIK_BLOCK
#define RSIK_BLOCK 256
#endif
constexpr int kBlock = RSIK_BLOCK;

struct SolveArgs {
    int64_t n;
    const double* in[6];
    const uint8_t* arm;
    int theta_policy;
    const double* theta_in;
    double prev[7];
    double* joints;
    double* interval;
    double* elbow;
    uint8_t* reachable;
    uint8_t* state;
    ArmC arms[2];  // uniform launch: arms[0] is the arm; mixed launch: arms[0] = r, arms[1] = l
};

// Per-arm constant access.  Uniform launches read the block from the kernarg segment (scalar loads).  Mixed r/l
// launches stage both blocks in LDS once per workgroup and every lane reads its own arm's value with one ds_read
// (selecting between two scalar values would cost two v_cndmask per use and spill the scalar file).
typedef const __attribute__((address_space(3))) double* LdsConst;
template <bool MIXED>
struct Acc {
    const ArmC* a;
    bool isl;
    LdsConst lds;  // MIXED only: this lane's arm block in LDS
    UnitAtanTab utab;  // LDS copy of the unit-vector atan2 table (rsik_math.hpp)
    __device__ __forceinline__ double operator()(int i) const {
        if constexpr (MIXED) return lds[i];
        else return a[0].v[i];
    }
};
// Same, with the uniform block addressed through an explicit kernarg-segment (constant address space) pointer.
typedef const __attribute__((address_space(4))) double* KConst;
// Entries of the constant block that can differ between a right and a left arm that are mirror images of each other
// (everything with a y component or a handedness: shoulder y, tip y, the shoulder frame, the elbow singularity y, the
// side sign, the projection plane).  In a mixed launch whose two blocks agree everywhere else (checked by the host:
// SolveArgs.mirror) only these come from the per-lane LDS copy; the rest are the same scalar loads as in a
// uniform launch (a mixed launch reads ~85 constants per wave, ~35 of them from this shared set).
__host__ __device__ constexpr bool arm_const_is_sided(int i) {
    return i == RSIK_C_SHOULDER + 1 || i == RSIK_C_TIPL + 1 || (i >= RSIK_C_MST && i < RSIK_C_TSH + 3) || i == RSIK_C_ES + 1 ||
           i == RSIK_C_SIDE || (i >= RSIK_C_PLANE_P && i < RSIK_C_PROJ_CENTER + 3);
}
// MIXED: 0 = one arm for the whole launch, 1 = per-lane arm, every constant from LDS, 2 = per-lane arm, mirrored blocks
template <int MIXED>
struct AccK {
    KConst k;
    LdsConst lds;
    UnitAtanTab utab;
    __device__ __forceinline__ double operator()(int i) const {
        if constexpr (MIXED == 1) return lds[i];
        else if constexpr (MIXED == 2) return arm_const_is_sided(i) ? lds[i] : k[i];
        else return k[i];
    }
};
// Workgroup-shared read-only data: the per-arm blocks (mixed launches) and the unit-vector atan2 table.
struct SharedTables {
    double arm[2][RSIK_ARM_CONSTS_COUNT];
    double utab[3][kUnitAtanRows];  // column-major, see unit_atan2_n
};
// The kernels read their ~1 KB argument block (pointers, launch constants, the arm constants) with scalar loads that the
// compiler places where the values are first needed — a dozen first touches of different 64-byte lines, spread over the
// whole kernel, and the scalar cache starts every launch cold: every wave of a launch's first round stalls on each of them
// (measured: RSIK_WARM_KERNARG 0 vs 1).  warm_kernarg<BYTES>() touches every line of the block once, after the wave
// has issued its input loads and the table-staging loads (stage_tables): the misses overlap each other and those loads'
// latency, later reads hit.
// (The values are discarded: all loads target one clobbered scalar register and are waited for inside the block.)
#ifndef RSIK_WARM_KERNARG
#define RSIK_WARM_KERNARG 1
#endif
template <int BYTES>
__device__ __forceinline__ void warm_kernarg() {
#if RSIK_WARM_KERNARG
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
#define RSIK_TOUCH(off) if constexpr (BYTES > (off)) asm volatile("s_load_dword s90, %0, " #off ::"s"(ka) : "s90", "memory")
    RSIK_TOUCH(0x40); RSIK_TOUCH(0x80); RSIK_TOUCH(0xc0); RSIK_TOUCH(0x100); RSIK_TOUCH(0x140); RSIK_TOUCH(0x180);
    RSIK_TOUCH(0x1c0); RSIK_TOUCH(0x200); RSIK_TOUCH(0x240); RSIK_TOUCH(0x280); RSIK_TOUCH(0x2c0); RSIK_TOUCH(0x300);
    RSIK_TOUCH(0x340); RSIK_TOUCH(0x380); RSIK_TOUCH(0x3c0); RSIK_TOUCH(0x400); RSIK_TOUCH(0x440); RSIK_TOUCH(0x480);
    RSIK_TOUCH(0x4c0); RSIK_TOUCH(0x500); RSIK_TOUCH(0x540); RSIK_TOUCH(0x580); RSIK_TOUCH(0x5c0); RSIK_TOUCH(0x600);
#undef RSIK_TOUCH
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "s90", "memory");
#endif
}

// All global reads of the staging are issued first and the LDS writes follow, so a workgroup pays ONE memory round trip
// before its barrier (a copy loop per table serialises one round trip per table: +0.6 us on every wave's start-up).
// WARM: bytes of the kernel's argument block to warm in the scalar cache (warm_kernarg) while the staging loads fly.
// NB: threads of the workgroup (a power of two)
template <bool MIXED, int WARM = 0, int NB = kBlock>
__device__ __forceinline__ void stage_tables(SharedTables& S, const ArmC* arms) {
    constexpr int NA = kUnitAtanRows * 3, NS = kSinCosRows * 2, NC = 2 * RSIK_ARM_CONSTS_COUNT;
    constexpr int RA = (NA + NB - 1) / NB, RS = (NS + NB - 1) / NB, RC = (NC + NB - 1) / NB;
    const unsigned t = threadIdx.x & (NB - 1);  // the launch uses NB threads: tells the compiler t < NB
    const double* ga = &c_unit_atan_tab[0][0];
    const double* gs = &c_sincos_tab[0][0];
    double va[RA], vs[RS], vc[RC];
    // chunk r of a table covers elements [r NB, (r+1) NB): only a table's last chunk can be partial
#pragma unroll
    for (int r = 0; r < RA; r++) va[r] = ((r + 1) * NB <= NA || t + r * NB < NA) ? ga[t + r * NB] : 0.0;
#pragma unroll
    for (int r = 0; r < RS; r++) vs[r] = ((r + 1) * NB <= NS || t + r * NB < NS) ? gs[t + r * NB] : 0.0;
    if constexpr (MIXED) {
#pragma unroll
        for (int r = 0; r < RC; r++) {
            const unsigned k = t + r * NB;
            vc[r] = ((r + 1) * NB <= NC || k < NC) ? arms[k / RSIK_ARM_CONSTS_COUNT].v[k % RSIK_ARM_CONSTS_COUNT] : 0.0;
        }
    }
        double* la = &S.utab[0][0];
    double* ls = &g_sincos_tab[0][0];
#pragma unroll
    for (int r = 0; r < RA; r++)
        if ((r + 1) * NB <= NA || t + r * NB < NA) la[t + r * NB] = va[r];
#pragma unroll
    for (int r = 0; r < RS; r++)
        if ((r + 1) * NB <= NS || t + r * NB < NS) ls[t + r * NB] = vs[r];
    if constexpr (MIXED) {
        double* lc = &S.arm[0][0];
#pragma unroll
        for (int r = 0; r < RC; r++)
            if ((r + 1) * NB <= NC || t + r * NB < NC) lc[t + r * NB] = vc[r];
    }
    __syncthreads();
}
template <bool MIXED>
__device__ __forceinline__ Acc<MIXED> make_acc(const ArmC* arms, bool isl, SharedTables& S) {
    Acc<MIXED> A{arms, isl, (LdsConst)S.arm[isl ? 1 : 0], (UnitAtanTab)&S.utab[0][0]};
    return A;
}

// Batch inputs are read once and outputs written once: streaming (non-temporal) accesses keep them from displacing
// each other in L2 and leave fewer dirty lines for the end-of-kernel write-back.
#ifndef RSIK_NT_STORE
#define RSIK_NT_STORE 1  // config 2: 45.2 -> 44.7 us per 1 M poses; non-temporal LOADS cost 0.5 us (inputs of back-to-back launches sit in the 256 MB Infinity Cache)
#endif
#ifndef RSIK_NT_LOAD
#define RSIK_NT_LOAD 0
#endif
typedef double f64x2 __attribute__((ext_vector_type(2)));
// kStoreStream: non-temporal.  kStoreThrough: written through to system scope (sc0 sc1) — nothing of the kernel's output is left dirty in
// the eight L2s for the write-back at its end, which the next launch of a stream (and every launch that waits for this one) sits behind.
// Round 6, same box, interleaved: the discrete kernel 15.0 -> 14.6 us per 262 144 matrices, the trajectory pipeline -1.3 ... -1.5 % per pass
// in every launch form (its kernels' ends are what its hand-overs wait for); the solve kernel, four rounds of waves long, 31.3 -> 31.7 us
// with it: that one keeps the non-temporal form (docs/experiments.md R6.8).
constexpr int kStoreStream = 1, kStoreThrough = 2;
template <int POLICY = kStoreStream, class T>
__device__ __forceinline__ void st_stream(T* p, T v) {
#if RSIK_NT_STORE
    if constexpr (POLICY == kStoreThrough && sizeof(T) <= 8) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <class T>
__device__ __forceinline__ T ld_stream(const T* p) {
#if RSIK_NT_LOAD
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// Writes ROWxW doubles per lane as a contiguous [64*W] slab per wave (row-major [n,W] output).
template <int W, int POLICY = kStoreStream>
__device__ __forceinline__ void store_rows(double* __restrict__ out, int64_t wave_base, int64_t n, int lane,
                                           double* __restrict__ lds_wave, const double (&vals)[W]) {
#pragma unroll
    for (int k = 0; k < W; k++) lds_wave[lane * W + k] = vals[k];
    // same-wave LDS exchange: the wave executes in lock-step, only the LDS counter must drain
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    int64_t rows = n - wave_base;
    if (rows > 64) rows = 64;
    double* dst = out + wave_base * W;
    if (__builtin_amdgcn_readfirstlane((int)rows) == 64) {  // every wave but the last: no per-row bounds test
        double v[W];
#pragma unroll
        for (int k = 0; k < W; k++) v[k] = lds_wave[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < W; k++) st_stream<POLICY>(dst + k * 64 + lane, v[k]);
    } else {
        const int64_t total = rows * W;
#pragma unroll
        for (int k = 0; k < W; k++) {
            int idx = k * 64 + lane;
            if (idx < total) st_stream<POLICY>(dst + idx, lds_wave[idx]);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// flush_rows for a wave whose 64 rows all exist (every wave but the last of a launch): the W row reads are issued
// together and the W stores share one base address, no per-row bounds test.
template <int W>
__device__ __forceinline__ void flush_rows_full(double* __restrict__ out, int64_t wave_base, int lane,
                                                const double* __restrict__ lds_rows) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    double v[W];
#pragma unroll
    for (int k = 0; k < W; k++) v[k] = lds_rows[k * 64 + lane];
    double* dst = out + wave_base * W + lane;
#pragma unroll
    for (int k = 0; k < W; k++) st_stream(dst + k * 64, v[k]);
    __builtin_amdgcn_wave_barrier();
}

// Second half of store_rows for values the lanes have already put in LDS.
template <int W>
__device__ __forceinline__ void flush_rows(double* __restrict__ out, int64_t wave_base, int64_t n, int lane,
                                           const double* __restrict__ lds_rows) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    int64_t rows = n - wave_base;
    if (rows > 64) rows = 64;
    const int64_t total = rows * W;
    double* dst = out + wave_base * W;
#pragma unroll
    for (int k = 0; k < W; k++) {
        int idx = k * 64 + lane;
        if (idx < total) st_stream(dst + idx, lds_rows[idx]);
    }
    __builtin_amdgcn_wave_barrier();
}

#ifndef RSIK_SOLVE_MIN_WAVES
#define RSIK_SOLVE_MIN_WAVES 1
#endif

// One workgroup = one tile of kBlock consecutive poses, one pose per lane.  Every global address is a scalar base
// (column pointer + tile offset, computed on the SALU) plus a small per-lane offset, so the six loads and all the
// stores share one or two address registers.  Lanes past the end of the batch recompute the last pose; their stores
// are masked.  (A persistent variant that walks several tiles per workgroup with the next tile prefetched was
// measured slower at every depth: 46.0 / 47.7 / 52.5 us for 2 / 4 / 8 tiles against 45.5 us, see
// profiles/r01/timeline/: under the power-managed clock it is the executed instruction count that sets the time, not
// how well the waves overlap.)
// TIPZ: every arm of the launch has tip_x = tip_y = 0 (goal_from_euler_tipz: -24 fp64 operations per pose).
template <int MIXED, bool TIPZ>
__global__ __launch_bounds__(kBlock, RSIK_SOLVE_MIN_WAVES) void solve_kernel(const SolveArgs K) {
    __shared__ SharedTables lds_tab;
    __shared__ double lds[kBlock / 64][64 * 10];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: wave-level tests stay on the SALU
    const int64_t tile0 = (int64_t)blockIdx.x * kBlock;
    const int64_t left = K.n - tile0;                                   // >= 1 (grid = ceil(n / kBlock))
    const unsigned rows = left < kBlock ? (unsigned)left : (unsigned)kBlock;
    const unsigned t = threadIdx.x & (kBlock - 1);                      // (tells the compiler t < kBlock)
    const unsigned tt = (t < rows ? t : rows - 1) & (kBlock - 1);       // clamped pose index inside the tile
    const bool live = t < rows;

#ifdef RSIK_CLOCK_PROBE
    const uint64_t probe_c0 = __builtin_readcyclecounter(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef RSIK_TIMELINE_PROBE
    const uint64_t probe_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    // the six pose loads are issued before the table staging so that their latency overlaps it
    double in[6];
#pragma unroll
    for (int k = 0; k < 6; k++) in[k] = ld_stream(K.in[k] + tile0 + tt);
    // (MIXED == 1 takes every constant from LDS: nothing to warm.  Here the warm-up goes BEFORE the staging loads are issued,
    // in the other kernels between their issue and their use (stage_tables<., WARM>): measured both ways per kernel, config
    // 2 31.1 vs 31.9 us, config 3 15.9 vs 15.7 us)
    warm_kernarg<(MIXED == 1 ? 0 : (int)offsetof(SolveArgs, arms) + (int)sizeof(ArmC))>();
    stage_tables<(MIXED != 0)>(lds_tab, K.arms);
#ifdef RSIK_TIMELINE_PROBE
    const uint64_t probe_t1 = __builtin_amdgcn_s_memrealtime();
    uint64_t probe_mid = 0, probe_goal = 0, probe_reach = 0;
#define RSIK_SOLVE_PROBE(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RSIK_SOLVE_PROBE(v) do { } while (0)
#endif
    const AccK<MIXED> A{(KConst)&((const __attribute__((address_space(4))) SolveArgs*)__builtin_amdgcn_kernarg_segment_ptr())->arms[0].v[0],
                        (LdsConst)lds_tab.arm[(MIXED != 0 && K.arm[tile0 + tt] != 0) ? 1 : 0], (UnitAtanTab)&lds_tab.utab[0][0]};
    double* lds_wave = lds[wave];

    // rsik.h "Rows that are not numbers": judged here, while the six values are at hand (further down it would keep them all alive)
    const bool invalid = !all_finite(in);
    const V3 pos = {in[0], in[1], in[2]};
    Goal G;
    if constexpr (TIPZ) {
        G = goal_from_euler_tipz(A, in[3], in[4], in[5]);
    } else {
        RSIK_MARK("euler");
        const Rot Rg = rot_from_euler(in[3], in[4], in[5]);
        RSIK_MARK("goal");
        G = make_goal(A, Rg);
    }
    RSIK_MARK("reach_start");
    RSIK_SOLVE_PROBE(probe_goal);
    Reach r = reach_g<false, false>(A, pos, G.woff);
    RSIK_SOLVE_PROBE(probe_reach);
    if (RSIK_RARE(invalid)) {  // where the reference raises (S:580) or projects an infinity
        r.ok = false;
        r.state = RSIK_STATE_INVALID_INPUT;
        r.i0 = r.i1 = __builtin_nan("");
    }
    RSIK_MARK("after_reach");

    // joints [64,7] and elbow [64,3] of the wave are staged in LDS (row-major, as they go to HBM) by whichever branch
    // the lane takes, then written out with coalesced rows: failed poses only cost their NaN fill when one exists
    if (K.theta_policy != RSIK_THETA_NONE) {
        double* jrow = lds_wave + lane * 7;
        double* erow = lds_wave + 64 * 7 + lane * 3;
        if (r.ok) {
            double ct = r.ct0, st = r.st0;  // theta = interval[0]: cos/sin come straight from the intersection point
            if (K.theta_policy != RSIK_THETA_INTERVAL0) {
                const double th_in = K.theta_in[tile0 + tt];
                double theta = th_in;
                if (K.theta_policy != RSIK_THETA_EXPLICIT) {
                    double a = r.i0, b = r.i1;
                    if (a > b) b += kTwoPi;
                    theta = a + th_in * (b - a);
                }
                fast_sincos(theta, &st, &ct);
            }
            JointsOut o = joints_from_theta_g<true, TIPZ>(A, r, G, ct, st, (const double*)K.prev);
            RSIK_MARK("stores");
#pragma unroll
            for (int k = 0; k < 7; k++) jrow[k] = o.j[k];
            erow[0] = o.elbow.x; erow[1] = o.elbow.y; erow[2] = o.elbow.z;
        } else {
            // (`opaque`: the value is made inside this branch — otherwise the compiler merges the two branches' LDS writes and
            // every wave, reachable or not, first fills ten registers pairs with NaN: 20 v_mov in the all-reachable config 2)
            const double nan = opaque(__builtin_nan(""));
#pragma unroll
            for (int k = 0; k < 7; k++) jrow[k] = nan;
            erow[0] = nan; erow[1] = nan; erow[2] = nan;
        }
#ifdef RSIK_TIMELINE_PROBE
        probe_mid = __builtin_amdgcn_s_memrealtime();  // all arithmetic done, outputs staged in LDS
#endif
        const int64_t wave_base = tile0 + wave * 64;
        if (rows >= (unsigned)(wave * 64 + 64)) {  // the wave's 64 rows all exist (wave-uniform, scalar)
            if (K.joints) flush_rows_full<7>(K.joints, wave_base, lane, lds_wave);
            if (K.elbow) flush_rows_full<3>(K.elbow, wave_base, lane, lds_wave + 64 * 7);
        } else if (rows > (unsigned)(wave * 64)) {
            if (K.joints) flush_rows<7>(K.joints, wave_base, K.n, lane, lds_wave);
            if (K.elbow) flush_rows<3>(K.elbow, wave_base, K.n, lane, lds_wave + 64 * 7);
        }
    }
    if (live) {
        if (K.interval) {
            const f64x2 iv = {r.i0, r.i1};  // one 16-B store per lane
            st_stream(reinterpret_cast<f64x2*>(K.interval + 2 * tile0) + t, iv);
        }
        if (K.reachable) st_stream(K.reachable + tile0 + t, (uint8_t)(r.ok ? 1 : 0));
        if (K.state) st_stream(K.state + tile0 + t, (uint8_t)r.state);
    }
#ifdef RSIK_TIMELINE_PROBE
    // diagnostic build only (scripts/timeline_probe.py): lanes 0-2 of every wave overwrite their interval rows with
    // (start, tables staged), (outputs staged, stores issued), (HW_ID, XCC_ID)
    if (lane < 4 && K.interval && live) {
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t t3 = __builtin_amdgcn_s_memrealtime();
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
        double2 iv;
        if (lane == 0) iv = {(double)probe_t0, (double)probe_t1};
        else if (lane == 1) iv = {(double)probe_mid, (double)t3};
        else if (lane == 2) iv = {(double)hw, (double)xcc};
        else iv = {(double)probe_goal, (double)probe_reach};  // (the stage timers' extra stamps: goal vectors done, is_reachable done)
        reinterpret_cast<double2*>(K.interval)[tile0 + t] = iv;
    }
#endif
#ifdef RSIK_CLOCK_PROBE
    // diagnostic build only (scripts/clock_probe.py): lane 0 of every wave overwrites its interval row with the wave's
    // lifetime in core-clock ticks (s_memtime) and in 100 MHz ticks (s_memrealtime)
    if (lane == 0 && K.interval && live) {
        const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        double2 iv = {(double)(c1 - probe_c0), (double)(r1 - probe_r0)};
        reinterpret_cast<double2*>(K.interval)[tile0 + t] = iv;
    }
#endif
}

// C:212-217: M -> goal pose.  np.allclose(R, I) snaps to the identity.  Otherwise the reference converts R to
// extrinsic xyz Euler angles (U:84-90) and the solver rebuilds the rotation from them (S:420).  For a proper rotation
// away from gimbal lock that round trip reproduces R to rounding, so R is consumed directly (Q6); the round trip is
// really made (euler_xyz_from_matrix + rot_from_euler) exactly where it changes the result:
//   - R is not orthonormal to 1e-12 (SciPy then substitutes the nearest rotation), or
//   - the pitch is within ~1e-5 of +-pi/2 (inside 1e-7 of the lock SciPy sets yaw := 0, which moves the joints by up
//     to ~4e-6 rad: measured on the G8 goldens).
// mode (RSIK_OPT_EULER_ROUNDTRIP): 0 = as above, 1 = always, 2 = never.
// `special` (optional): set when the matrix did not go through as it came — the identity shortcut, the Euler round trip —
// or is not a proper rotation whose third row is the cross product of the other two: the trajectory pipeline's joints
// phase re-reads all twelve entries only for those (cont_joints_kernel).
__device__ __forceinline__ void goal_from_m12(const double (&m)[12], Rot& Rg, V3& pos, int mode, bool* special = nullptr) {
#pragma unroll
    for (int k = 0; k < 9; k++) Rg.m[k] = m[k];
    if (special) {
        // row 2 against row 0 x row 1, entry by entry, to 1e-14: a few roundings of the entries themselves.  (A looser test — 1e-9
        // until round 4 — let the joints phase rebuild the third row of a matrix that is NOT orthonormal to rounding, and where
        // the arm is stretched out the elbow-yaw / wrist-yaw split amplifies such a difference 2e4 times and more: the pipeline
        // and the step kernel could then differ by ~1e-5 rad under RSIK_EULER_NEVER.  Anything else re-reads the three entries.)
        const double c6 = fma(m[1], m[5], -(m[2] * m[4])), c7 = fma(m[2], m[3], -(m[0] * m[5])), c8 = fma(m[0], m[4], -(m[1] * m[3]));
        *special = !(fabs(c6 - m[6]) <= 1e-14 && fabs(c7 - m[7]) <= 1e-14 && fabs(c8 - m[8]) <= 1e-14);
    }
    // np.allclose(R, I) needs all nine entries close; R00 alone rules it out for nearly every goal
    bool eye = RSIK_RARE(np_isclose(Rg.m[0], 1.0));
    if (eye) {
#pragma unroll
        for (int k = 1; k < 9; k++) eye = eye && np_isclose(Rg.m[k], (k % 4 == 0) ? 1.0 : 0.0);
    }
    if (eye) {  // C:212-214 np.allclose(R, I)
        if (special) *special = true;
#pragma unroll
        for (int k = 0; k < 9; k++) Rg.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
    } else {
        bool rt = mode == 1;
        if (mode == 0) rt = (fabs(Rg.m[6]) > 1.0 - 1e-10) || !gram_is_identity(Rg.m);
        if (special && rt) *special = true;
        if (RSIK_RARE(rt)) {
            double eul[3];
            euler_xyz_from_matrix(Rg.m, eul);
            Rg = rot_from_euler(eul[0], eul[1], eul[2]);
        }
    }
    pos = {m[9], m[10], m[11]};
}
__device__ __forceinline__ void load_m12(const double* const* in, int64_t i, Rot& Rg, V3& pos, int mode) {
    double m[12];
#pragma unroll
    for (int k = 0; k < 12; k++) m[k] = in[k][i];
    goal_from_m12(m, Rg, pos, mode);
}

// utils.get_euler_from_homogeneous_matrix for a batch (U:84-90), optionally with ControlIK's identity shortcut
// (C:212-214): m12 SoA -> pose SoA (px, py, pz, roll, pitch, yaw), the input layout of rsik_solve.
struct MatrixToPoseArgs {
    int64_t n;
    const double* in[12];
    double* out[6];
    int identity_shortcut;
};
__global__ __launch_bounds__(kBlock) void matrix_to_pose_kernel(const MatrixToPoseArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= K.n) return;
    double m[9], eul[3];
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = K.in[k][i];
    bool eye = K.identity_shortcut != 0;
#pragma unroll
    for (int k = 0; k < 9; k++) eye = eye && np_isclose(m[k], (k % 4 == 0) ? 1.0 : 0.0);
    if (eye) { eul[0] = 0.0; eul[1] = 0.0; eul[2] = 0.0; }
    else euler_xyz_from_matrix(m, eul);
#pragma unroll
    for (int k = 0; k < 3; k++) { K.out[k][i] = K.in[9 + k][i]; K.out[3 + k][i] = eul[k]; }
}

}  // namespace rsik
