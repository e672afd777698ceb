// rsik_lib.hip — HIP kernels (gfx950) and the C ABI of include/rsik.h.
//
// Kernel shape: one pose per lane, 256-thread workgroups (4 wave64), SoA float64 inputs so that
// every global load is a fully coalesced 512-B wave access; the [n,7] / [n,3] row outputs are
// transposed through LDS so that each wave writes its 3584-B / 1536-B slab with unit-stride stores.
// Per-arm constants travel in the kernarg segment (scalar loads, wave-uniform).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "rsik_device.hpp"

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
template <class T>
__device__ __forceinline__ void st_stream(T* p, T v) {
#if RSIK_NT_STORE
    __builtin_nontemporal_store(v, p);
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
template <int W>
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
        for (int k = 0; k < W; k++) st_stream(dst + k * 64 + lane, v[k]);
    } else {
        const int64_t total = rows * W;
#pragma unroll
        for (int k = 0; k < W; k++) {
            int idx = k * 64 + lane;
            if (idx < total) st_stream(dst + idx, lds_wave[idx]);
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
    uint64_t probe_mid = 0;
#endif
    const AccK<MIXED> A{(KConst)&((const __attribute__((address_space(4))) SolveArgs*)__builtin_amdgcn_kernarg_segment_ptr())->arms[0].v[0],
                        (LdsConst)lds_tab.arm[(MIXED != 0 && K.arm[tile0 + tt] != 0) ? 1 : 0], (UnitAtanTab)&lds_tab.utab[0][0]};
    double* lds_wave = lds[wave];

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
    Reach r = reach_g<false, false>(A, pos, G.woff);
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
    if (lane < 3 && K.interval && live) {
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t t3 = __builtin_amdgcn_s_memrealtime();
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
        double2 iv;
        if (lane == 0) iv = {(double)probe_t0, (double)probe_t1};
        else if (lane == 1) iv = {(double)probe_mid, (double)t3};
        else iv = {(double)hw, (double)xcc};
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
        // row 2 against row 0 x row 1, entry by entry (1e-9: far above rounding, far below anything the solver resolves)
        const double c6 = fma(m[1], m[5], -(m[2] * m[4])), c7 = fma(m[2], m[3], -(m[0] * m[5])), c8 = fma(m[0], m[4], -(m[1] * m[3]));
        *special = !(fabs(c6 - m[6]) <= 1e-9 && fabs(c7 - m[7]) <= 1e-9 && fabs(c8 - m[8]) <= 1e-9);
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
    uint64_t probe_t[6];
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
    Reach r = reach_g<false, false>(A, pos, G.woff);
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
    const int em = safety_checks(A.utab, jv, c4, s4, c5, s5, c6, s6, prev, K.max_angle, K.cos_max, K.sin_max);
    RSIK_MARK("disc_store");
    RSIK_DISC_PROBE(4);
    store_rows<7>(K.joints, wave_base, K.n, lane, &lds_slab[wave][0][0], jv);
    if (live) {
        if (K.reachable) K.reachable[i] = found ? 1 : 0;
        if (K.state) K.state[i] = (uint8_t)st_code;
        if (K.emergency) K.emergency[i] = (uint8_t)em;  // RSIK_EMERGENCY_* cause bits
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
    if (live && lane == 1) K.joints[i * 7] = (double)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
#endif
}

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
        load_m12(K.in, ii, Rg, pos, K.euler_roundtrip);
        if (K.first_timed_out || (K.timed_out && K.timed_out[ii])) { has_prev = false; init = true; }  // C:298-304
        if (!has_prev) {  // C:306-325
            has_prev = true;
            continuous_reinit(A, K, ii, K.pref_arg[slot], prev_theta, prev_sol);
        }
        const Goal G = make_goal(A, Rg);
        Reach r;
        const ThetaTarget T = continuous_target<PLANE>(A, pos, G.woff, K.pref_self[slot], K.pref_self_cs[slot], K.pref_self_sn[slot], r);
        if (RSIK_RARE(!T.ok_limits && !r.ok)) {
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

// ------------------------------------------------------------------------------------------
// rsik_control_continuous_run: the phased trajectory pipeline (include/rsik.h).  Workspace of one block of T steps:
//   ws[t][n] doubles  the step's goal for the theta recurrence (phase 1) -> the step's theta (phase 2)
//   flags[t][n] bytes bit 0 is_reachable succeeded, bit 1 the grid search found a theta; (phase 3) bit 2: get_joints hit an
//                     exact singularity and needs previous_sol (recomputed in phase 4); bit 3: the goal matrix is not a
//                     plain proper rotation (goal_from_m12's `special`): phase 3 reads all of it
// Nothing else travels between the phases: the pipeline is bound by HBM traffic, not by arithmetic, so the joint phase
// re-derives the circle it needs from the goal matrix (the geometric half of is_reachable, ~150 instructions) instead
// of reading 22 doubles per trajectory-step that the prepare phase would have to write (652 -> 400 B per step).
// ------------------------------------------------------------------------------------------
// steps whose operands the sequential phases fetch at once, one batch ahead of the one being computed (blocks are multiples
// of both; measured on 4096 x 1000 steps: theta batches of 8 / 16 / 32 steps 0.544 / 0.525 / 0.552 ms per pass)
#ifndef RSIK_THETA_BATCH
#define RSIK_THETA_BATCH 16
#endif
#ifndef RSIK_CHAIN_BATCH
#define RSIK_CHAIN_BATCH 16  // chunks of the joints phase whose first / last rows the chain phase fetches at once
#endif
constexpr int kThetaBatch = RSIK_THETA_BATCH, kChainBatch = RSIK_CHAIN_BATCH;
// consecutive steps of a trajectory that one thread of the joints phase walks (and that the chain phase accepts or redoes as
// one unit)
#ifndef RSIK_JOINT_CHUNK
#define RSIK_JOINT_CHUNK 8
#endif
constexpr int kJointChunk = RSIK_JOINT_CHUNK;

// threads per workgroup of the theta phase: single waves — a workgroup of four has to find four wave slots on ONE compute
// unit while the throughput phases of the neighbouring blocks keep the chip full (4096 x 1000 steps: 0.486 -> 0.448 ms
// per pass).  Measured and not kept: a wave that claims its SIMD's whole register file (512 registers, nothing else
// resident beside it) runs its block in 35-57 us instead of 60-80 us, but the SIMDs it takes from the throughput
// phases cost as much (0.463 ms per pass).
#ifndef RSIK_THETA_BLOCK
#define RSIK_THETA_BLOCK 64
#endif
constexpr int kThetaBlock = RSIK_THETA_BLOCK;
#ifndef RSIK_CHAIN_BLOCK
#define RSIK_CHAIN_BLOCK 256
#endif
constexpr int kChainBlock = RSIK_CHAIN_BLOCK;  // the chain phase: no such gain from single waves (0.447 / 0.452 ms with 256 / 64)
constexpr int kSeqBatch = kThetaBatch > kJointChunk ? kThetaBatch : kJointChunk;
static_assert(kSeqBatch % kThetaBatch == 0 && kSeqBatch % kJointChunk == 0, "block sizes are multiples of the theta batch and of the joint chunk");
struct ContRunArgs {
    int64_t n;
    int64_t t0;                   // first step of this block
    int64_t T;                    // steps in this block
    const double* m12_steps;      // [n_steps][12][n]
    const uint8_t* arm;
    int euler_roundtrip;
    double pref_arg[2], pref_self[2];
    double pref_self_cs[2], pref_self_sn[2];
    double lim[2][2];
    double d_theta_max;
    double max_angle, cos_max, sin_max;
    double* ws;                   // [T][n]: the step's theta goal (phase 1), overwritten by the step's theta (phase 2)
    double* gw;                   // [T][n]: the goal after limit_theta_to_interval's wrap (phase 1 -> phase 2)
    uint8_t* flags;               // [T][n]
    uint8_t* chunk_event;         // [ceil(T / kJointChunk)][n]: phase 3 -> phase 4, see cont_joints_kernel
    int8_t* chunk_turns;          // [ceil(T / kJointChunk)][n][8]: whole turns phase 4 found a chunk's joints away from the step
                                  // before it, applied by phase 5
    double snap_tdag;             // phase 2, single-arm launches: see continuous_next_theta_lean (the kind is a template argument)
    double* theta_carry;          // [n]: previous_theta between the blocks of one run (phase 2's own state)
    int first_block, last_block;
    double* st;                   // cont_state
    double* joints;               // [n_steps][n][7]
    uint8_t* reachable;           // [n_steps][n] or NULL
    uint8_t* state;               // [n_steps][n] or NULL
    ArmC arms[2];
};
#define RSIK_WS(K, t, i) (K).ws[(int64_t)(t) * (K).n + (i)]

// phase 1: one thread per (trajectory, step of the block)
template <bool MIXED, bool PLANE>
__global__ __launch_bounds__(kBlock) void cont_prepare_kernel(const ContRunArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t t = blockIdx.y;
    const bool live = i < K.n;
    const int64_t ii = live ? i : (K.n - 1);
    double m[12];  // loads first: their latency overlaps the table staging
    const double* src = K.m12_steps + (K.t0 + t) * 12 * K.n + ii;
#pragma unroll
    for (int k = 0; k < 12; k++) m[k] = src[k * K.n];
    const bool lane_isl = MIXED ? (K.arm[ii] != 0) : false;
    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(ContRunArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, lane_isl, lds_tab);
    const int slot = MIXED ? (A.isl ? 1 : 0) : 0;
    Rot Rg;
    V3 pos;
    bool special;
    goal_from_m12(m, Rg, pos, K.euler_roundtrip, &special);
    const Goal G = make_goal(A, Rg);
    Reach r;
    const ThetaTarget T = continuous_target<PLANE, false>(A, pos, G.woff, K.pref_self[slot], K.pref_self_cs[slot], K.pref_self_sn[slot], r);
    if (!live) return;
    // the step's goal for the theta phase: the search's theta, NaN = nothing found, stay (U:252-264 with goal =
    // previous_theta), or the preferred theta of an unreachable pose (U:115-127)
    const double goal = T.ok_limits ? (T.found ? T.theta : __builtin_nan("")) : K.pref_arg[slot];
    RSIK_WS(K, t, i) = goal;
    // what limit_theta_to_interval makes of theta = goal before it looks at the interval (U:93-97): this phase has the
    // issue slots for it, the theta phase (a lone wave per SIMD) has not
    K.gw[t * K.n + i] = wrap_theta_to_pi(goal);
    K.flags[t * K.n + i] = (uint8_t)((T.ok_limits ? 1 : 0) | (T.found ? 2 : 0) | (special ? 8 : 0));
    if (K.state) K.state[(K.t0 + t) * K.n + i] = (uint8_t)T.code;
    if (K.reachable) K.reachable[(K.t0 + t) * K.n + i] = (T.ok_limits && T.found) ? 1 : 0;
}

// Row + lane addressing for the sequential phases: a step's row starts `row` bytes into the block's array (the same for
// the whole wave: a scalar register), the lane's element `lane` bytes into the row — buffer instructions take exactly
// these two, so an access costs one scalar addition and no 64-bit address arithmetic per lane (a batch of 32 steps would
// otherwise hold 64 vector registers of addresses, or recompute them in the lone wave's instruction stream).
// (raw buffer, 2 GB window: the host keeps a block's arrays below that)
typedef unsigned RowWords2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_buffer(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ double ld_row_f64(__amdgpu_buffer_rsrc_t buf, unsigned lane, unsigned row) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(buf, lane, row, 0));
}
__device__ __forceinline__ void st_row_f64(__amdgpu_buffer_rsrc_t buf, unsigned lane, unsigned row, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(RowWords2, v), buf, lane, row, 0);
}
__device__ __forceinline__ int ld_row_u8(__amdgpu_buffer_rsrc_t buf, unsigned lane, unsigned row) {
    return (int)__builtin_amdgcn_raw_buffer_load_b8(buf, lane, row, 0);
}

// phase 2: one thread per trajectory walks the block's steps: the recurrence on previous_theta.
// KIND: kSnapInner / kSnapWrap = the step specialised for the launch's control interval (continuous_next_theta_lean;
// single-arm launches), kSnapGeneric = the reference's own sequence of operations for any interval.
template <bool MIXED, int KIND>
__global__ __launch_bounds__(kThetaBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void cont_theta_kernel(const ContRunArgs K) {
    static_assert(!MIXED || KIND == kSnapGeneric, "a mixed launch has an interval per lane");
    // a serial phase: its few waves share their SIMDs with the chip-filling phases of the neighbouring blocks (other
    // streams) and must win the issue arbitration, or every instruction waits behind throughput work
    __builtin_amdgcn_s_setprio(3);
    const int64_t i = (int64_t)blockIdx.x * kThetaBlock + threadIdx.x;
    if (i >= K.n) return;
    const int slot = MIXED ? (K.arm[i] != 0 ? 1 : 0) : 0;
    // previous_theta travels from block to block in theta_carry: this phase runs ahead of phase 4 (other streams), which
    // alone decides what ends up in the state's row 0 — the theta of the last step, or of the step that latched the
    // emergency stop (C:205-210; what this phase computes for a latched trajectory is never looked at).
    double prev_theta = K.first_block ? K.st[0 * K.n + i] : K.theta_carry[i];
    const double l0 = K.lim[slot][0], l1 = K.lim[slot][1];
    // A lone wave per SIMD: every instruction of a step is paid in full (~4.5 cycles each, rsik_device.hpp `opaque`), and
    // the memory round trip of a step's operands would double a step, so they are fetched kThetaBatch steps at a time,
    // one batch ahead of the one being computed, into two register sets that take turns (no copies); what is left of the
    // block after the last full batch goes step by step.
    const int64_t n = K.n;
    // this trajectory's goal / theta of the step the wave is at: (wbuf, off, row), its wrapped goal (gbuf, off, row); a step
    // further is `stride` bytes further
    const __amdgpu_buffer_rsrc_t wbuf = row_buffer(K.ws), gbuf = row_buffer(K.gw);
    const unsigned off = (unsigned)(i * sizeof(double)), stride = (unsigned)(n * sizeof(double));
    unsigned row = 0;
    // launch constants that a select or a sign transfer needs as a vector operand: pinned in vector registers once
    const double dmax_v = opaque(K.d_theta_max), l0v = opaque(l0), l1v = opaque(l1), tdag_v = opaque(K.snap_tdag);
    // `g` is the step's goal as the prepare phase encoded it: the search's theta, the preferred theta for an unreachable
    // pose, NaN = "stay".  Straight-line arithmetic only.
    auto generic = [&](double g) {
        return continuous_next_theta_goal((g != g) ? prev_theta : g, prev_theta, K.d_theta_max, l0, l1, dmax_v, l1v);
    };
    auto one = [&](double g, double gw, unsigned dst_row) {
        if constexpr (KIND == kSnapGeneric) prev_theta = generic(g);
        else prev_theta = continuous_next_theta_lean<KIND>(g, gw, prev_theta, dmax_v, l0v, l1v, tdag_v);
        st_row_f64(wbuf, off, dst_row, prev_theta);
    };
    int64_t left = K.T;
    if (KIND != kSnapGeneric && K.first_block && left > 0) {
        // the state a run starts from is the caller's: only from the first result on is previous_theta known to lie in
        // [-pi, pi], which the specialised step relies on
        prev_theta = generic(ld_row_f64(wbuf, off, row));
        st_row_f64(wbuf, off, row, prev_theta);
        row += stride; left -= 1;
    }
    struct Operands { double g[kThetaBatch], gw[kThetaBatch]; };
    // (`valid` < kThetaBatch: the block's last, partial batch — the steps past its end repeat the last one and are skipped)
    auto fetch = [&](Operands& o, int ahead, int valid) {
#pragma unroll
        for (int u = 0; u < kThetaBatch; u++) {
            const unsigned at = row + (unsigned)(ahead + (u < valid ? u : valid - 1)) * stride;
            o.g[u] = ld_row_f64(wbuf, off, at);
            if constexpr (KIND != kSnapGeneric) o.gw[u] = ld_row_f64(gbuf, off, at);
        }
    };
    auto compute = [&](const Operands& o, auto partial, int valid) {  // the batch at `row`; leaves `row` at the next one
        constexpr bool kPartial = decltype(partial)::value;
        const unsigned r0 = row;
        row += (unsigned)(kPartial ? valid : kThetaBatch) * stride;
        // one wait for the whole set (it was fetched a batch ago) instead of one per operand: a wait is an issue slot too
        asm volatile("" : : "v"(o.g[kThetaBatch - 1]), "v"(o.gw[KIND != kSnapGeneric ? kThetaBatch - 1 : 0]));
#pragma unroll
        for (int u = 0; u < kThetaBatch; u++) {
            if (!kPartial || u < valid) one(o.g[u], o.gw[u], r0 + (unsigned)u * stride);  // (launch-uniform: a scalar branch)
        }
    };
    int64_t batches = left / kThetaBatch;
    left -= batches * kThetaBatch;
    Operands a, b;
    if (batches > 0) fetch(a, 0, kThetaBatch);
#pragma unroll 1
    while (batches >= 2) {
        fetch(b, kThetaBatch, kThetaBatch);
        compute(a, std::false_type{}, kThetaBatch);
        if (batches > 2) fetch(a, kThetaBatch, kThetaBatch);
        compute(b, std::false_type{}, kThetaBatch);
        batches -= 2;
    }
    if (batches == 1) {
        if (left > 0) fetch(b, kThetaBatch, (int)left);
        compute(a, std::false_type{}, kThetaBatch);
        if (left > 0) compute(b, std::true_type{}, (int)left);
    } else if (left > 0) {
        fetch(a, 0, (int)left);
        compute(a, std::true_type{}, (int)left);
    }
    K.theta_carry[i] = prev_theta;
}

// What get_joints reads of a step (S:697-863), re-derived from the step's goal matrix: the goal vectors and the circle
// is_reachable (flag bit 0 set) or is_reachable_no_limits (clear; C:371) left on the solver — the same device code the
// step kernel runs, so the joints are the same to the last bit.  `m`: the step's twelve matrix entries.
// `plain`: the prepare phase found the matrix a plain proper rotation (no identity shortcut, no Euler round trip): taken as it is.
template <class Acc>
__device__ __forceinline__ void step_geometry(const Acc& A, const double (&m)[12], int euler_roundtrip, bool no_limits, Reach& r, Goal& G,
                                              bool plain = false) {
    Rot Rg;
    V3 pos;
    if (plain) {
#pragma unroll
        for (int k = 0; k < 9; k++) Rg.m[k] = m[k];
        pos = {m[9], m[10], m[11]};
    } else {
        goal_from_m12(m, Rg, pos, euler_roundtrip);
    }
    G = make_goal(A, Rg);
    r = reach_impl<false, true>(A, pos, G.woff, no_limits);
}
__device__ __forceinline__ void load_step_m12(const ContRunArgs& K, int64_t t, int64_t i, double (&m)[12]) {
    const double* src = K.m12_steps + (K.t0 + t) * 12 * K.n + i;
#pragma unroll
    for (int k = 0; k < 12; k++) m[k] = src[k * K.n];
}

// get_joints at theta (S:697-863) + the Orbita3D cone clamp (U:508-532): everything of a step's joints that does not
// need previous_sol.  `sing`: an exact singularity fell back to prev (S:751-753, 782-784).
template <class Acc>
__device__ __forceinline__ void step_joints(const Acc& A, const ContRunArgs& K, Reach& r, const Goal& G, double theta,
                                            const double* prev, double (&jv)[7], bool& sing) {
    double sn, cs;
    fast_sincos(theta, &sn, &cs);
    JointsOut o = joints_from_theta_g<true>(A, r, G, cs, sn, prev);
#pragma unroll
    for (int k = 0; k < 7; k++) jv[k] = o.j[k];
    sing = o.sing;
    limit_wrist_cone(A.utab, jv, o.c4, o.s4, o.c5, o.s5, o.c6, o.s6, K.cos_max, K.sin_max);
}

// phase 3: one thread per (trajectory, step of the block); a wave holds a CHUNK of kJointChunk = 8 consecutive steps of 8
// neighbouring trajectories (lane = 8 * step + trajectory), so that besides get_joints + the cone clamp it can do the quiet
// part of the previous_sol recurrence itself.  allow_multiturn (U:493-505) is previous + angle_diff(joint, previous): the
// representative of the raw joint (mod 2 pi) nearest the previous step's.  Inside a chunk that is a prefix sum of whole
// turns: lane (s, i) takes the raw joints of step s - 1 from the lane eight below it, turn(s) = -rint((raw(s) - raw(s-1)) /
// 2 pi) (zero unless a raw angle crossed its branch cut), three shuffle rounds add them up, joint = raw + 2 pi turns; the
// chunk's first step keeps its raw value.  The turns a chunk AS A WHOLE sits away from the step before it are the
// sequential phase's business (phase 4 finds them from the chunks' first and last rows, phase 5 adds them in): they are
// not zero often enough to guess — shoulder pitch and elbow yaw swing by more than pi within a few hundred steps when the
// arm passes its shoulder singularity (8 % of config 5's steps have them outside [-pi, pi]).  What the reference decides
// step by step — the continuity thresholds (U:571-589, C:398), the +-6 pi limit (U:535-568), an exact singularity that
// needs previous_sol (S:751-753, 782-784) — is only DETECTED here, with a margin of 1e-9: the chunk's event byte tells
// phase 4 to walk that chunk with the reference's own sequence of operations.  So the joints make ONE trip to HBM but
// for the shifted elements (phase 4 used to read and rewrite all of them, 112 of the 412 bytes a control step moved),
// and a quiet step's value is its raw joint plus whole turns: within 2 ulp of the reference's previous + angle_diff(raw,
// previous), no accumulation.
template <bool MIXED>
__global__ __launch_bounds__(kBlock) void cont_joints_kernel(const ContRunArgs K) {
    static_assert(kJointChunk == 8, "lane = 8 * step + trajectory");
    __shared__ double lds_out[kBlock / 64][64 * 7];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tl = lane & 7, sl = lane >> 3;
    const int64_t n = K.n;
    const int64_t grp = (int64_t)blockIdx.x * (kBlock / 64) + wave;  // this wave's group of 8 trajectories
    const int64_t i = grp * 8 + tl;
    const int64_t c = blockIdx.y;
    const int64_t t = c * kJointChunk + sl;
    const bool live = i < n && t < K.T;
    const int64_t ii = i < n ? i : (n - 1);
    const int64_t tt = t < K.T ? t : (K.T - 1);
    // loads first: their latency overlaps the table staging.  Of the goal matrix the first two rows of the rotation and
    // the translation: for a proper rotation, which the prepare phase has checked (flag bit 3 clear), the third row is their
    // cross product — to 1e-16, the rounding of the entries themselves — and 24 of the 161 bytes this phase moves per step
    // need not be read.
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
    Reach r;
    Goal G;
    step_geometry(A, m, K.euler_roundtrip, (flag & 1) == 0, r, G, !special);
    const double zeros[7] = {0, 0, 0, 0, 0, 0, 0};
    double jv[7];
    bool sing;
    step_joints(A, K, r, G, theta, zeros, jv, sing);
    // Steps relative to the step before (lane - 8; none for the chunk's first step, which phase 4 judges).  Whole turns
    // only for the four joints whose raw angle has a branch cut to cross — shoulder pitch, elbow yaw, wrist roll, wrist yaw
    // (atan2 values, S:751-786, 815-848 / U:508-519); shoulder roll is atan2(q_y, q_x >= 0), elbow pitch is clamped to
    // +-elbow_limit < pi (S:853-861) and wrist pitch is an asin (U:517), whatever the arm's geometry: for those a turn
    // could only be part of a step beyond the continuity thresholds, which is an event either way.
    const int below = (sl == 0 ? lane : lane - 8) << 2;
    auto from_below = [&](double v) {
        const int lo_ = __builtin_amdgcn_ds_bpermute(below, (int)__double2loint(v));
        const int hi_ = __builtin_amdgcn_ds_bpermute(below, (int)__double2hiint(v));
        return __hiloint2double(hi_, lo_);
    };
    double worst_a = 0.0, worst_b = 0.0;  // largest |step| among joints 0-3 (threshold 0.5) and 4-6 (1.0), C:398
    double packed = 0.0;                   // (8 + turn) of joints 6, 4, 2, 0 as base-256 digits: eight of them add up without a carry
    double turn[7];
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
    // a singular step (NaN joints, here or in the lane below) is an event too: fmax drops NaNs, so it is told by the flags
    const unsigned long long sing_mask = __ballot(sing);
    const bool sing_below = sl > 0 && ((sing_mask >> (lane - 8)) & 1ull) != 0;
    const bool ev = sing || sing_below || !(worst_a <= 0.5 - 1e-9) || !(worst_b <= 1.0 - 1e-9) || !(fabs(packed) < 4.0e9);
    unsigned word = (unsigned)packed;  // (garbage for a NaN: the chunk is an event then)
#pragma unroll
    for (int step = 1; step < 8; step *= 2) {  // inclusive prefix sum over the chunk's steps (lane stride 8)
        const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute((lane - 8 * step) << 2, (int)word);
        if (sl >= step) word += w;
    }
#pragma unroll
    for (int k = 0; k < 7; k++) turn[k] = 0.0;
    {
        const int bias = 8 * (sl + 1);
        turn[0] = (double)((int)(word & 0xffu) - bias);
        turn[2] = (double)((int)((word >> 8) & 0xffu) - bias);
        turn[4] = (double)((int)((word >> 16) & 0xffu) - bias);
        turn[6] = (double)((int)(word >> 24) - bias);
    }
    double out[7];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const double o = (k == 0 || k == 2 || k == 4 || k == 6) ? fma(turn[k], kTwoPi, jv[k]) : jv[k];
        out[k] = sing ? __builtin_nan("") : o;  // (singular: needs previous_sol, phase 4 recomputes the step — flag bit 2)
    }
    if (live && sing) K.flags[t * n + i] = (uint8_t)(flag | 4);
    // one event byte per (chunk, trajectory): OR over the chunk's steps
    const unsigned long long evm = __ballot(ev && live);
    if (live && sl == 0) K.chunk_event[c * n + i] = ((evm >> tl) & 0x0101010101010101ull) != 0 ? 1 : 0;
    // rows out: the wave's 64 rows are 8 runs (one per step) of 8 x 7 consecutive doubles; 32-bit offsets from the chunk's
    // first row (a block's joints stay below 2 GB, see rsik_control_continuous_run)
    double* lw = lds_out[wave];
#pragma unroll
    for (int k = 0; k < 7; k++) lw[lane * 7 + k] = out[k];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int traj_left = (int)((n - grp * 8) < 8 ? (n - grp * 8) : 8);  // trajectories of this group that exist (<= 0 past the end)
    const int steps_left = (int)((K.T - c * kJointChunk) < kJointChunk ? (K.T - c * kJointChunk) : kJointChunk);
    const __amdgpu_buffer_rsrc_t obuf = row_buffer(K.joints + ((K.t0 + c * kJointChunk) * n + grp * 8) * 7);
    const unsigned row_bytes = (unsigned)(n * 7 * sizeof(double));
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const int idx = k * 64 + lane;
        const int s_ = idx / 56, off = idx - s_ * 56;
        if (s_ < steps_left && off < traj_left * 7) st_row_f64(obuf, (unsigned)s_ * row_bytes + (unsigned)off * 8u, 0, lw[idx]);
    }
}

// phase 4: eight lanes per trajectory, lane j < 7 owns joint j; sequential over the block's steps: the recurrence on
// previous_sol (allow_multiturn U:493-505, multiturn_safety_check U:535-568, continuity_check U:571-589, the emergency
// latch C:205-210, C:398-405).
template <bool MIXED>
__global__ __launch_bounds__(kChainBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void cont_chain_kernel(const ContRunArgs K) {
    // a serial phase beside throughput phases (see cont_theta_kernel), one step below the theta phase, which is the
    // critical path where the two share a SIMD (0.544 -> 0.536 ms per 4096 x 1000 pass)
    __builtin_amdgcn_s_setprio(2);
    const int64_t gid = (int64_t)blockIdx.x * kChainBlock + threadIdx.x;
    const int64_t i = gid >> 3;
    const int j = (int)(gid & 7);
    const int lane = threadIdx.x & 63;
    const int gshift = lane & ~7;
    const bool live = i < K.n;
    const int64_t ii = live ? i : (K.n - 1);
    const int jj = j < 7 ? j : 6;
    const bool owner = live && j < 7;
    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(ContRunArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC)), kChainBlock>(lds_tab, K.arms);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[ii] != 0) : false, lds_tab);
    const int64_t n = K.n;
    double prev = K.st[(1 + jj) * n + ii];
    bool init = K.st[8 * n + ii] != 0.0;
    bool emergency = K.st[9 * n + ii] != 0.0;
    const double thr = jj < 4 ? 0.5 : 1.0;                                       // continuity thresholds, C:398
    const double lim = (jj == 0 || jj == 2 || jj == 6) ? 6 * kPi : __builtin_inf();  // multiturn limit of this lane's joint (U:535-568)
    const int hit_bit = jj == 0 ? RSIK_EMERGENCY_SHOULDER_PITCH : (jj == 2 ? RSIK_EMERGENCY_ELBOW_YAW : RSIK_EMERGENCY_WRIST_YAW);
    // OR over the 8 lanes of a trajectory, left in every one of them: two quad permutes and a half-row mirror (DPP)
    auto group_or = [](int v) -> int {
        v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
        v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
        return v;
    };
    // Operands of kChainBatch steps are fetched together, one batch ahead (see cont_theta_kernel).  A step is straight-line
    // code: a latched trajectory (rare) goes through the same arithmetic and only its selects differ.
    // this lane's joint of the step the wave is at: jrow[joff]; its flag byte: frow[foff] (ld_row); a step further is
    // step_stride doubles / n bytes further
    const __amdgpu_buffer_rsrc_t jbuf = row_buffer(K.joints + K.t0 * n * 7), fbuf = row_buffer(K.flags);
    const unsigned joff = (unsigned)((ii * 7 + jj) * sizeof(double)), foff = (unsigned)ii;
    const unsigned jstride = (unsigned)(n * 7 * sizeof(double)), fstride = (unsigned)n;
    int64_t t_abs = K.t0;
    auto one = [&](double cur, int f, int64_t t, unsigned jrow) {  // step t of the block; this lane's joint of it at (jbuf, joff, jrow)
        if (RSIK_RARE((f & 4) != 0 && !emergency)) {  // the same byte in all 8 lanes of the trajectory
            // exact singularity in get_joints: the step is recomputed with the real previous_sol (every lane of the
            // group computes all seven joints and keeps its own)
            double pv[7];
#pragma unroll
            for (int k = 0; k < 7; k++) pv[k] = __shfl(prev, gshift + k);
            Reach r;
            Goal G;
            double m[12];
            load_step_m12(K, t, ii, m);
            step_geometry(A, m, K.euler_roundtrip, (f & 1) == 0, r, G);
            double jv[7];
            bool sing;
            step_joints(A, K, r, G, RSIK_WS(K, t, ii), pv, jv, sing);
            cur = jv[0];
#pragma unroll
            for (int k = 1; k < 7; k++) cur = (jj == k) ? jv[k] : cur;
        }
        const double turned = allow_multiturn_one_straight(cur, prev);        // U:493-505
        const double clamped = fmin(fmax(turned, -lim), lim);                 // U:535-568 (lim = inf for joints 1, 3, 4, 5)
        int code = (clamped != turned && j < 7) ? hit_bit : 0;
        // U:571-589: |angle_diff(joint, previous)| against the joint's threshold, on the limited value like the reference
        code |= (j < 7 && fabs(angle_diff_straight(clamped, prev)) > thr) ? 16 : 0;
        code = group_or(code);
        const bool disc = !init && (code & 16) != 0;
        const int cause = (code & 7) | (disc ? RSIK_EMERGENCY_CONTINUITY : 0);
        const double accepted = disc ? prev : clamped;
        const bool trips = cause != 0 && !emergency;
        const double result = emergency ? prev : accepted;                    // latched (C:205-210): previous_sol
        if (owner) st_row_f64(jbuf, joff, jrow, result);
        if (RSIK_RARE(emergency || trips) && live) {
            if (emergency) {
                if (j == 7) {
                    if (K.state) K.state[t_abs * n + i] = (uint8_t)RSIK_STATE_EMERGENCY;
                    if (K.reachable) K.reachable[t_abs * n + i] = 0;
                }
            } else if (j == 7) {
                K.st[11 * n + i] = (double)cause;
                K.st[0 * n + i] = RSIK_WS(K, t, i);  // previous_theta of the step that tripped (phase 2 ran ahead)
            } else if (disc) {
                K.st[(12 + j) * n + i] = clamped;       // the joints that failed the check
            }
        }
        prev = (emergency || trips) ? prev : accepted;
        init = emergency ? init : false;
        emergency = emergency || trips;
        t_abs += 1;
    };
    // Phase 3 has already done the quiet part of the recurrence (see cont_joints_kernel): this phase walks the block CHUNK
    // by chunk.  A chunk stands as phase 3 wrote it when its event byte is clear, the trajectory is neither latched nor at
    // its first step after a (re)initialisation, and its first step lies within the continuity threshold (less 1e-9) of
    // previous_sol — which also says that phase 3 picked the right turn; previous_sol then becomes the chunk's last row.
    // Otherwise the chunk's steps go through `one`, the reference's own sequence of operations, in place (it re-bases
    // whatever representative phase 3 wrote).  Per chunk this reads two rows of the joints and a byte instead of
    // reading and rewriting every row; the first / last rows and event bytes of kChainBatch chunks are fetched at once.
    const double thr_short = thr - 1e-9;
    const __amdgpu_buffer_rsrc_t ebuf = row_buffer(K.chunk_event);
    // step by step with `one` (the only copy of it), the operands of the next three steps in flight meanwhile
    auto stepwise = [&](int64_t t_blk, int64_t count) {  // the steps [t_blk, t_blk + count) of the block
        unsigned jrow = (unsigned)t_blk * jstride, frow = (unsigned)t_blk * fstride;
        t_abs = K.t0 + t_blk;
        auto at = [&](int64_t k) { return k < count ? k : count - 1; };
        auto raw_at = [&](int64_t k) { return ld_row_f64(jbuf, joff, jrow + (unsigned)k * jstride); };
        auto flag_at = [&](int64_t k) { return ld_row_u8(fbuf, foff, frow + (unsigned)k * fstride); };
        double r0 = raw_at(0), r1 = raw_at(at(1)), r2 = raw_at(at(2));
        int f0 = flag_at(0), f1 = flag_at(at(1)), f2 = flag_at(at(2));
#pragma unroll 1
        for (int64_t k = 0; k < count; ++k) {
            const int64_t ahead = at(k + 3) - k;
            const double rn = raw_at(ahead);
            const int fn = flag_at(ahead);
            one(r0, f0, t_blk, jrow);
            r0 = r1; r1 = r2; r2 = rn;
            f0 = f1; f1 = f2; f2 = fn;
            jrow += jstride;
            frow += fstride;
            t_blk += 1;
        }
    };
    const int64_t n_chunks = (K.T + kJointChunk - 1) / kJointChunk;
    struct Operands { double first[kChainBatch], last[kChainBatch]; int ev[kChainBatch]; };
    auto chunk_len = [&](int64_t c) { return (K.T - c * kJointChunk) < kJointChunk ? (K.T - c * kJointChunk) : (int64_t)kJointChunk; };
    auto fetch = [&](Operands& o, int64_t c0) {
#pragma unroll
        for (int u = 0; u < kChainBatch; u++) {
            const int64_t c = (c0 + u) < n_chunks ? (c0 + u) : (n_chunks - 1);  // (past the end: the last chunk again, skipped)
            const unsigned r_first = (unsigned)(c * kJointChunk) * jstride;
            o.first[u] = ld_row_f64(jbuf, joff, r_first);
            o.last[u] = ld_row_f64(jbuf, joff, r_first + (unsigned)(chunk_len(c) - 1) * jstride);
            o.ev[u] = ld_row_u8(ebuf, foff, (unsigned)c * fstride);
        }
    };
    // Walks the fetched chunks until one does not stand: returns its index in the batch (kChainBatch: all stood).
    // Phase 3 left each chunk on the turn of its first step's raw joints; `turns` (this lane's joint, almost always 0) is
    // how many whole turns that is away from previous_sol.  They go to chunk_turns for phase 5, which adds them to the
    // chunk's rows — nothing sequential, and only the elements that need it.  The limits (U:535-568): phase 3 cannot test
    // them without the turn, so they are tested here on the chunk's first step with the slack its other steps can use
    // up — they lie within (chunk - 1) continuity thresholds of it.
    const bool limited = jj == 0 || jj == 2 || jj == 6;
    const double clear_of_limit = 6 * kPi - (kJointChunk - 1) * 1.0 - 1e-6;
    int8_t* const turns_out = K.chunk_turns + ii * 8 + j;  // (+ chunk * n * 8)
    auto walk = [&](const Operands& o, int64_t c0) -> int {
        int stop = kChainBatch;
#pragma unroll
        for (int u = 0; u < kChainBatch; u++) {
            const double turns = -rint((o.first[u] - prev) * 0.15915494309189535);
            const double sh = turns * kTwoPi;
            const double f2 = o.first[u] + sh;
            const bool quiet = !emergency && !init && o.ev[u] == 0 && (fabs(f2 - prev) <= thr_short) && (fabs(turns) <= 100.0) &&
                               (!limited || fabs(f2) <= clear_of_limit);
            const bool inside = c0 + u < n_chunks;
            const bool stands = !__any(!quiet) && inside;  // (wave-uniform)
            const bool taken = stop == kChainBatch && stands;
            if (stop == kChainBatch && !stands) stop = u;
            if (taken) prev = o.last[u] + sh;
            // (a chunk that goes through `one` instead is rewritten there: no turns to add)
            if (live && inside) turns_out[(c0 + u) * n * 8] = (int8_t)(taken ? (int)turns : 0);
        }
        return stop;
    };
    {
        Operands oa, ob;
        int64_t c0 = 0;
        fetch(oa, 0);
#pragma unroll 1
        while (c0 < n_chunks) {
            const bool more = c0 + kChainBatch < n_chunks;
            if (more) fetch(ob, c0 + kChainBatch);
            const int stop = walk(oa, c0);
            if (RSIK_RARE(c0 + stop < n_chunks && stop < kChainBatch)) {
                // an eventful chunk: the reference's own sequence of operations for its steps, then the walk resumes behind it
                stepwise((c0 + stop) * kJointChunk, chunk_len(c0 + stop));
                c0 += stop + 1;
                if (c0 < n_chunks) fetch(oa, c0);
            } else {
                c0 += kChainBatch;
                oa = ob;
            }
        }
    }
    if (owner) K.st[(1 + j) * n + i] = prev;
    if (live && j == 7) {
        K.st[8 * n + i] = init ? 1.0 : 0.0;
        K.st[9 * n + i] = emergency ? 1.0 : 0.0;
        if (K.last_block && !emergency) K.st[0 * n + i] = RSIK_WS(K, K.T - 1, i);  // previous_theta after the last step
    }
}

// phase 5: adds the whole turns phase 4 found (chunk_turns) to the chunk's rows: one thread per (chunk, trajectory), most of
// which find eight zero bytes and leave; the others read the elements of every joint that turns (all at once: one memory
// round trip), add and write them back.
__global__ __launch_bounds__(kBlock) void cont_turns_kernel(const ContRunArgs K) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (i >= K.n) return;
    const unsigned long long packed = *reinterpret_cast<const unsigned long long*>(K.chunk_turns + (c * K.n + i) * 8);
    if (packed == 0) return;
    const int64_t t_begin = c * kJointChunk;
    const int len = (int)((K.T - t_begin) < kJointChunk ? (K.T - t_begin) : kJointChunk);
    double* const p = K.joints + ((K.t0 + t_begin) * K.n + i) * 7;
    const int64_t row = K.n * 7;
    double v[7][kJointChunk];
#pragma unroll
    for (int k = 0; k < 7; k++) {
        if (((packed >> (8 * k)) & 0xff) != 0) {
#pragma unroll
            for (int q = 0; q < kJointChunk; q++) v[k][q] = p[(int64_t)(q < len ? q : len - 1) * row + k];
        }
    }
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const int turns = (int)(int8_t)((packed >> (8 * k)) & 0xff);
        if (turns != 0) {
            const double sh = (double)turns * kTwoPi;
#pragma unroll
            for (int q = 0; q < kJointChunk; q++)
                if (q < len) p[(int64_t)q * row + k] = v[k][q] + sh;
        }
    }
}

// C:296-325 for the trajectories of a batch that (re)initialise: previous_sol, previous_theta, init.
// PAIR: two lanes per trajectory share the start-up search (best_theta_to_current_joints<PAIR>): half its latency, which
// is on the critical path of a run.
template <bool MIXED, bool PAIR>
__global__ __launch_bounds__(kBlock) void cont_init_kernel(const ContinuousArgs K) {
    __builtin_amdgcn_s_setprio(3);  // a few lone waves on the critical path, beside the chip-filling prepare phase
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
    Rot Rg = rot_from_euler(e0, e1, e2);
    Reach r = K.no_limits ? reach<true>(A, pos, Rg) : reach<false>(A, pos, Rg);
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

// =====================================================================================
// C ABI
// =====================================================================================
struct rsik_ctx {
    int device;
    hipStream_t stream;
    bool have_arm[2];
    rsik::ArmC arms[2];
    int options[RSIK_OPT_COUNT];
    void* ws;          // workspace of rsik_control_continuous_run's phased pipeline (device), grown on demand
    size_t ws_bytes;
    std::vector<void*> retired_ws;   // outgrown workspaces: kept until rsik_destroy (a captured hipGraph may still point into them)
    hipStream_t side[3];             // the pipeline's own streams (prepare / joints / chain), created on first use
    std::vector<hipEvent_t> events;  // reusable, timing disabled
    bool have_side;
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(rsik_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_create_err = msg;
    return code;
}
static int hip_fail(rsik_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, RSIK_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define RSIK_HIP(ctx, call)                                   \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

extern "C" {

int rsik_abi_version(void) { return RSIK_ABI_VERSION; }
#ifndef RSIK_SOURCE_HASH
#define RSIK_SOURCE_HASH "00000000000000000000000000000000"
#endif
const char* rsik_build_id(void) { return "RSIK_SRC_HASH=" RSIK_SOURCE_HASH; }
int rsik_arm_consts_count(void) { return RSIK_ARM_CONSTS_COUNT; }

int rsik_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rsik_create(int device_id, rsik_ctx** out) {
    if (!out) return fail(nullptr, RSIK_E_INVALID, "rsik_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, RSIK_E_NO_DEVICE, "rsik_create: no HIP device available");
    if (device_id < 0 || device_id >= n) return fail(nullptr, RSIK_E_NO_DEVICE, "rsik_create: device id out of range");
    rsik_ctx* c = new (std::nothrow) rsik_ctx();
    if (!c) return fail(nullptr, RSIK_E_INVALID, "rsik_create: out of host memory");
    c->device = device_id;
    c->stream = nullptr;
    c->have_arm[0] = c->have_arm[1] = false;
    for (int k = 0; k < RSIK_OPT_COUNT; k++) c->options[k] = 0;
    c->ws = nullptr;
    c->ws_bytes = 0;
    c->have_side = false;
    for (auto& st : c->side) st = nullptr;
    *out = c;
    return RSIK_OK;
}

int rsik_destroy(rsik_ctx* ctx) {
    if (ctx && hipSetDevice(ctx->device) == hipSuccess) {
        if (ctx->ws) (void)hipFree(ctx->ws);
        for (void* w : ctx->retired_ws) (void)hipFree(w);
        for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
        if (ctx->have_side)
            for (hipStream_t st : ctx->side) (void)hipStreamDestroy(st);
    }
    delete ctx;
    return RSIK_OK;
}

const char* rsik_last_error(const rsik_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int rsik_set_stream(rsik_ctx* ctx, void* hip_stream) {
    if (!ctx) return RSIK_E_INVALID;
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return RSIK_OK;
}

int rsik_sync(rsik_ctx* ctx) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RSIK_OK;
}

int rsik_set_arm(rsik_ctx* ctx, int arm, const double* consts_host, int count) {
    if (!ctx) return RSIK_E_INVALID;
    if (arm != RSIK_ARM_R && arm != RSIK_ARM_L) return fail(ctx, RSIK_E_INVALID, "rsik_set_arm: arm must be 0 (r) or 1 (l)");
    if (!consts_host || count != RSIK_ARM_CONSTS_COUNT)
        return fail(ctx, RSIK_E_INVALID, "rsik_set_arm: expected RSIK_ARM_CONSTS_COUNT doubles");
    std::memcpy(ctx->arms[arm].v, consts_host, sizeof(double) * RSIK_ARM_CONSTS_COUNT);
    ctx->have_arm[arm] = true;
    return RSIK_OK;
}

int rsik_set_option(rsik_ctx* ctx, int option, int value) {
    if (!ctx) return RSIK_E_INVALID;
    if (option < 0 || option >= RSIK_OPT_COUNT) return fail(ctx, RSIK_E_INVALID, "rsik_set_option: unknown option");
    static const int max_value[RSIK_OPT_COUNT] = {RSIK_EULER_NEVER, 2, 1, 1, RSIK_CONT_RUN_STEPS, 65535};
    if (value < 0 || value > max_value[option]) return fail(ctx, RSIK_E_INVALID, "rsik_set_option: value out of range");
    ctx->options[option] = value;
    return RSIK_OK;
}
int rsik_get_option(const rsik_ctx* ctx, int option, int* value) {
    if (!ctx || !value || option < 0 || option >= RSIK_OPT_COUNT) return RSIK_E_INVALID;
    *value = ctx->options[option];
    return RSIK_OK;
}

int rsik_malloc(rsik_ctx* ctx, size_t bytes, void** dev_ptr) {
    if (!ctx || !dev_ptr) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipMalloc(dev_ptr, bytes));
    return RSIK_OK;
}
int rsik_free(rsik_ctx* ctx, void* dev_ptr) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipFree(dev_ptr));
    return RSIK_OK;
}
int rsik_memcpy_h2d(rsik_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RSIK_OK;
}
int rsik_memcpy_d2h(rsik_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!ctx) return RSIK_E_INVALID;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    RSIK_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RSIK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RSIK_OK;
}

static int check_arms(rsik_ctx* ctx, const uint8_t* arm, int arm_uniform, const char* who) {
    if (arm) {
        if (!ctx->have_arm[0] || !ctx->have_arm[1])
            return fail(ctx, RSIK_E_NOT_SET, std::string(who) + ": per-pose arm ids need both arms' constants (rsik_set_arm)");
    } else {
        if (arm_uniform != RSIK_ARM_R && arm_uniform != RSIK_ARM_L)
            return fail(ctx, RSIK_E_INVALID, std::string(who) + ": arm_uniform must be 0 (r) or 1 (l)");
        if (!ctx->have_arm[arm_uniform])
            return fail(ctx, RSIK_E_NOT_SET, std::string(who) + ": constants of the requested arm were not uploaded");
    }
    return RSIK_OK;
}

int rsik_solve(rsik_ctx* ctx, int64_t n, const double* const pose_soa[6], const uint8_t* arm, int arm_uniform,
               int theta_policy, const double* theta_in, const double* previous_joints_host, double* joints,
               double* interval, double* elbow, uint8_t* reachable, uint8_t* state) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_solve: n < 0");
    if (theta_policy < RSIK_THETA_INTERVAL0 || theta_policy > RSIK_THETA_NONE)
        return fail(ctx, RSIK_E_INVALID, "rsik_solve: unknown theta_policy");
    if ((theta_policy == RSIK_THETA_EXPLICIT || theta_policy == RSIK_THETA_FRACTION) && !theta_in && n > 0)
        return fail(ctx, RSIK_E_INVALID, "rsik_solve: theta_in is required for this theta_policy");
    int rc = check_arms(ctx, arm, arm_uniform, "rsik_solve");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!pose_soa) return fail(ctx, RSIK_E_INVALID, "rsik_solve: pose_soa is NULL");
    rsik::SolveArgs K;
    K.n = n;
    for (int k = 0; k < 6; k++) {
        if (!pose_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_solve: a pose_soa column is NULL");
        K.in[k] = pose_soa[k];
    }
    K.arm = arm;
    K.theta_policy = theta_policy;
    K.theta_in = theta_in;
    for (int k = 0; k < 7; k++) K.prev[k] = previous_joints_host ? previous_joints_host[k] : 0.0;
    K.joints = joints; K.interval = interval; K.elbow = elbow; K.reachable = reachable; K.state = state;
    if (arm) { K.arms[0] = ctx->arms[0]; K.arms[1] = ctx->arms[1]; }
    else { K.arms[0] = ctx->arms[arm_uniform]; K.arms[1] = ctx->arms[arm_uniform]; }
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t tile = (int64_t)rsik::kBlock;
    const int64_t blocks = (n + tile - 1) / tile;
    if (blocks > 0x7fffffffLL) return fail(ctx, RSIK_E_INVALID, "rsik_solve: n too large for one launch");
    dim3 grid((unsigned)blocks), block(rsik::kBlock);
    // tip offset along the goal z axis only (the default arm / the URDF): the specialised goal stage applies
    const bool tipz = K.arms[0].v[RSIK_C_TIPL] == 0.0 && K.arms[0].v[RSIK_C_TIPL + 1] == 0.0 &&
                      K.arms[1].v[RSIK_C_TIPL] == 0.0 && K.arms[1].v[RSIK_C_TIPL + 1] == 0.0 && !ctx->options[RSIK_OPT_NO_TIPZ];
    // mixed launch: do the two blocks agree in everything that has no handedness (arm_const_is_sided)?
    bool mirror = arm != nullptr && !ctx->options[RSIK_OPT_NO_MIRROR];
    for (int i = 0; mirror && i < RSIK_ARM_CONSTS_COUNT; i++)
        if (!rsik::arm_const_is_sided(i) && std::memcmp(&K.arms[0].v[i], &K.arms[1].v[i], sizeof(double)) != 0) mirror = false;
    if (arm) {
        if (mirror) {
            if (tipz) hipLaunchKernelGGL((rsik::solve_kernel<2, true>), grid, block, 0, ctx->stream, K);
            else hipLaunchKernelGGL((rsik::solve_kernel<2, false>), grid, block, 0, ctx->stream, K);
        } else {
            if (tipz) hipLaunchKernelGGL((rsik::solve_kernel<1, true>), grid, block, 0, ctx->stream, K);
            else hipLaunchKernelGGL((rsik::solve_kernel<1, false>), grid, block, 0, ctx->stream, K);
        }
    } else {
        if (tipz) hipLaunchKernelGGL((rsik::solve_kernel<0, true>), grid, block, 0, ctx->stream, K);
        else hipLaunchKernelGGL((rsik::solve_kernel<0, false>), grid, block, 0, ctx->stream, K);
    }
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

// Python float modulo (sign of the divisor), used for the l-arm limit wrap (C:243-250).
static double host_pymod(double a, double b) {
    double m = std::fmod(a, b);
    if (m != 0.0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = std::copysign(0.0, b);
    }
    return m;
}

// C:225-252: interval_limit per constrained mode, mirrored and re-wrapped for the left arm.
static void control_limits(int arm, int constrained_mode, double preferred_theta, double lim[2], double* pref) {
    const double pi = rsik::kPi;
    if (constrained_mode == RSIK_MODE_UNCONSTRAINED) { lim[0] = 3 * pi / 4; lim[1] = -2 * pi / 6; }
    else { lim[0] = -4 * pi / 5; lim[1] = 0; }
    if (arm == RSIK_ARM_L) {
        double a = -pi - lim[1], b = -pi - lim[0];
        lim[0] = a; lim[1] = b;
        if (lim[0] < -pi) lim[0] = host_pymod(lim[0], 2 * pi);
        if (lim[1] < -pi) lim[1] = host_pymod(lim[1], 2 * pi);
        if (lim[0] > pi) lim[0] = host_pymod(lim[0], -2 * pi);
        if (lim[1] > pi) lim[1] = host_pymod(lim[1], -2 * pi);
        preferred_theta = -pi - preferred_theta;
    }
    *pref = preferred_theta;
}

// The theta phase's specialised step (continuous_next_theta_lean) replaces limit_theta_to_interval's choice of the nearer
// interval end — |angle_diff(theta, l1)| < |angle_diff(theta, l0)|, U:105-111 — by one comparison with a threshold.  Here
// that threshold is found with the reference's own arithmetic (Python's float `%`), by bisection over the doubles of the
// gap, and the equivalence is then checked on a sample of the gap and on the doubles around the threshold; an interval
// for which it does not hold (or a rate limit the step's range analysis does not cover) keeps the generic step.
static double host_angle_diff(double a, double b) { return host_pymod((a - b) + rsik::kPi, 2 * rsik::kPi) - rsik::kPi; }
static int theta_snap_plan(double l0, double l1, double d_theta_max, double* tdag) {
    const double pi = rsik::kPi;
    *tdag = 0.0;
    if (!(d_theta_max >= 0.0 && d_theta_max < 3.0)) return rsik::kSnapGeneric;
    if (!(std::fabs(l0) <= pi && std::fabs(l1) <= pi)) return rsik::kSnapGeneric;
    if (l0 == l1 || (std::fabs(l0) == pi && std::fabs(l1) == pi)) return rsik::kSnapGeneric;  // the whole circle (U:468-474)
    auto nearer_is_l1 = [&](double t) { return std::fabs(host_angle_diff(t, l1)) < std::fabs(host_angle_diff(t, l0)); };
    const bool wrap = !(l0 < l1);
    // the stretch of the gap that starts at l1: up to l0 (wrap) or up to pi (the rest, (-pi, l0), must answer l0)
    double lo = l1, hi = wrap ? l0 : pi;
    if (!(lo < hi)) return rsik::kSnapGeneric;
    if (!nearer_is_l1(std::nextafter(lo, hi)) || nearer_is_l1(hi)) return rsik::kSnapGeneric;
    lo = std::nextafter(lo, hi);
    while (std::nextafter(lo, hi) < hi) {
        const double mid = lo + (hi - lo) / 2;
        if (nearer_is_l1(mid)) lo = mid; else hi = mid;
    }
    const double t = hi;  // the smallest double of the stretch for which l1 is not the nearer end
    auto agrees = [&](double x) {
        const bool valid = wrap ? (l0 <= x || x <= l1) : (l0 <= x && x <= l1);
        if (valid || !(x > -pi && x <= pi)) return true;
        const bool want = nearer_is_l1(x);
        const bool got = wrap ? (x < t) : (x >= l0 && x < t);  // (below l0 the specialised step answers l0)
        return want == got;
    };
    double x = t;
    for (int k = 0; k < 64; k++) { x = std::nextafter(x, -4.0); if (!agrees(x)) return rsik::kSnapGeneric; }
    x = t;
    for (int k = 0; k < 64; k++) { if (!agrees(x)) return rsik::kSnapGeneric; x = std::nextafter(x, 4.0); }
    const int samples = 4096;
    for (int k = 0; k <= samples; k++) {
        if (!agrees(-pi + (2 * pi) * k / samples)) return rsik::kSnapGeneric;
        if (!agrees(std::nextafter(l1, 4.0) + (t - l1) * k / samples)) return rsik::kSnapGeneric;
    }
    *tdag = t;
    return wrap ? rsik::kSnapWrap : rsik::kSnapInner;
}

static int launch_dims(rsik_ctx* ctx, int64_t n, dim3* grid, const char* who, int threads = rsik::kBlock) {
    const int64_t blocks = (n + threads - 1) / threads;
    if (blocks > 0x7fffffffLL) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": n too large for one launch");
    *grid = dim3((unsigned)blocks);
    return RSIK_OK;
}

// Can the singularity-plane half of is_elbow_ok (utils.py:459-464) fail at all?  The elbow lies on the sphere of
// radius u around the shoulder, so e_z - c e_x <= s_z - c s_x + u sqrt(1 + c^2); when that bound stays below the
// plane's right-hand side (the non-DVT offset -1.01: by a metre) the test is compiled out of the launch.
static bool singularity_plane_binds(const rsik::ArmC (&arms)[2]) {
    for (int slot = 0; slot < 2; slot++) {
        const double* c = arms[slot].v;
        const double sc = c[RSIK_C_SING_COEFF];
        const double rhs = c[RSIK_C_ES + 2] - c[RSIK_C_SING_OFFSET] - sc * c[RSIK_C_ES];
        const double reach_max = c[RSIK_C_SHOULDER + 2] - sc * c[RSIK_C_SHOULDER] + c[RSIK_C_UPPER_ARM] * std::sqrt(1.0 + sc * sc);
        if (!(rhs > reach_max + 1e-6)) return true;
    }
    return false;
}

int rsik_control_discrete(rsik_ctx* ctx, int64_t n, const double* const m12_soa[12], const uint8_t* arm,
                          int arm_uniform, int nb_search_points, double preferred_theta, int constrained_mode,
                          const double* previous_sol_host, const double* current_joints, double orbita3d_max_angle,
                          double* joints, uint8_t* reachable, uint8_t* state, uint8_t* emergency) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: n < 0");
    if (nb_search_points < 2) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: nb_search_points must be >= 2");
    if (constrained_mode != RSIK_MODE_UNCONSTRAINED && constrained_mode != RSIK_MODE_LOW_ELBOW)
        return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: unknown constrained_mode");
    if (!previous_sol_host) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: previous_sol_host is NULL");
    int rc = check_arms(ctx, arm, arm_uniform, "rsik_control_discrete");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!m12_soa || !joints) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: m12_soa / joints is NULL");
    rsik::DiscreteArgs K;
    K.n = n;
    for (int k = 0; k < 12; k++) {
        if (!m12_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_control_discrete: an m12_soa column is NULL");
        K.in[k] = m12_soa[k];
    }
    K.arm = arm;
    K.nb = nb_search_points;
    int lg = 0;
    while ((1 << lg) < nb_search_points && lg < 6) lg++;
    K.log2p = lg;
    K.sweep_mode = ctx->options[RSIK_OPT_SWEEP_MODE];  // 0 unless a test / A-B run forces one of the two strategies
    K.euler_roundtrip = ctx->options[RSIK_OPT_EULER_ROUNDTRIP];
    for (int slot = 0; slot < 2; slot++) {
        const int a = arm ? slot : arm_uniform;
        control_limits(a, constrained_mode, preferred_theta, K.lim[slot], &K.pref[slot]);
        K.pref_cs[slot] = std::cos(K.pref[slot]);  // np.cos / np.sin of the reference (U:359-360), once per launch
        K.pref_sn[slot] = std::sin(K.pref[slot]);
        for (int k = 0; k < 7; k++) K.prev_sol[slot][k] = previous_sol_host[7 * a + k];
        for (int k = 0; k < 3; k++) {
            K.prev_cs[slot][k] = std::cos(K.prev_sol[slot][4 + k]);
            K.prev_sn[slot][k] = std::sin(K.prev_sol[slot][4 + k]);
        }
        K.arms[slot] = ctx->arms[a];
    }
    const bool plane_binds = singularity_plane_binds(K.arms);
    K.current_joints = current_joints;
    K.max_angle = orbita3d_max_angle;
    K.cos_max = std::cos(orbita3d_max_angle);
    K.sin_max = std::sin(orbita3d_max_angle);
    K.joints = joints; K.reachable = reachable; K.state = state; K.emergency = emergency;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kDiscBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_control_discrete", rsik::kDiscBlock);
    if (rc != RSIK_OK) return rc;
    if (arm) {
        if (plane_binds) hipLaunchKernelGGL((rsik::control_discrete_kernel<true, true>), grid, block, 0, ctx->stream, K);
        else hipLaunchKernelGGL((rsik::control_discrete_kernel<true, false>), grid, block, 0, ctx->stream, K);
    } else {
        if (plane_binds) hipLaunchKernelGGL((rsik::control_discrete_kernel<false, true>), grid, block, 0, ctx->stream, K);
        else hipLaunchKernelGGL((rsik::control_discrete_kernel<false, false>), grid, block, 0, ctx->stream, K);
    }
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

// Arguments shared by the continuous-mode launches (validated once).
static int fill_continuous(rsik_ctx* ctx, const char* who, rsik::ContinuousArgs& K, int64_t n, const double* const m12_soa[12],
                           const double* const current_pose_m12_soa[12], const uint8_t* arm, int arm_uniform,
                           const uint8_t* timed_out, int first_timed_out, double preferred_theta,
                           const double* preferred_theta_self_host, int constrained_mode, double d_theta_max,
                           const double* current_joints, double orbita3d_max_angle, double* cont_state, double* joints,
                           uint8_t* reachable, uint8_t* state) {
    if (constrained_mode != RSIK_MODE_UNCONSTRAINED && constrained_mode != RSIK_MODE_LOW_ELBOW)
        return fail(ctx, RSIK_E_INVALID, std::string(who) + ": unknown constrained_mode");
    if (!preferred_theta_self_host) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": preferred_theta_self_host is NULL");
    int rc = check_arms(ctx, arm, arm_uniform, who);
    if (rc != RSIK_OK) return rc;
    if (!m12_soa || !cont_state || !joints) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": m12_soa / cont_state / joints is NULL");
    std::memset(&K, 0, sizeof K);
    K.n = n;
    K.first_timed_out = first_timed_out;
    for (int k = 0; k < 12; k++) {
        if (!m12_soa[k]) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": an m12_soa column is NULL");
        K.in[k] = m12_soa[k];
        K.cur_pose[k] = current_pose_m12_soa ? current_pose_m12_soa[k] : nullptr;
        if (current_pose_m12_soa && !current_pose_m12_soa[k])
            return fail(ctx, RSIK_E_INVALID, std::string(who) + ": a current_pose column is NULL");
    }
    K.arm = arm;
    K.timed_out = timed_out;
    K.euler_roundtrip = ctx->options[RSIK_OPT_EULER_ROUNDTRIP];
    for (int slot = 0; slot < 2; slot++) {
        const int a = arm ? slot : arm_uniform;
        control_limits(a, constrained_mode, preferred_theta, K.lim[slot], &K.pref_arg[slot]);
        K.pref_self[slot] = preferred_theta_self_host[a];
        K.pref_self_cs[slot] = std::cos(K.pref_self[slot]);  // np.cos / np.sin of the reference (U:359-360)
        K.pref_self_sn[slot] = std::sin(K.pref_self[slot]);
        K.arms[slot] = ctx->arms[a];
    }
    K.d_theta_max = d_theta_max;
    K.current_joints = current_joints;
    K.max_angle = orbita3d_max_angle;
    K.cos_max = std::cos(orbita3d_max_angle);
    K.sin_max = std::sin(orbita3d_max_angle);
    K.st = cont_state; K.joints = joints; K.reachable = reachable; K.state = state;
    return RSIK_OK;
}

int rsik_control_continuous_step(rsik_ctx* ctx, int64_t n, const double* const m12_soa[12],
                                 const double* const current_pose_m12_soa[12], const uint8_t* arm, int arm_uniform,
                                 const uint8_t* timed_out, double preferred_theta, const double* preferred_theta_self_host,
                                 int constrained_mode, double d_theta_max, const double* current_joints,
                                 double orbita3d_max_angle, double* cont_state, double* joints, uint8_t* reachable,
                                 uint8_t* state) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_control_continuous_step: n < 0");
    if (n == 0) {
        int rc0 = check_arms(ctx, arm, arm_uniform, "rsik_control_continuous_step");
        return rc0;
    }
    rsik::ContinuousArgs K;
    int rc = fill_continuous(ctx, "rsik_control_continuous_step", K, n, m12_soa, current_pose_m12_soa, arm, arm_uniform, timed_out,
                             0, preferred_theta, preferred_theta_self_host, constrained_mode, d_theta_max, current_joints,
                             orbita3d_max_angle, cont_state, joints, reachable, state);
    if (rc != RSIK_OK) return rc;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_control_continuous_step");
    if (rc != RSIK_OK) return rc;
    {
        const bool pb = singularity_plane_binds(K.arms);
        if (arm) { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<true, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<true, false>), grid, block, 0, ctx->stream, K); }
        else { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<false, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<false, false>), grid, block, 0, ctx->stream, K); }
    }
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

// How rsik_control_continuous_run cuts a run of n trajectories x n_steps steps into blocks, and what it needs for that.
struct ContPlan {
    int64_t T;                           // steps per block (the last one may be shorter)
    std::vector<int64_t> block_t0, block_T;
    size_t per_step, chunks_per_block, slot_bytes, carry_bytes, need;
    int slots;
    size_t n_events;
};
constexpr int kContSlots = 8;  // workspace slots in flight (block b + 8 reuses the slot of block b once its phase 5 has finished)
static int cont_plan(rsik_ctx* ctx, const char* who, int64_t n, int64_t n_steps, ContPlan& P) {
    // (the sequential phases address a block's arrays through 2 GB buffer windows: rows of n * 56 bytes, blocks of <= 384 MB
    // of workspace, i.e. <= 1.3 GB of joints; every block costs the host five launches, so blocks are as long as that allows)
    if (n > (int64_t)30 << 20) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": more than 30 Mi trajectories in one call");
    P.per_step = (size_t)n * (2 * sizeof(double) + 1);
    int64_t T_max = (int64_t)((size_t)384 << 20) / (int64_t)P.per_step;
    if (T_max < 1) T_max = 1;
    if (T_max > 65535) T_max = 65535;  // gridDim.y
    // block size: a quarter of the run (the phases of neighbouring blocks overlap: more blocks, shorter fill and drain;
    // fewer blocks, fewer of the ~12 us hand-overs between dependent launches: 4096 x 1000 steps take 0.49 / 0.48 / 0.46 /
    // 0.48 / 0.50 ms with blocks of 128 / 192 / 256 / 512 / 1000 steps), a multiple of the theta batch and of the joint
    // chunk; RSIK_OPT_CONT_BLOCK_STEPS overrides
    int64_t T = ctx->options[RSIK_OPT_CONT_BLOCK_STEPS] > 0 ? ctx->options[RSIK_OPT_CONT_BLOCK_STEPS] : (n_steps + 3) / 4;
    if (T < 64 && ctx->options[RSIK_OPT_CONT_BLOCK_STEPS] == 0) T = 64;
    T = (T + rsik::kSeqBatch - 1) / rsik::kSeqBatch * rsik::kSeqBatch;
    if (T > T_max) T = T_max >= rsik::kSeqBatch ? T_max / rsik::kSeqBatch * rsik::kSeqBatch : T_max;
    if (T > n_steps) T = n_steps;
    P.T = T;
    P.block_t0.clear(); P.block_T.clear();
    for (int64_t t0 = 0; t0 < n_steps; t0 += T) {
        P.block_t0.push_back(t0);
        P.block_T.push_back(n_steps - t0 < T ? n_steps - t0 : T);
    }
    const int64_t n_blocks = (int64_t)P.block_t0.size();
    P.chunks_per_block = ((size_t)T + rsik::kJointChunk - 1) / rsik::kJointChunk;
    P.slot_bytes = (((size_t)T * P.per_step + P.chunks_per_block * (size_t)n * 9 + 8 + 255) / 256) * 256;
    P.slots = n_blocks < kContSlots ? (int)n_blocks : kContSlots;
    P.carry_bytes = (((size_t)n * sizeof(double) + 255) / 256) * 256;
    P.need = P.slot_bytes * P.slots + P.carry_bytes;
    P.n_events = 2 + 5 * (size_t)n_blocks;
    return RSIK_OK;
}
// Workspace, side streams and events for a plan.  Nothing here may happen while the caller's stream is capturing (device
// allocation, stream and event creation are not capturable): a capture needs rsik_control_continuous_reserve, or an
// earlier run of at least this size, first.  An outgrown workspace is retired, not freed: a hipGraph captured earlier
// still points into it.
static int cont_resources(rsik_ctx* ctx, const char* who, const ContPlan& P) {
    const bool grow = ctx->ws_bytes < P.need, streams = !ctx->have_side, events = ctx->events.size() < P.n_events;
    if (!grow && !streams && !events) return RSIK_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(ctx, RSIK_E_INVALID, std::string(who) + ": the stream is capturing and this run needs a larger workspace / its streams / "
                    "more events than the context holds: call rsik_control_continuous_reserve(ctx, n, n_steps) before the capture");
    if (grow) {
        void* fresh = nullptr;
        RSIK_HIP(ctx, hipMalloc(&fresh, P.need));
        if (ctx->ws) ctx->retired_ws.push_back(ctx->ws);
        ctx->ws = fresh;
        ctx->ws_bytes = P.need;
    }
    if (streams) {
        for (auto& st : ctx->side) RSIK_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ctx->have_side = true;
    }
    while (ctx->events.size() < P.n_events) {
        hipEvent_t e;
        // (hipEventReleaseToDevice / hipEventDisableSystemFence measured: 0.443 / 0.428 against 0.429-0.439 ms per pass — the
        // ~12 us between dependent launches on different streams are not the cache write-back of the event's release)
        RSIK_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->events.push_back(e);
    }
    return RSIK_OK;
}

int rsik_control_continuous_reserve(rsik_ctx* ctx, int64_t n, int64_t n_steps) {
    const char* who = "rsik_control_continuous_reserve";
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0 || n_steps < 0) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": negative size");
    if (n == 0 || n_steps == 0) return RSIK_OK;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    ContPlan P;
    int rc = cont_plan(ctx, who, n, n_steps, P);
    if (rc != RSIK_OK) return rc;
    return cont_resources(ctx, who, P);
}

// The whole trajectory batch: the phased pipeline (include/rsik.h), or — RSIK_CONT_RUN_STEPS — one launch of the step
// kernel per control step.
int rsik_control_continuous_run(rsik_ctx* ctx, int64_t n, int64_t n_steps, const double* m12_steps,
                                const double* const current_pose_m12_soa[12], const uint8_t* arm, int arm_uniform,
                                int first_step_timed_out, double preferred_theta, const double* preferred_theta_self_host,
                                int constrained_mode, double d_theta_max, const double* current_joints,
                                double orbita3d_max_angle, double* cont_state, double* joints_steps,
                                uint8_t* reachable_steps, uint8_t* state_steps) {
    const char* who = "rsik_control_continuous_run";
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0 || n_steps < 0) return fail(ctx, RSIK_E_INVALID, "rsik_control_continuous_run: negative size");
    if (n == 0 || n_steps == 0) return check_arms(ctx, arm, arm_uniform, who);
    if (!m12_steps || !joints_steps) return fail(ctx, RSIK_E_INVALID, "rsik_control_continuous_run: NULL buffer");
    const double* cols[12];
    for (int c = 0; c < 12; c++) cols[c] = m12_steps + (size_t)c * (size_t)n;  // step 0; step s is 12 n doubles further
    rsik::ContinuousArgs K0;
    int rc = fill_continuous(ctx, who, K0, n, cols, current_pose_m12_soa, arm, arm_uniform, nullptr, first_step_timed_out ? 1 : 0,
                             preferred_theta, preferred_theta_self_host, constrained_mode, d_theta_max, current_joints,
                             orbita3d_max_angle, cont_state, joints_steps, reachable_steps, state_steps);
    if (rc != RSIK_OK) return rc;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, who);
    if (rc != RSIK_OK) return rc;
    // is_reachable_no_limits can only fail (C:385-387) for a projection margin that lets the pulled-back wrist land beyond
    // u + f (S:343-345); the pipeline's phases do not carry that outcome, the step kernel does.
    bool no_limits_can_fail = false;
    for (int slot = 0; slot < 2; slot++) no_limits_can_fail = no_limits_can_fail || !(K0.arms[slot].v[RSIK_C_PROJ_MARGIN] > 1e-12);
    if (ctx->options[RSIK_OPT_CONT_RUN_MODE] == RSIK_CONT_RUN_STEPS || no_limits_can_fail) {
        for (int64_t k = 0; k < n_steps; k++) {
            rsik::ContinuousArgs K = K0;
            for (int c = 0; c < 12; c++) K.in[c] = m12_steps + ((size_t)k * 12 + c) * (size_t)n;
            if (k > 0) {
                K.first_timed_out = 0;
                K.current_joints = nullptr;
                for (int c = 0; c < 12; c++) K.cur_pose[c] = nullptr;
            }
            K.joints = joints_steps + (size_t)k * n * 7;
            K.reachable = reachable_steps ? reachable_steps + (size_t)k * n : nullptr;
            K.state = state_steps ? state_steps + (size_t)k * n : nullptr;
            {
                const bool pb = singularity_plane_binds(K.arms);
                if (arm) { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<true, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<true, false>), grid, block, 0, ctx->stream, K); }
                else { if (pb) hipLaunchKernelGGL((rsik::control_continuous_kernel<false, true>), grid, block, 0, ctx->stream, K); else hipLaunchKernelGGL((rsik::control_continuous_kernel<false, false>), grid, block, 0, ctx->stream, K); }
            }
        }
        RSIK_HIP(ctx, hipGetLastError());
        return RSIK_OK;
    }
    // ---- phased pipeline.  The four phases of a block run on four streams (theta on the caller's, the others on the
    // context's own), ordered by events: prepare(b) -> theta(b) -> joints(b) -> chain(b), theta(b) after theta(b-1),
    // chain(b) after chain(b-1).  The two sequential phases (a lone wave per SIMD on a few CUs) then run beside each other
    // and beside the chip-filling ones of the neighbouring blocks.  Exactly four streams: the runtime multiplexes streams
    // onto four hardware queues, and a fifth stream shares a queue with another one — measured with theta on a stream of
    // its own: theta(b + 1) queued up behind chain(b)'s wait for joints(b), 0.85 -> 1.28 ms per 1000-step pass.  (Giving
    // the sequential phases compute units of their own with hipExtStreamCreateWithCUMask was measured too: every kernel
    // got slower, 2.4 ms per pass.)
    // A run is cut into blocks of steps; up to eight workspace slots are in flight (block b + 8 reuses the slot of block b
    // once its last phase has finished).
    ContPlan P;
    if ((rc = cont_plan(ctx, who, n, n_steps, P)) != RSIK_OK) return rc;
    if ((rc = cont_resources(ctx, who, P)) != RSIK_OK) return rc;
    const std::vector<int64_t>&block_t0 = P.block_t0, &block_T = P.block_T;
    const int64_t n_blocks = (int64_t)block_t0.size();
    const size_t slot_bytes = P.slot_bytes, carry_bytes = P.carry_bytes, chunks_per_block = P.chunks_per_block;
    const int slots = P.slots;
    (void)carry_bytes;
    hipStream_t s_main = ctx->stream, s_theta = ctx->stream, s_prep = ctx->side[0], s_joints = ctx->side[1], s_chain = ctx->side[2];
    auto ev = [&](int kind, int64_t b) { return ctx->events[2 + 5 * (size_t)b + kind]; };  // 0 prepared, 1 theta, 2 joints, 3 chain, 4 turns
    // (Re)initialisation of the trajectories that start here (C:296-325: the start-up search for previous_theta, ~55 us
    // of lone waves), then the pipeline's streams join in.  The prepare phase depends on the goal matrices alone, not on
    // the trajectory state: its stream forks off BEFORE the initialisation (behind whatever the caller queued ahead of
    // this call), so prepare(0) runs beside it and theta(0) starts when both are done; the joints and chain streams fork
    // behind it.
    RSIK_HIP(ctx, hipEventRecord(ctx->events[1], s_main));
    RSIK_HIP(ctx, hipStreamWaitEvent(s_prep, ctx->events[1], 0));
    {
        // two lanes per trajectory where get_joints cannot move the solver's state (no elbow projection possible)
        const bool pair = !singularity_plane_binds(K0.arms);
        dim3 grid_init = grid;
        if (pair && (rc = launch_dims(ctx, n * 2, &grid_init, who)) != RSIK_OK) return rc;
        if (arm) { if (pair) hipLaunchKernelGGL((rsik::cont_init_kernel<true, true>), grid_init, block, 0, s_main, K0); else hipLaunchKernelGGL((rsik::cont_init_kernel<true, false>), grid_init, block, 0, s_main, K0); }
        else { if (pair) hipLaunchKernelGGL((rsik::cont_init_kernel<false, true>), grid_init, block, 0, s_main, K0); else hipLaunchKernelGGL((rsik::cont_init_kernel<false, false>), grid_init, block, 0, s_main, K0); }
    }
    RSIK_HIP(ctx, hipEventRecord(ctx->events[0], s_main));
    RSIK_HIP(ctx, hipStreamWaitEvent(s_joints, ctx->events[0], 0));
    RSIK_HIP(ctx, hipStreamWaitEvent(s_chain, ctx->events[0], 0));
    rsik::ContRunArgs R;
    std::memset(&R, 0, sizeof R);
    R.n = n;
    R.m12_steps = m12_steps;
    R.arm = arm;
    R.euler_roundtrip = K0.euler_roundtrip;
    for (int slot = 0; slot < 2; slot++) {
        R.pref_arg[slot] = K0.pref_arg[slot]; R.pref_self[slot] = K0.pref_self[slot];
        R.pref_self_cs[slot] = K0.pref_self_cs[slot]; R.pref_self_sn[slot] = K0.pref_self_sn[slot];
        R.lim[slot][0] = K0.lim[slot][0]; R.lim[slot][1] = K0.lim[slot][1];
        R.arms[slot] = K0.arms[slot];
    }
    R.d_theta_max = d_theta_max;
    R.max_angle = K0.max_angle; R.cos_max = K0.cos_max; R.sin_max = K0.sin_max;
    R.st = cont_state; R.joints = joints_steps; R.reachable = reachable_steps; R.state = state_steps;
    R.theta_carry = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + slot_bytes * slots);
    const dim3 grid8((unsigned)((n * 8 + rsik::kChainBlock - 1) / rsik::kChainBlock));  // (n <= 30 Mi: fits)
    // The order in which the host issues the launches matters: a launch + its event calls cost the host ~10 us, a block's
    // four ~50 us, and a kernel that reaches its queue late starts late whatever its dependencies say.  The critical
    // path is theta(0) -> theta(1) -> ... (and chain behind it), fed by prepare(b): so the blocks that have a workspace
    // slot of their own get their prepare + theta launches first, then their joints + chain launches; a block that
    // reuses a slot can only be issued once the chain that frees the slot has been (its event must have been recorded).
    const bool plane_binds = singularity_plane_binds(R.arms);
    // the theta phase's step, specialised for the control interval where that is proven equivalent (single-arm launches)
    int snap_kind = rsik::kSnapGeneric;
    if (!arm) snap_kind = theta_snap_plan(R.lim[0][0], R.lim[0][1], d_theta_max, &R.snap_tdag);
    auto set_block = [&](int64_t b) {
        R.t0 = block_t0[b];
        R.T = block_T[b];
        R.first_block = b == 0;
        R.last_block = b == n_blocks - 1;
        R.ws = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + slot_bytes * (size_t)(b % slots));
        R.gw = R.ws + (size_t)R.T * (size_t)n;
        R.flags = reinterpret_cast<uint8_t*>(R.gw + (size_t)R.T * (size_t)n);
        R.chunk_event = R.flags + (size_t)R.T * (size_t)n;
        // (8-byte rows: the workspace slots are 256-byte aligned and flags + chunk events end on a multiple of 8 when n is;
        // otherwise round up)
        R.chunk_turns = reinterpret_cast<int8_t*>((reinterpret_cast<uintptr_t>(R.chunk_event + chunks_per_block * (size_t)n) + 7) & ~(uintptr_t)7);
    };
    auto issue_front = [&](int64_t b) -> int {  // prepare(b), theta(b)
        set_block(b);
        const dim3 grid2(grid.x, (unsigned)R.T);
        if (b >= slots) RSIK_HIP(ctx, hipStreamWaitEvent(s_prep, ev(4, b - slots), 0));  // the slot's previous block is done
        if (arm) { if (plane_binds) hipLaunchKernelGGL((rsik::cont_prepare_kernel<true, true>), grid2, block, 0, s_prep, R); else hipLaunchKernelGGL((rsik::cont_prepare_kernel<true, false>), grid2, block, 0, s_prep, R); }
        else { if (plane_binds) hipLaunchKernelGGL((rsik::cont_prepare_kernel<false, true>), grid2, block, 0, s_prep, R); else hipLaunchKernelGGL((rsik::cont_prepare_kernel<false, false>), grid2, block, 0, s_prep, R); }
        RSIK_HIP(ctx, hipEventRecord(ev(0, b), s_prep));
        RSIK_HIP(ctx, hipStreamWaitEvent(s_theta, ev(0, b), 0));
        const dim3 grid_t((unsigned)((n + rsik::kThetaBlock - 1) / rsik::kThetaBlock)), block_t(rsik::kThetaBlock);
        if (arm) hipLaunchKernelGGL((rsik::cont_theta_kernel<true, rsik::kSnapGeneric>), grid_t, block_t, 0, s_theta, R);
        else if (snap_kind == rsik::kSnapInner) hipLaunchKernelGGL((rsik::cont_theta_kernel<false, rsik::kSnapInner>), grid_t, block_t, 0, s_theta, R);
        else if (snap_kind == rsik::kSnapWrap) hipLaunchKernelGGL((rsik::cont_theta_kernel<false, rsik::kSnapWrap>), grid_t, block_t, 0, s_theta, R);
        else hipLaunchKernelGGL((rsik::cont_theta_kernel<false, rsik::kSnapGeneric>), grid_t, block_t, 0, s_theta, R);
        RSIK_HIP(ctx, hipEventRecord(ev(1, b), s_theta));
        return RSIK_OK;
    };
    auto issue_back = [&](int64_t b) -> int {  // joints(b), chain(b)
        set_block(b);
        // (a wave = 8 trajectories x 8 steps: n / 8 groups, 4 per workgroup)
        const dim3 grid2((unsigned)((n + 8 * (rsik::kBlock / 64) - 1) / (8 * (rsik::kBlock / 64))), (unsigned)((R.T + rsik::kJointChunk - 1) / rsik::kJointChunk));
        RSIK_HIP(ctx, hipStreamWaitEvent(s_joints, ev(1, b), 0));
        if (arm) hipLaunchKernelGGL(rsik::cont_joints_kernel<true>, grid2, block, 0, s_joints, R);
        else hipLaunchKernelGGL(rsik::cont_joints_kernel<false>, grid2, block, 0, s_joints, R);
        RSIK_HIP(ctx, hipEventRecord(ev(2, b), s_joints));
        RSIK_HIP(ctx, hipStreamWaitEvent(s_chain, ev(2, b), 0));
        if (arm) hipLaunchKernelGGL(rsik::cont_chain_kernel<true>, grid8, dim3(rsik::kChainBlock), 0, s_chain, R);
        else hipLaunchKernelGGL(rsik::cont_chain_kernel<false>, grid8, dim3(rsik::kChainBlock), 0, s_chain, R);
        RSIK_HIP(ctx, hipEventRecord(ev(3, b), s_chain));
        // phase 5 is independent parallel work behind chain(b): on the prepare stream where no prepare launch will be issued
        // after it (every block of a run that has a workspace slot per block, else the last block only: a prepare must not
        // queue behind it), so that the chain stream goes straight on to chain(b + 1); otherwise on the chain stream
        hipStream_t s_turns = (n_blocks <= slots || b == n_blocks - 1) ? s_prep : s_chain;
        if (s_turns != s_chain) RSIK_HIP(ctx, hipStreamWaitEvent(s_turns, ev(3, b), 0));
        hipLaunchKernelGGL(rsik::cont_turns_kernel, dim3(grid.x, (unsigned)((R.T + rsik::kJointChunk - 1) / rsik::kJointChunk)), block, 0, s_turns, R);
        RSIK_HIP(ctx, hipEventRecord(ev(4, b), s_turns));
        return RSIK_OK;
    };
    const int64_t head = n_blocks < slots ? n_blocks : slots;
    for (int64_t b = 0; b < head; b++)
        if ((rc = issue_front(b)) != RSIK_OK) return rc;
    for (int64_t b = 0; b < head; b++)
        if ((rc = issue_back(b)) != RSIK_OK) return rc;
    for (int64_t b = head; b < n_blocks; b++) {
        if ((rc = issue_front(b)) != RSIK_OK) return rc;
        if ((rc = issue_back(b)) != RSIK_OK) return rc;
    }
    // the caller's stream continues once the last chain (hence every phase of every block) is done
    RSIK_HIP(ctx, hipStreamWaitEvent(s_main, ev(3, n_blocks - 1), 0));
    RSIK_HIP(ctx, hipStreamWaitEvent(s_main, ev(4, n_blocks - 1), 0));  // (the turns kernels of one stream run in order)
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_matrix_to_pose(rsik_ctx* ctx, int64_t n, const double* const m12_soa[12], int identity_shortcut,
                        double* const pose_soa[6]) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: n < 0");
    if (n == 0) return RSIK_OK;
    if (!m12_soa || !pose_soa) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: NULL column table");
    rsik::MatrixToPoseArgs K;
    K.n = n;
    K.identity_shortcut = identity_shortcut;
    for (int k = 0; k < 12; k++) {
        if (!m12_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: an m12_soa column is NULL");
        K.in[k] = m12_soa[k];
    }
    for (int k = 0; k < 6; k++) {
        if (!pose_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_matrix_to_pose: a pose_soa column is NULL");
        K.out[k] = pose_soa[k];
    }
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    int rc = launch_dims(ctx, n, &grid, "rsik_matrix_to_pose");
    if (rc != RSIK_OK) return rc;
    hipLaunchKernelGGL(rsik::matrix_to_pose_kernel, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

static int fill_state_args(rsik_ctx* ctx, rsik::StateArgs* K, int64_t n, const uint8_t* arm, int arm_uniform,
                           const char* who) {
    if (n < 0) return fail(ctx, RSIK_E_INVALID, std::string(who) + ": n < 0");
    int rc = check_arms(ctx, arm, arm_uniform, who);
    if (rc != RSIK_OK) return rc;
    std::memset(K, 0, sizeof *K);
    K->n = n;
    K->arm = arm;
    for (int slot = 0; slot < 2; slot++) K->arms[slot] = ctx->arms[arm ? slot : arm_uniform];
    return RSIK_OK;
}

int rsik_reach_state(rsik_ctx* ctx, int64_t n, const double* const pose_soa[6], const uint8_t* arm, int arm_uniform,
                     int no_limits, double* solver_state, double* interval, uint8_t* reachable, uint8_t* state) {
    if (!ctx) return RSIK_E_INVALID;
    rsik::StateArgs K;
    int rc = fill_state_args(ctx, &K, n, arm, arm_uniform, "rsik_reach_state");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!pose_soa || !solver_state) return fail(ctx, RSIK_E_INVALID, "rsik_reach_state: pose_soa / solver_state is NULL");
    for (int k = 0; k < 6; k++) {
        if (!pose_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_reach_state: a pose_soa column is NULL");
        K.in[k] = pose_soa[k];
    }
    K.no_limits = no_limits ? 1 : 0;
    K.solver_state = solver_state; K.interval = interval; K.reachable = reachable; K.state = state;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_reach_state");
    if (rc != RSIK_OK) return rc;
    if (arm) hipLaunchKernelGGL(rsik::reach_state_kernel<true>, grid, block, 0, ctx->stream, K);
    else hipLaunchKernelGGL(rsik::reach_state_kernel<false>, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_joints_from_state(rsik_ctx* ctx, int64_t n, double* solver_state, const uint8_t* arm, int arm_uniform,
                           const double* theta, const double* previous_joints, double* joints, double* elbow) {
    if (!ctx) return RSIK_E_INVALID;
    rsik::StateArgs K;
    int rc = fill_state_args(ctx, &K, n, arm, arm_uniform, "rsik_joints_from_state");
    if (rc != RSIK_OK) return rc;
    if (n == 0) return RSIK_OK;
    if (!solver_state || !theta)  // joints may be NULL: the row's slots 24-30 carry them too
        return fail(ctx, RSIK_E_INVALID, "rsik_joints_from_state: solver_state / theta is NULL");
    K.solver_state = solver_state; K.theta = theta; K.prev = previous_joints; K.joints = joints; K.elbow = elbow;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, "rsik_joints_from_state");
    if (rc != RSIK_OK) return rc;
    if (arm) hipLaunchKernelGGL(rsik::joints_state_kernel<true>, grid, block, 0, ctx->stream, K);
    else hipLaunchKernelGGL(rsik::joints_state_kernel<false>, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_elbow_from_state(rsik_ctx* ctx, int64_t n, const double* solver_state, const double* theta, double* elbow) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_elbow_from_state: n < 0");
    if (n == 0) return RSIK_OK;
    if (!solver_state || !theta || !elbow)
        return fail(ctx, RSIK_E_INVALID, "rsik_elbow_from_state: solver_state / theta / elbow is NULL");
    rsik::StateArgs K;
    std::memset(&K, 0, sizeof K);
    K.n = n;
    K.solver_state = const_cast<double*>(solver_state); K.theta = theta; K.elbow = elbow;
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    int rc = launch_dims(ctx, n, &grid, "rsik_elbow_from_state");
    if (rc != RSIK_OK) return rc;
    hipLaunchKernelGGL(rsik::elbow_state_kernel, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

static int launch_fk(rsik_ctx* ctx, rsik::FkArgs& K, int64_t n, const uint8_t* arm, int arm_uniform, const char* who) {
    int rc = check_arms(ctx, arm, arm_uniform, who);
    if (rc != RSIK_OK) return rc;
    K.n = n;
    K.arm = arm;
    for (int slot = 0; slot < 2; slot++) K.arms[slot] = ctx->arms[arm ? slot : arm_uniform];
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    rc = launch_dims(ctx, n, &grid, who);
    if (rc != RSIK_OK) return rc;
    if (arm) hipLaunchKernelGGL(rsik::fk_kernel<true>, grid, block, 0, ctx->stream, K);
    else hipLaunchKernelGGL(rsik::fk_kernel<false>, grid, block, 0, ctx->stream, K);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

int rsik_forward_kinematics(rsik_ctx* ctx, int64_t n, const double* joints, const uint8_t* arm, int arm_uniform,
                            double* position, double* rotation) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_forward_kinematics: n < 0");
    if (n == 0) return RSIK_OK;
    if (!joints || (!position && !rotation))
        return fail(ctx, RSIK_E_INVALID, "rsik_forward_kinematics: joints or both outputs are NULL");
    rsik::FkArgs K;
    std::memset(&K, 0, sizeof K);
    K.joints = joints; K.pos = position; K.rot = rotation;
    return launch_fk(ctx, K, n, arm, arm_uniform, "rsik_forward_kinematics");
}

int rsik_fk_residual(rsik_ctx* ctx, int64_t n, int goal_kind, const double* const* goal_soa, const double* joints,
                     const uint8_t* arm, int arm_uniform, double* err) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0) return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: n < 0");
    if (goal_kind != RSIK_GOAL_POSE6 && goal_kind != RSIK_GOAL_M12)
        return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: goal_kind must be RSIK_GOAL_POSE6 or RSIK_GOAL_M12");
    if (n == 0) return RSIK_OK;
    if (!goal_soa || !joints || !err) return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: goal_soa / joints / err is NULL");
    rsik::FkArgs K;
    std::memset(&K, 0, sizeof K);
    const int cols = goal_kind == RSIK_GOAL_M12 ? 12 : 6;
    for (int k = 0; k < cols; k++) {
        if (!goal_soa[k]) return fail(ctx, RSIK_E_INVALID, "rsik_fk_residual: a goal_soa column is NULL");
        K.goal[k] = goal_soa[k];
    }
    K.goal_kind = goal_kind; K.joints = joints; K.err = err;
    return launch_fk(ctx, K, n, arm, arm_uniform, "rsik_fk_residual");
}

int rsik_debug_math(rsik_ctx* ctx, int op, int64_t n, const double* a, const double* b, double* out0, double* out1) {
    if (!ctx) return RSIK_E_INVALID;
    if (n < 0 || op < 0 || op > 8) return fail(ctx, RSIK_E_INVALID, "rsik_debug_math: bad op or n");
    if (n == 0) return RSIK_OK;
    if (!a || !out0 || ((op == 3 || op == 5 || op == 6 || op == 7) && !b)) return fail(ctx, RSIK_E_INVALID, "rsik_debug_math: NULL operand");
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid, block(rsik::kBlock);
    if (op == 8) {  // clock monitor: n waves, one per 64-thread workgroup so that they spread over the chip
        if (!out1 || n > 4096) return fail(ctx, RSIK_E_INVALID, "rsik_debug_math: op 8 needs out1 and n <= 4096 waves");
        hipLaunchKernelGGL(rsik::clock_monitor_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, a, n, out0, out1);
        RSIK_HIP(ctx, hipGetLastError());
        return RSIK_OK;
    }
    int rc = launch_dims(ctx, n, &grid, "rsik_debug_math");
    if (rc != RSIK_OK) return rc;
    hipLaunchKernelGGL(rsik::debug_math_kernel, grid, block, 0, ctx->stream, op, n, a, b, out0, out1);
    RSIK_HIP(ctx, hipGetLastError());
    return RSIK_OK;
}

// ------------------------------------------------------------------------------------------
// Multi-GPU: the all-gather of the final arrays over RCCL, for hosts without torch.distributed.
// librccl is opened at run time (dlopen), so single-GPU users never need it installed.
// ------------------------------------------------------------------------------------------
namespace {
struct NcclUid { char internal[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(NcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, NcclUid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl* rccl() {
    static Rccl R;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) {
            R.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (R.so) break;
        }
        if (!R.so) { R.err = "librccl.so not found (dlopen)"; return &R; }
        R.GetUniqueId = (int (*)(NcclUid*))dlsym(R.so, "ncclGetUniqueId");
        R.CommInitRank = (int (*)(void**, int, NcclUid, int))dlsym(R.so, "ncclCommInitRank");
        R.CommDestroy = (int (*)(void*))dlsym(R.so, "ncclCommDestroy");
        R.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(R.so, "ncclAllGather");
        R.GetErrorString = (const char* (*)(int))dlsym(R.so, "ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) R.err = "librccl.so lacks an expected symbol";
    }
    return &R;
}
int rccl_fail(rsik_ctx* ctx, const char* what, int code) {
    Rccl* R = rccl();
    return fail(ctx, RSIK_E_HIP, std::string(what) + ": " + ((R->GetErrorString && code) ? R->GetErrorString(code) : R->err.c_str()));
}
}  // namespace

int rsik_comm_unique_id(void* id128) {
    Rccl* R = rccl();
    if (!id128 || !R->err.empty()) return fail(nullptr, RSIK_E_HIP, "rsik_comm_unique_id: " + (id128 ? R->err : std::string("NULL buffer")));
    NcclUid u;
    int rc = R->GetUniqueId(&u);
    if (rc != 0) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
    std::memcpy(id128, u.internal, sizeof u.internal);
    return RSIK_OK;
}

int rsik_comm_init_rank(rsik_ctx* ctx, int nranks, int rank, const void* id128, void** comm) {
    if (!ctx) return RSIK_E_INVALID;
    if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, RSIK_E_INVALID, "rsik_comm_init_rank: bad argument");
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(ctx, RSIK_E_HIP, "rsik_comm_init_rank: " + R->err);
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    NcclUid u;
    std::memcpy(u.internal, id128, sizeof u.internal);
    *comm = nullptr;
    int rc = R->CommInitRank(comm, nranks, u, rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    return RSIK_OK;
}

int rsik_comm_destroy(rsik_ctx* ctx, void* comm) {
    if (!ctx) return RSIK_E_INVALID;
    if (!comm) return RSIK_OK;
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(ctx, RSIK_E_HIP, "rsik_comm_destroy: " + R->err);
    int rc = R->CommDestroy(comm);
    if (rc != 0) return rccl_fail(ctx, "ncclCommDestroy", rc);
    return RSIK_OK;
}

int rsik_allgather(rsik_ctx* ctx, void* comm, const void* send, void* recv, size_t bytes_per_rank) {
    if (!ctx) return RSIK_E_INVALID;
    if (!comm || !recv || (!send && bytes_per_rank)) return fail(ctx, RSIK_E_INVALID, "rsik_allgather: NULL argument");
    if (bytes_per_rank == 0) return RSIK_OK;
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(ctx, RSIK_E_HIP, "rsik_allgather: " + R->err);
    RSIK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = R->AllGather(send, recv, bytes_per_rank, /*ncclInt8*/ 0, comm, ctx->stream);
    if (rc != 0) return rccl_fail(ctx, "ncclAllGather", rc);
    return RSIK_OK;
}

}  // extern "C"
