// rsik_kernel_fused.hpp — rsik_control_continuous_run as ONE self-scheduling launch
// (one translation unit: included by rsik_lib.hip behind rsik_kernel_pipeline.hpp, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// The phased pipeline (rsik_kernel_pipeline.hpp) is four kernels per block of steps on four streams, tied by events, and on
// this runtime every dependency between two launches costs the dependent one 15-55 us: a third of a 4096 x 1000 pass was
// hand-overs.  Here the same four bodies (cont_prepare_step, cont_theta_walk, cont_joints_chunk, cont_chain_walk — the same
// device code, so the same bits) run inside ONE launch that resolves the dependencies itself.
//
// Workgroups of 1024 threads — sixteen waves, four per SIMD at the 128 registers the launch bounds allow — so that a
// workgroup owns its compute unit.  A workgroup learns what it is when it starts, from an atomic counter:
//   * the first ceil(G / 4) to arrive (G = groups of 64 trajectories) are THETA workgroups, four groups each: per group one
//     wave walks the run's steps (the recurrence on previous_theta is a chain of dependent instructions; the wave has its
//     SIMD almost to itself), fed through an LDS ring by a loader wave and relieved of its results by a writer wave (below);
//   * every other workgroup is sixteen independent WORKER waves.  A worker takes tickets from a second atomic counter; ticket
//     k names one item of work, and the tickets are laid out in dependency order — round by round: the chain items of block
//     r - L - 1, the joints items of block r - L, the prepare items of block r (L = look-ahead) — so that everything an item
//     waits for is either a ticket below its own (held by a wave that is running, or done) or a theta wave (running since
//     before the first ticket was handed out): no wave ever waits for work that has not been started.  That holds for any grid
//     size and whatever else occupies the chip; the roles are taken by workgroups that RUN, in the order they start.
//       prepare item (b, g, sub)   Sp steps of block b for group g: goals, wrapped goals, flags -> workspace;  signals pdone[b][g]
//       joints item  (b, g, c)     chunk c (8 steps) of block b for group g, eight waves' worth of cont_joints_chunk in a row;
//                                  waits for tprog[g] > b (the theta wave's progress), signals jdone[b][g]
//       chain item   (b, g, h)     block b for the trajectories 64 g + 8 h ... + 7 (eight lanes each): waits for jdone[b][g] and
//                                  for cprog[g][h] = b (the item of the block before: the trajectory state travels in cont_state)
// Every wait is bounded (three seconds of the 100 MHz clock): a wave whose wait runs out raises the abort word, which every
// other wait polls too, and the grid drains; the host sees the word at the next rsik_sync.
//
// What crosses from one item to another crosses compute units — XCDs — whose L2 caches are not coherent with each other for
// ordinary accesses: the goal / theta / flag / event arrays, the joints rows and the trajectory state are written through and
// read past the L2 (the COH forms of the shared bodies), a producer waits for its stores to be acknowledged before it
// signals, a consumer's loads are issued behind the poll that saw the signal.  Counters and the chain's fp64 atomic adds
// are agent-scope read-modify-writes, which the hardware performs where all XCDs see them.
// ------------------------------------------------------------------------------------------
constexpr int kFusedThreads = 1024;          // sixteen waves: one workgroup per compute unit
constexpr int kFusedWaves = kFusedThreads / 64;
constexpr int kFusedThetaBatch = 8;          // (two register sets of 2 x 8 doubles: the launch's 128-register budget)
constexpr int kFusedChainBatch = 8;
// words of the sync area (zeroed before every launch): three counters on lines of their own, then the arrays
constexpr int kSyncRole = 0, kSyncTicket = 32, kSyncAbort = 64, kSyncChain = 96, kSyncAlive = 100, kSyncArrays = 128;
constexpr unsigned long long kFusedWaitTicks = 300000000ull;  // 3 s of s_memrealtime (100 MHz)
// analysis builds: compile only some of the roles (bit 0 prepare, 1 joints, 2 chain, 3 theta) to see what each costs in registers
#ifndef RSIK_FUSED_ROLES
#define RSIK_FUSED_ROLES 15
#endif

struct FusedArgs {
    ContRunArgs R;              // t0 = 0, T = n_steps; ws / gw / flags / chunk_event cover the whole run
    unsigned* sync;
    unsigned long long* trace;  // diagnostic builds' per-item records, or NULL
    double* scratch;            // n doubles nothing reads (cont_theta_walk)
    int S, Sp, L, CL;           // steps per block, steps per prepare item, look-ahead of the prepare items over the joints items
                                // and lag of the chain items behind them, in blocks
    int B, G;                   // blocks of the run, groups of 64 trajectories
    int PI, CH, JQ, Jh;         // prepare items / chunks per (block, group); joints items per chunk, of Jh sub-groups each
    int snap_kind;              // the theta step's form (single-arm launches), see theta_snap_plan
    int theta_wgs;              // workgroups that take the theta role
    int chain_waves;            // waves per worker workgroup that try for a chain sub-group first (see fused_chain_wave)
    int flags_mode;             // the launch is one of rsik_kernel_flags.hpp's: pdone / jdone count workgroups, see there
    const unsigned* init_word;  // flags mode: reaches init_seq once the (re)initialisation kernel has finished (written by its stream)
    unsigned init_seq;
    const unsigned* jwords;     // flags mode: word b reaches init_seq once the joints launch of block b has completed (its stream writes it)
    unsigned tickets;           // items of the run
    unsigned trace_cap;
};

// The launch's argument block, re-derived from the kernel-argument segment behind an opaque asm: a persistent loop makes
// every scalar load of it loop-invariant, the compiler hoists all of them (pointers, launch constants, the arm's 53
// constants) to the top of the kernel, runs out of scalar registers and parks them in vector-register lanes — 460 spilled
// scalars, a v_readlane per use inside the hot loops (measured: the prepare items 60 % slower than the phased kernel's).
// Behind the asm the loads belong to the item that uses them, as in a kernel that runs once.
// (fk: the kernel's own view of the segment — __builtin_amdgcn_kernarg_segment_ptr() is null inside a function that is not a
// kernel, so the role functions receive it as an argument.)
typedef const __attribute__((address_space(4))) FusedArgs* FusedArgsK;
__device__ __forceinline__ const FusedArgs& fused_args(FusedArgsK p) {
    asm volatile("" : "+s"(p));
    return *(const FusedArgs*)p;
}
// (a function's arguments arrive in vector registers: what is the same in every lane goes back to scalar ones)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ FusedArgsK uniform(FusedArgsK p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (FusedArgsK)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ unsigned sync_load(const unsigned* p) {
    return __hip_atomic_load(const_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sync_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sync_add(unsigned* p, unsigned v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a producer's stores have been acknowledged (they were written through: another XCD's reads see them) — then the signal
__device__ __forceinline__ void stores_done() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    asm volatile("" ::: "memory");
}
// Waits until *p >= want.  False: the wait ran out, or another wave's did (the abort word is raised; the caller leaves).
__device__ __forceinline__ bool sync_wait(const FusedArgs& F, const unsigned* p, unsigned want) {
    if (sync_load(p) >= want) { asm volatile("" ::: "memory"); return true; }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        __builtin_amdgcn_s_sleep(8);
        if (sync_load(p) >= want) break;
        if (sync_load(F.sync + kSyncAbort) != 0) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > kFusedWaitTicks) {
            // which wait it was (word offset in the sync area), what it wanted and what it saw: rsik_sync reports them
            if (__hip_atomic_exchange(F.sync + kSyncAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                sync_store(F.sync + kSyncAbort + 1, (unsigned)(p - F.sync));
                sync_store(F.sync + kSyncAbort + 2, want);
                sync_store(F.sync + kSyncAbort + 3, sync_load(p));
            }
            return false;
        }
    }
    asm volatile("" ::: "memory");  // (the loads behind the wait are issued behind it)
    return true;
}

// items handed out before round r: the joints items of blocks < r - L, the prepare items of blocks < r
__device__ __forceinline__ long long fused_round_start(const FusedArgs& F, int r) {
    auto clampB = [&](int x) { return x < 0 ? 0 : (x > F.B ? F.B : x); };
    return (long long)F.G * ((long long)F.PI * clampB(r) + (long long)F.CH * F.JQ * clampB(r - F.L));
}

struct FusedItem { int kind, b, g, sub; };  // kind 0 chain, 1 joints, 2 prepare
__device__ __forceinline__ FusedItem fused_decode(const FusedArgs& F, unsigned k) {
    // the round: the last r with round_start(r) <= k (rounds 0 ... B + L + 1; monotone: bisection, wave-uniform scalar work)
    int lo = 0, hi = F.B + F.L;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (fused_round_start(F, mid) <= (long long)k) lo = mid; else hi = mid - 1;
    }
    const int r = lo;
    unsigned rest = (unsigned)((long long)k - fused_round_start(F, r));
    FusedItem it;
    const int bj = r - F.L;
    if (bj >= 0 && bj < F.B) {
        const unsigned per = (unsigned)F.CH * (unsigned)F.JQ, w = (unsigned)F.G * per;
        if (rest < w) { it.kind = 1; it.b = bj; it.g = (int)(rest / per); it.sub = (int)(rest % per); return it; }  // sub = chunk * JQ + part
        rest -= w;
    }
    it.kind = 2; it.b = r; it.g = (int)(rest / (unsigned)F.PI); it.sub = (int)(rest % (unsigned)F.PI);
    return it;
}

// diagnostic: one record per item / theta block (scripts/fused_timeline.py): start, end (100 MHz), what, where
__device__ __forceinline__ void fused_trace(const FusedArgs& F, int kind, int b, int g, int sub, unsigned long long t_start,
                                            unsigned long long t_ready) {
    if (F.trace == nullptr) return;
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        const unsigned long long at = atomicAdd(F.trace, 1ull);
        if (at < F.trace_cap) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
            unsigned long long* rec = F.trace + 4 + at * 4;
            rec[0] = t_start; rec[1] = t_ready; rec[2] = t_end;
            // kind [0:3], block [4:19], group [20:31], sub [32:39], XCC [40:43], HW_ID's low 16 bits (wave, SIMD, pipe, CU, SH, SE) [44:59]
            rec[3] = (unsigned long long)((unsigned)kind & 0xfu) | ((unsigned long long)((unsigned)b & 0xffffu) << 4) |
                     ((unsigned long long)((unsigned)g & 0xfffu) << 20) | ((unsigned long long)((unsigned)sub & 0xffu) << 32) |
                     ((unsigned long long)(xcc & 0xfu) << 40) | ((unsigned long long)(hw & 0xffffu) << 44);
        }
    }
}

// The roles are functions of their own (not inlined): one function holding all of them is allocated registers as a whole, and
// with the chain walk and the theta walk in it the prepare and joints loops — which fit the 128 registers alone — came out with
// a few dozen values spilled to scratch; behind the written-through stores a wave has in flight, every reload of one waits for
// those stores' acknowledgements too (one in-order counter): 20 us per joints iteration instead of 2.  LDS objects travel as
// LDS-address-space pointers, so that the callee's accesses stay LDS instructions.
typedef __attribute__((address_space(3))) SharedTables* LdsTabPtr;
typedef __attribute__((address_space(3))) double* LdsF64Ptr;

// ---- the theta workgroup.  The recurrence on previous_theta is a chain of ~40 dependent instructions a step; a wave that also
// fetched its operands and stored its results would wait for memory at every batch (written-through stores and the loads
// behind them share the wave's one in-order counter: 13-16 us per 64 steps instead of 6).  So the workgroup's sixteen waves
// split the job, four groups of trajectories per workgroup, per group:
//   walker  (waves 0-3, one per SIMD): reads a step's (goal, wrapped goal) from an LDS ring, computes, writes theta back into
//           the ring — no global memory instruction at all;
//   loader  (waves 4-7): goals from the workspace into the ring, three batches of eight steps in flight, as far ahead as the
//           ring (32 steps) and the prepare items' progress (pdone, polled once per block) allow;
//   writer  (waves 8-11): thetas from the ring to the workspace (written through), and the group's progress word once a
//           block's stores have been acknowledged.
// They meet in three LDS counters per group (steps filled / done / written out), polled with s_sleep in between.
constexpr int kRingSteps = 32, kRingBatch = 8;
struct ThetaRing {
    double cell[4][kRingSteps][64][2];
    unsigned filled[4], done[4], written[4], abort;
    unsigned char tagb[4][kRingSteps][64];  // PAIRS: the low byte of a step's tag (state code, flags) on its way from loader to writer
};
typedef __attribute__((address_space(3))) ThetaRing* ThetaRingPtr;
typedef __attribute__((address_space(3))) unsigned* LdsU32Ptr;
__device__ __forceinline__ unsigned lds_load(LdsU32Ptr p) { return __hip_atomic_load((unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(LdsU32Ptr p, unsigned v) { __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_done() {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS accesses have been performed
    asm volatile("" ::: "memory");
}
// waits until the LDS counter reaches `want`; false: a wait of this workgroup or of the launch ran out
__device__ __forceinline__ bool ring_wait(const FusedArgs& F, ThetaRingPtr ring, LdsU32Ptr p, unsigned want) {
    if (lds_load(p) >= want) { asm volatile("" ::: "memory"); return true; }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 1;; spins++) {
        __builtin_amdgcn_s_sleep(1);
        if (lds_load(p) >= want) break;
        if (lds_load(&ring->abort) != 0) return false;
        if ((spins & 0x3ffu) == 0) {
            if (sync_load(F.sync + kSyncAbort) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > kFusedWaitTicks) {
                sync_store(F.sync + kSyncAbort, 1u);
                lds_store(&ring->abort, 1u);
                return false;
            }
        }
    }
    asm volatile("" ::: "memory");
    return true;
}

template <bool MIXED, int KIND>
__device__ __forceinline__ void theta_walker(FusedArgsK fk, int g, int w, ThetaRingPtr ring) {
    const int lane = threadIdx.x & 63;
    const FusedArgs& F = fused_args(fk);
    const ContRunArgs& K = F.R;
    int64_t i = (int64_t)g * 64 + lane;
    if (i >= K.n) i = K.n - 1;  // (lanes past the end repeat the last trajectory)
    const int slot = MIXED ? (K.arm[i] != 0 ? 1 : 0) : 0;
    const double l0 = K.lim[slot][0], l1 = K.lim[slot][1];
    const int N = (int)K.T, S = F.S;
    const bool traced = F.trace != nullptr;
    // the state the run starts from: written (by the (re)initialisation) before the first goals were, so read behind them
    if (N > 0 && !ring_wait(F, ring, &ring->filled[w], 1u)) return;
    if (F.init_word != nullptr && !sync_wait(F, F.init_word, F.init_seq)) return;
    double prev_theta = ldc_f64<true>(&K.st[0 * K.n + i]);
    // launch constants that a select or a sign transfer needs as a vector operand: pinned in vector registers once
    const double d_max = K.d_theta_max;
    const double dmax_v = opaque(d_max), l0v = opaque(l0), l1v = opaque(l1), tdag_v = opaque(K.snap_tdag);
    auto generic = [&](double gg) { return continuous_next_theta_goal((gg != gg) ? prev_theta : gg, prev_theta, d_max, l0, l1, dmax_v, l1v); };
    auto one = [&](double gg, double gw) {
        if constexpr (KIND == kSnapGeneric) prev_theta = generic(gg);
        else prev_theta = continuous_next_theta_lean<KIND>(gg, gw, prev_theta, dmax_v, l0v, l1v, tdag_v);
        return prev_theta;
    };
    typedef double f64x2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) f64x2v* CellPtr;
    const CellPtr cells = (CellPtr)&ring->cell[w][0][lane][0];  // a step further: 64 cells further
    unsigned long long t_blk = traced ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int t = 0;
    if (KIND != kSnapGeneric && N > 0) {
        // the state a run starts from is the caller's: only from the first result on is previous_theta known to lie in
        // [-pi, pi], which the specialised step relies on — the first step alone goes through the generic form
        if (!ring_wait(F, ring, &ring->filled[w], 1u)) return;
        const f64x2v c = cells[0];
        prev_theta = generic(c.x);
        *(__attribute__((address_space(3))) double*)&cells[0] = prev_theta;
        lds_done();
        if (lane == 0) lds_store(&ring->done[w], 1u);
        t = 1;
    }
    // (the batches after that: steps t ... t + 7; with the first step gone they sit one step off the loader's, which is fine:
    // the counters count steps)
    unsigned long long waited = 0;  // (traced runs: ticks of the block spent waiting for the ring to fill)
#pragma unroll 1
    while (t + kRingBatch <= N) {
        const unsigned long long tw = traced ? __builtin_amdgcn_s_memrealtime() : 0ull;
        if (!ring_wait(F, ring, &ring->filled[w], (unsigned)(t + kRingBatch))) return;
        if (traced) waited += __builtin_amdgcn_s_memrealtime() - tw;
        f64x2v c[kRingBatch];
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) c[u] = cells[((t + u) & (kRingSteps - 1)) * 64];
        double th[kRingBatch];
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) th[u] = one(c[u].x, c[u].y);
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) *(__attribute__((address_space(3))) double*)&cells[((t + u) & (kRingSteps - 1)) * 64] = th[u];
        lds_done();
        t += kRingBatch;
        if (lane == 0) lds_store(&ring->done[w], (unsigned)t);
        if (traced && (t / S) != ((t - kRingBatch) / S)) {  // (a block's last step went by)
            const FusedArgs& Ft = fused_args(fk);
            fused_trace(Ft, 3, (t - kRingBatch) / S, g, 0, t_blk, t_blk + waited);
            t_blk = __builtin_amdgcn_s_memrealtime();
            waited = 0;
        }
    }
#pragma unroll 1
    for (; t < N; t++) {
        if (!ring_wait(F, ring, &ring->filled[w], (unsigned)(t + 1))) return;
        const f64x2v c = cells[(t & (kRingSteps - 1)) * 64];
        const double th = one(c.x, c.y);
        *(__attribute__((address_space(3))) double*)&cells[(t & (kRingSteps - 1)) * 64] = th;
        lds_done();
        if (lane == 0) lds_store(&ring->done[w], (unsigned)(t + 1));
    }
    if (traced) fused_trace(fused_args(fk), 3, (N - 1) / S, g, 1, t_blk, t_blk);
}

// The loader keeps three batches of loads in flight — and counts them itself: its loads are asm statements the compiler does
// not see as memory operations, waited for by `s_waitcnt vmcnt(N)` with N = the loads issued since (one in-order counter).
// (With loads the compiler tracks, the rare paths of this loop — the poll at a block's start, the reload of a stale batch —
// made it wait for everything at every batch: one batch per memory round trip, 230 ns a step, the walker starved half the time.)
// PAIRS (the flag-synchronised form): the goals arrive as (goal, tag) pairs that are valid by themselves — no counter to
// poll; a batch with an old tag in it is loaded again — and the wrapped goal is formed here, where issue slots are free.
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
template <bool PAIRS>
struct LoaderSet;
template <>
struct LoaderSet<true> { u32x4v p[kRingBatch]; };               // (goal, tag)
template <>
struct LoaderSet<false> { u32x2v g[kRingBatch], gw[kRingBatch]; };  // goal, wrapped goal
constexpr int kLoaderOpsPerSet = 8;  // PAIRS: one load per step; else two (see issue)
template <bool PAIRS>
__device__ __forceinline__ void loader_issue(LoaderSet<PAIRS>& o, const ContRunArgs& K, int64_t i, int t, int N) {
#pragma unroll
    for (int u = 0; u < kRingBatch; u++) {
        const int64_t row = (t + u) < N ? (t + u) : (N - 1);  // (rows past the end repeat the last one and are never used)
        if constexpr (PAIRS) {
            const double* a = K.gw + (row * K.n + i) * 2;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(o.p[u]) : "v"(a) : "memory");
        } else {
            const double* a = K.ws + row * K.n + i;
            const double* b = K.gw + row * K.n + i;
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(o.g[u]) : "v"(a) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(o.gw[u]) : "v"(b) : "memory");
        }
    }
}
// all but the `newer` most recent loads have returned; the set's registers may be read from here on
template <bool PAIRS, int NEWER>
__device__ __forceinline__ void loader_wait(LoaderSet<PAIRS>& o) {
    if constexpr (PAIRS) {
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(o.p[0]), "+v"(o.p[1]), "+v"(o.p[2]), "+v"(o.p[3]), "+v"(o.p[4]), "+v"(o.p[5]), "+v"(o.p[6]), "+v"(o.p[7])
                     : "n"(NEWER)
                     : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%16)"
                     : "+v"(o.g[0]), "+v"(o.g[1]), "+v"(o.g[2]), "+v"(o.g[3]), "+v"(o.g[4]), "+v"(o.g[5]), "+v"(o.g[6]), "+v"(o.g[7]), "+v"(o.gw[0]),
                       "+v"(o.gw[1]), "+v"(o.gw[2]), "+v"(o.gw[3]), "+v"(o.gw[4]), "+v"(o.gw[5]), "+v"(o.gw[6]), "+v"(o.gw[7])
                     : "n"(NEWER)
                     : "memory");
    }
}
template <bool PAIRS>
__device__ __forceinline__ void theta_loader(FusedArgsK fk, int g, int w, ThetaRingPtr ring) {
    const int lane = threadIdx.x & 63;
    const FusedArgs& F = fused_args(fk);
    const ContRunArgs& K = F.R;
    int64_t i = (int64_t)g * 64 + lane;
    if (i >= K.n) i = K.n - 1;
    const int N = (int)K.T, S = F.S;
    const int NB = (N + kRingBatch - 1) / kRingBatch;
    typedef double f64x2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) f64x2v* CellPtr;
    const CellPtr cells = (CellPtr)&ring->cell[w][0][lane][0];
    constexpr int kOps = PAIRS ? kRingBatch : 2 * kRingBatch;  // loads per batch
    bool ok = true;
    // a block's goals are there (the single launch: every prepare item of it has signalled; polled at a block's first batch)
    auto block_ready = [&](int t) {
        if (!PAIRS && t < N && (t % S) == 0) {
            const unsigned want = F.flags_mode ? (unsigned)((N - t) < S ? (N - t) : S) : (unsigned)F.PI;
            ok = ok && sync_wait(F, F.sync + kSyncArrays + (size_t)(t / S) * F.G + g, want);
        }
    };
    auto land = [&](LoaderSet<PAIRS>& o, int k) {
        const int t = k * kRingBatch;
        const int cnt = (N - t) < kRingBatch ? (N - t) : kRingBatch;
        double gg[kRingBatch], gw[kRingBatch];
        if constexpr (PAIRS) {
            // every pair of the batch carries this run's tag?  Else the prepare kernel has not got there yet: wait for its
            // count of the block's steps (a hint: bumped behind the stores, not waiting for them), then the whole batch again
            auto stale = [&]() {
                bool bad = false;
#pragma unroll
                for (int u = 0; u < kRingBatch; u++) {
                    const f64x2v c = __builtin_bit_cast(f64x2v, o.p[u]);
                    bad = bad || (u < cnt && !pair_tag_valid(c.y, K.epoch));
                }
                return __any(bad) != 0;
            };
            if (stale()) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                const int t_last = t + cnt - 1;
                const unsigned* hint = F.sync + kSyncArrays + (size_t)(t_last / S) * F.G + g;
                const unsigned want = (unsigned)(t_last % S) + 1u;
                do {
                    do {
                        __builtin_amdgcn_s_sleep(16);
                    } while (sync_load(hint) < want && sync_load(F.sync + kSyncAbort) == 0 && __builtin_amdgcn_s_memrealtime() - t0 <= kFusedWaitTicks);
                    if (sync_load(F.sync + kSyncAbort) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > kFusedWaitTicks) {
                        sync_store(F.sync + kSyncAbort, 1u);
                        lds_store(&ring->abort, 1u);
                        ok = false;
                        break;
                    }
                    loader_issue<PAIRS>(o, K, i, t, N);
                    loader_wait<PAIRS, 0>(o);  // (everything: the two batches behind this one have landed as well)
                } while (stale());
            }
#pragma unroll
            for (int u = 0; u < kRingBatch; u++) {
                gg[u] = __builtin_bit_cast(f64x2v, o.p[u]).x;
                gw[u] = wrap_theta_to_pi(gg[u]);  // U:93-97, see cont_prepare_step
            }
        } else {
#pragma unroll
            for (int u = 0; u < kRingBatch; u++) {
                gg[u] = __builtin_bit_cast(double, o.g[u]);
                gw[u] = __builtin_bit_cast(double, o.gw[u]);
            }
        }
        if (k >= kRingSteps / kRingBatch) ok = ok && ring_wait(F, ring, &ring->written[w], (unsigned)(t - kRingSteps + kRingBatch));
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) {
            f64x2v c;
            c.x = gg[u]; c.y = gw[u];
            if (u < cnt) cells[((t + u) & (kRingSteps - 1)) * 64] = c;
            if constexpr (PAIRS) {
                if (u < cnt) ring->tagb[w][(t + u) & (kRingSteps - 1)][lane] = (unsigned char)pair_tag_low(__builtin_bit_cast(f64x2v, o.p[u]).y, K.epoch);
            }
        }
        lds_done();
        if (lane == 0) lds_store(&ring->filled[w], (unsigned)(t + cnt));
    };
    // three batches in flight, always (past the run's end: loads of its last row, never looked at), so that "all but the last
    // two batches' loads" is the batch to land
    LoaderSet<PAIRS> a, b, c;
    block_ready(0);
    loader_issue<PAIRS>(a, K, i, 0, N);
    block_ready(kRingBatch);
    loader_issue<PAIRS>(b, K, i, kRingBatch, N);
    block_ready(2 * kRingBatch);
    loader_issue<PAIRS>(c, K, i, 2 * kRingBatch, N);
#pragma unroll 1
    for (int k = 0; k < NB && ok; k += 3) {
        loader_wait<PAIRS, 2 * kOps>(a);
        land(a, k);
        block_ready((k + 3) * kRingBatch);
        loader_issue<PAIRS>(a, K, i, (k + 3) * kRingBatch, N);
        loader_wait<PAIRS, 2 * kOps>(b);
        if (k + 1 < NB && ok) land(b, k + 1);
        block_ready((k + 4) * kRingBatch);
        loader_issue<PAIRS>(b, K, i, (k + 4) * kRingBatch, N);
        loader_wait<PAIRS, 2 * kOps>(c);
        if (k + 2 < NB && ok) land(c, k + 2);
        block_ready((k + 5) * kRingBatch);
        loader_issue<PAIRS>(c, K, i, (k + 5) * kRingBatch, N);
    }
    loader_wait<PAIRS, 0>(a);  // (nothing of this wave's is in flight when it leaves)
    loader_wait<PAIRS, 0>(b);
    loader_wait<PAIRS, 0>(c);
}

template <bool PAIRS>
__device__ __forceinline__ void theta_writer(FusedArgsK fk, int g, int w, ThetaRingPtr ring) {
    const int lane = threadIdx.x & 63;
    const FusedArgs& F = fused_args(fk);
    const ContRunArgs& K = F.R;
    int64_t i = (int64_t)g * 64 + lane;
    if (i >= K.n) i = K.n - 1;
    const int N = (int)K.T, S = F.S;
    const int64_t n = K.n;
    unsigned* const tprog = F.sync + kSyncArrays + 2 * (size_t)F.B * F.G + g;
    typedef __attribute__((address_space(3))) double* ThPtr;
    const ThPtr cells = (ThPtr)&ring->cell[w][0][lane][0];  // theta of a step: 128 doubles further per step
    int pending = -1;  // block whose last stores are in flight and whose progress word is still to be published
#pragma unroll 1
    for (int t = 0; t < N; t += kRingBatch) {
        const int cnt = (N - t) < kRingBatch ? (N - t) : kRingBatch;
        if (!ring_wait(F, ring, &ring->done[w], (unsigned)(t + cnt))) return;
        double th[kRingBatch];
        unsigned tb[kRingBatch];
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) {
            const int slot = (t + (u < cnt ? u : cnt - 1)) & (kRingSteps - 1);
            th[u] = cells[slot * 128];
            if constexpr (PAIRS) tb[u] = ring->tagb[w][slot][lane];
        }
        lds_done();
        if (lane == 0) lds_store(&ring->written[w], (unsigned)(t + cnt));  // (the slots are free: the values are in registers)
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) {
            const int64_t row = t + (u < cnt ? u : cnt - 1);  // (a short last batch stores its last step again: eight stores, always)
            if constexpr (PAIRS) st_pair(K.ws, row * n + i, th[u], pair_tag(K.epoch, 0, (int)tb[u]));  // (valid by itself: no progress word, no wait)
            else stc_f64<true>(K.ws + row * n + i, th[u]);
        }
        if constexpr (PAIRS) {
            // hints, issued behind the stores without waiting for them: the group's steps stored (a joints wave that found an old
            // tag polls it), and per block the groups that are through it (the host's streams hold the block's joints launch on it)
            if (lane == 0) {
                sync_store(tprog, (unsigned)(t + cnt));
                const int t_next = t + cnt;
                if (t_next == N || (t_next % S) == 0) sync_add(F.sync + kSyncArrays + 2 * (size_t)F.B * F.G + F.G + (size_t)((t_next - 1) / S), 1u);  // tdone[block]
            }
            continue;
        }
        if (pending >= 0) {
            // the stores of the batch before this one — a block's last — have been acknowledged once all but this batch's eight
            // have (one in-order counter): the block is published a batch late, and this wave never waits for its newest stores
            __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8)
            asm volatile("" ::: "memory");
            if (lane == 0) sync_store(tprog, (unsigned)(pending + 1));
            pending = -1;
        }
        const int t_next = t + cnt;
        if (t_next == N || (t_next % S) == 0) pending = (t_next - 1) / S;
    }
    if (pending >= 0) {
        stores_done();
        if (lane == 0) sync_store(tprog, (unsigned)(pending + 1));
    }
}

// ---- the flag-synchronised form's loader and writer (PAIRS: the goals come, and the thetas go, as tagged 16-byte pairs).
// Both share their walker's SIMD, and a wave issues one instruction every ~4.5 cycles whatever it is: what these two execute per
// batch of eight steps comes on top of the walker's ~320 instructions.  So the hot loops below are written for their
// instruction count: full batches only (the run's last, partial batch goes step by step behind them), no per-step bounds or
// block tests, rows addressed by a scalar base + a constant per-lane offset (no per-lane address arithmetic), loads counted by
// hand (see LoaderSet), validity = one 32-bit compare per pair.  (The first version of these loops — generic lambdas, clamped
// rows, per-step predicates — ran to ~950 instructions a batch: the loader alone took 2 us per batch, the walker waited for it
// half of the time, 230-500 ns a step.)
__device__ __forceinline__ void theta_loader_pairs(FusedArgsK fk, int g, int w, ThetaRingPtr ring) {
    const int lane = threadIdx.x & 63;
    const FusedArgs& F = fused_args(fk);
    const ContRunArgs& K = F.R;
    int64_t i = (int64_t)g * 64 + lane;
    if (i >= K.n) i = K.n - 1;
    const int N = (int)K.T, S = F.S;
    const unsigned E = (unsigned)K.epoch;
    const unsigned voff = (unsigned)(i * 16);
    const char* const base = reinterpret_cast<const char*>(K.gw);
    const unsigned long long row_bytes = (unsigned long long)K.n * 16ull;
    typedef double f64x2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) f64x2v* CellPtr;
    typedef __attribute__((address_space(3))) unsigned char* BytePtr;
    const CellPtr cells = (CellPtr)&ring->cell[w][0][lane][0];   // a step further: 64 cells further
    const BytePtr tags = (BytePtr)&ring->tagb[w][0][lane];        // a step further: 64 bytes further
    struct Set { u32x4v p[kRingBatch]; };
    auto issue = [&](Set& o, int t) {  // the pairs of steps t ... t + 7 (all inside the run)
        const char* row = base + (unsigned long long)t * row_bytes;
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) {
            asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(o.p[u]) : "v"(voff), "s"(row) : "memory");
            row += row_bytes;
        }
    };
    auto wait_set = [&](Set& o, auto newer) {
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(o.p[0]), "+v"(o.p[1]), "+v"(o.p[2]), "+v"(o.p[3]), "+v"(o.p[4]), "+v"(o.p[5]), "+v"(o.p[6]), "+v"(o.p[7])
                     : "n"(decltype(newer)::value)
                     : "memory");
    };
    bool ok = true;
    auto land = [&](Set& o, int t) {
        // every pair carries this run's epoch?  Else the prepare kernel has not got there yet: poll its count of the block's
        // steps (a hint), then the whole batch again
        auto stale = [&]() {
            unsigned bad = 0;
#pragma unroll
            for (int u = 0; u < kRingBatch; u++) bad |= o.p[u].w ^ E;
            return __any(bad != 0) != 0;
        };
        if (__builtin_expect(stale(), 0)) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            const int t_last = t + kRingBatch - 1;
            const unsigned* hint = F.sync + kSyncArrays + (size_t)(t_last / S) * F.G + g;
            const unsigned want = (unsigned)(t_last % S) + 1u;
            do {
                do {
                    __builtin_amdgcn_s_sleep(16);
                } while (sync_load(hint) < want && sync_load(F.sync + kSyncAbort) == 0 && __builtin_amdgcn_s_memrealtime() - t0 <= kFusedWaitTicks);
                if (sync_load(F.sync + kSyncAbort) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > kFusedWaitTicks) {
                    sync_store(F.sync + kSyncAbort, 1u);
                    lds_store(&ring->abort, 1u);
                    ok = false;
                    return;
                }
                issue(o, t);
                wait_set(o, std::integral_constant<int, 0>{});  // (everything: the batches behind this one have landed as well)
            } while (stale());
        }
        if (t >= kRingSteps) ok = ok && ring_wait(F, ring, &ring->written[w], (unsigned)(t - kRingSteps + kRingBatch));
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) {
            const unsigned long long gbits = ((unsigned long long)o.p[u].y << 32) | o.p[u].x;
            const double gg = __builtin_bit_cast(double, gbits);
            f64x2v c;
            c.x = gg;
            c.y = wrap_theta_to_pi(gg);  // U:93-97, see cont_prepare_step
            const int slot = (t + u) & (kRingSteps - 1);
            cells[slot * 64] = c;
            tags[slot * 64] = (unsigned char)o.p[u].z;
        }
        lds_done();
        if (lane == 0) lds_store(&ring->filled[w], (unsigned)(t + kRingBatch));
    };
    // three batches in flight, always: past the last full batch the last one is requested again (never looked at), so that
    // "all but the last two batches' loads" is always the batch to land
    const int nfull = N / kRingBatch;
    if (nfull > 0) {
        auto clampk = [&](int k) { return (k < nfull ? k : nfull - 1) * kRingBatch; };
        Set a, b, c;
        issue(a, clampk(0));
        issue(b, clampk(1));
        issue(c, clampk(2));
#pragma unroll 1
        for (int k = 0; k < nfull && ok; k += 3) {
            wait_set(a, std::integral_constant<int, 2 * kRingBatch>{});
            land(a, k * kRingBatch);
            issue(a, clampk(k + 3));
            wait_set(b, std::integral_constant<int, 2 * kRingBatch>{});
            if (k + 1 < nfull && ok) land(b, (k + 1) * kRingBatch);
            issue(b, clampk(k + 4));
            wait_set(c, std::integral_constant<int, 2 * kRingBatch>{});
            if (k + 2 < nfull && ok) land(c, (k + 2) * kRingBatch);
            issue(c, clampk(k + 5));
        }
        wait_set(a, std::integral_constant<int, 0>{});  // (nothing of this wave's is in flight from here on)
        wait_set(b, std::integral_constant<int, 0>{});
        wait_set(c, std::integral_constant<int, 0>{});
    }
    // the run's last steps (less than a batch), one at a time
    for (int t = nfull * kRingBatch; t < N && ok; t++) {
        const unsigned* hint = F.sync + kSyncArrays + (size_t)(t / S) * F.G + g;
        f64x2p p = ld_pair(K.gw, (int64_t)t * K.n + i);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__any(!pair_tag_valid(p.y, K.epoch))) {
            __builtin_amdgcn_s_sleep(16);
            if (sync_load(F.sync + kSyncAbort) != 0 || __builtin_amdgcn_s_memrealtime() - t0 > kFusedWaitTicks) {
                sync_store(F.sync + kSyncAbort, 1u);
                lds_store(&ring->abort, 1u);
                ok = false;
                break;
            }
            if (sync_load(hint) >= (unsigned)(t % S) + 1u) p = ld_pair(K.gw, (int64_t)t * K.n + i);
        }
        if (!ok) break;
        if (t >= kRingSteps) ok = ok && ring_wait(F, ring, &ring->written[w], (unsigned)(t - kRingSteps + 1));
        f64x2v c;
        c.x = p.x;
        c.y = wrap_theta_to_pi(p.x);
        cells[(t & (kRingSteps - 1)) * 64] = c;
        tags[(t & (kRingSteps - 1)) * 64] = (unsigned char)pair_tag_low(p.y, K.epoch);
        lds_done();
        if (lane == 0) lds_store(&ring->filled[w], (unsigned)(t + 1));
    }
}

__device__ __forceinline__ void theta_writer_pairs(FusedArgsK fk, int g, int w, ThetaRingPtr ring) {
    const int lane = threadIdx.x & 63;
    const FusedArgs& F = fused_args(fk);
    const ContRunArgs& K = F.R;
    int64_t i = (int64_t)g * 64 + lane;
    if (i >= K.n) i = K.n - 1;
    const int N = (int)K.T, S = F.S;
    const unsigned E = (unsigned)K.epoch;
    unsigned* const tprog = F.sync + kSyncArrays + 2 * (size_t)F.B * F.G + g;       // the group's steps stored: a joints wave's hint
    unsigned* const tdone = F.sync + kSyncArrays + 2 * (size_t)F.B * F.G + F.G;     // per block: groups through it (the host's gate)
    typedef __attribute__((address_space(3))) double* ThPtr;
    typedef __attribute__((address_space(3))) unsigned char* BytePtr;
    const ThPtr cells = (ThPtr)&ring->cell[w][0][lane][0];  // theta of a step: 128 doubles further per step
    const BytePtr tags = (BytePtr)&ring->tagb[w][0][lane];
    const __amdgpu_buffer_rsrc_t buf = __builtin_amdgcn_make_buffer_rsrc(K.ws, 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(i * 16);
    const unsigned row_bytes = (unsigned)(K.n * 16);  // (the pair arrays stay below 2 GB: fused_plan)
    auto store_step = [&](int t, double th, unsigned low) {  // (theta, tag): valid by itself, nobody waits for it
        u32x4v v;
        const unsigned long long tb = __builtin_bit_cast(unsigned long long, th);
        v.x = (unsigned)tb; v.y = (unsigned)(tb >> 32); v.z = low; v.w = E;
        __builtin_amdgcn_raw_buffer_store_b128(v, buf, voff, (unsigned)t * row_bytes, 16);
    };
    auto hints = [&](int t_next) {  // issued behind the stores, not waiting for them
        if (lane == 0) {
            sync_store(tprog, (unsigned)t_next);
            if (t_next == N || (t_next % S) == 0) sync_add(tdone + (t_next - 1) / S, 1u);
        }
    };
    const int nfull = N / kRingBatch;
#pragma unroll 1
    for (int k = 0; k < nfull; k++) {
        const int t = k * kRingBatch;
        if (!ring_wait(F, ring, &ring->done[w], (unsigned)(t + kRingBatch))) return;
        double th[kRingBatch];
        unsigned tb[kRingBatch];
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) {
            const int slot = (t + u) & (kRingSteps - 1);
            th[u] = cells[slot * 128];
            tb[u] = tags[slot * 64];
        }
        lds_done();
        if (lane == 0) lds_store(&ring->written[w], (unsigned)(t + kRingBatch));  // (the slots are free: the values are in registers)
#pragma unroll
        for (int u = 0; u < kRingBatch; u++) store_step(t + u, th[u], tb[u]);
        hints(t + kRingBatch);
    }
    for (int t = nfull * kRingBatch; t < N; t++) {
        if (!ring_wait(F, ring, &ring->done[w], (unsigned)(t + 1))) return;
        const int slot = t & (kRingSteps - 1);
        const double th = cells[slot * 128];
        const unsigned tb = tags[slot * 64];
        lds_done();
        if (lane == 0) lds_store(&ring->written[w], (unsigned)(t + 1));
        store_step(t, th, tb);
        hints(t + 1);
    }
}

// one wave of a theta workgroup (role = the workgroup's index among them)
template <bool MIXED, bool PAIRS = false>
__device__ __noinline__ void fused_theta_wave(FusedArgsK fk_, int role_, int wave_, ThetaRingPtr ring) {
    const FusedArgsK fk = uniform(fk_);
    const int wave = uniform(wave_), w = wave & 3;
    const int g = uniform(role_) * 4 + w;
    int n_groups, kind;
    {
        const FusedArgs& F = fused_args(fk);
        n_groups = F.G;
        kind = F.snap_kind;
    }
    if (wave >= 12 || g >= n_groups) return;
    if (wave >= 8) {
        if constexpr (PAIRS) theta_writer_pairs(fk, g, w, ring);
        else theta_writer<false>(fk, g, w, ring);
        return;
    }
    if (wave >= 4) {
        if constexpr (PAIRS) theta_loader_pairs(fk, g, w, ring);
        else theta_loader<false>(fk, g, w, ring);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    if constexpr (MIXED) {
        theta_walker<true, kSnapGeneric>(fk, g, w, ring);
    } else {
        if (kind == kSnapInner) theta_walker<false, kSnapInner>(fk, g, w, ring);
        else if (kind == kSnapWrap) theta_walker<false, kSnapWrap>(fk, g, w, ring);
        else theta_walker<false, kSnapGeneric>(fk, g, w, ring);
    }
}

// prepare item (b, g, sub): Sp steps of one group, a step per iteration, a trajectory per lane
template <bool MIXED, bool PLANE>
__device__ __noinline__ void fused_item_prepare(FusedArgsK fk_, int b_, int g_, int sub_, LdsTabPtr tab3) {
    const FusedArgsK fk = uniform(fk_);
    const int b = uniform(b_), g = uniform(g_), sub = uniform(sub_);
    SharedTables& lds_tab = *(SharedTables*)tab3;
    const int lane = threadIdx.x & 63;
    int64_t i, ii, t_first, n;
    bool live, isl;
    int steps;
    {
        const FusedArgs& F = fused_args(fk);
        n = F.R.n;
        i = (int64_t)g * 64 + lane;
        live = i < n;
        ii = live ? i : (n - 1);
        isl = MIXED ? (F.R.arm[ii] != 0) : false;
        t_first = (int64_t)b * F.S + (int64_t)sub * F.Sp;
        steps = F.Sp;
    }
    for (int s = 0; s < steps; s++) {
        const FusedArgs& Fs = fused_args(fk);
        const ContRunArgs& Ks = Fs.R;
        const int64_t t = t_first + s;
        if (t >= Ks.T) break;
        double m[12];
        const double* src = Ks.m12_steps + t * 12 * n + ii;
#pragma unroll
        for (int q = 0; q < 12; q++) m[q] = src[q * n];
        const Acc<MIXED> A = make_acc<MIXED>(Ks.arms, isl, lds_tab);
        cont_prepare_step<MIXED, PLANE, true>(Ks, A, MIXED ? (isl ? 1 : 0) : 0, m, t, t, i, live);
    }
    stores_done();
    const FusedArgs& F = fused_args(fk);
    if (lane == 0) sync_add(F.sync + kSyncArrays + (size_t)b * F.G + g, 1u);
}

// joints item (b, g, sub = chunk * JQ + part): Jh sub-groups (of eight trajectories, a wave's worth each) of one chunk of one group.
// Returns when its dependencies were met (100 MHz clock; traced runs), 0 untraced, ~0: a wait ran out.
template <bool MIXED>
__device__ __noinline__ unsigned long long fused_item_joints(FusedArgsK fk_, int b_, int g_, int sub_, LdsTabPtr tab3, LdsF64Ptr lw3) {
    const FusedArgsK fk = uniform(fk_);
    const int b = uniform(b_), g = uniform(g_), sub = uniform(sub_);
    unsigned long long t_ready = 0;
    SharedTables& lds_tab = *(SharedTables*)tab3;
    double* const lw = (double*)lw3;
    const int lane = threadIdx.x & 63;
    const FusedArgs& F = fused_args(fk);
    unsigned* const jdone = F.sync + kSyncArrays + (size_t)F.B * F.G;
    unsigned* const tprog = jdone + (size_t)F.B * F.G;
    const int chunk = sub / F.JQ, part = sub % F.JQ;
    const int64_t c = (int64_t)b * F.CH + chunk;
    const int64_t n = F.R.n;
    if (c * kJointChunk < F.R.T) {
        if (!sync_wait(F, tprog + g, (unsigned)(b + 1))) return ~0ull;
        if (F.trace) t_ready = __builtin_amdgcn_s_memrealtime();
        const int h0 = part * F.Jh, h1 = h0 + F.Jh;
        for (int h = h0; h < h1; h++) {
            const FusedArgs& Fs = fused_args(fk);
            const int64_t grp = (int64_t)g * 8 + h;
            if (grp * 8 >= n) break;
            cont_joints_chunk<MIXED, true>(Fs.R, lds_tab, lw, grp, c);
        }
        stores_done();
    }
    const FusedArgs& Fe = fused_args(fk);
    if (lane == 0) sync_add(Fe.sync + kSyncArrays + (size_t)Fe.B * Fe.G + (size_t)b * Fe.G + g, 1u);
    return t_ready;
}

// A chain wave: the recurrence on previous_sol for eight trajectories (eight lanes each), block by block behind the joints items
// of its group — ONE wave for the whole run.  (As work items, one per block, the chain was the launch's critical path: the item
// of block b + 1 waits for the item of block b, which some other wave holds, slowed by three busy waves on its SIMD and a
// hand-over through memory: 23 us a block, 380 us a run, with a thousand waves parked in such waits.)  The first waves of
// the worker workgroups to start take the sub-groups, a counter hands them out — to waves that run, like every role.
template <bool MIXED, bool PAIRS = false>
__device__ __noinline__ bool fused_chain_wave(FusedArgsK fk_, int sg_, LdsTabPtr tab3) {
    const FusedArgsK fk = uniform(fk_);
    const int sg = uniform(sg_);
    SharedTables& lds_tab = *(SharedTables*)tab3;
    const int lane = threadIdx.x & 63;
    const int g = sg >> 3, h = sg & 7;
    int n_blocks;
    {
        const FusedArgs& F = fused_args(fk);
        n_blocks = F.B;
        if ((int64_t)sg * 8 >= F.R.n) return true;
    }
    const int64_t i = (int64_t)sg * 8 + (lane >> 3);
    ChainCarry carry;  // (the trajectory state stays in registers from block to block; cont_state sees it at the end)
    carry.prev = 0.0; carry.init = false; carry.emergency = false;
    for (int b = 0; b < n_blocks; b++) {
        const FusedArgs& F = fused_args(fk);
        const ContRunArgs& K = F.R;
        unsigned* const jdone = F.sync + kSyncArrays + (size_t)F.B * F.G;
        const unsigned long long t_start = F.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const int64_t t0 = (int64_t)b * F.S;
        const int64_t T = (K.T - t0) < F.S ? (K.T - t0) : (int64_t)F.S;
        if (F.flags_mode) {
            // the block's joints launch has completed (a kernel boundary: its rows are in memory) and, before the first block,
            // the (re)initialisation
            if (b == 0 && !sync_wait(F, F.init_word, F.init_seq)) return false;
            if (!sync_wait(F, F.jwords + b, F.init_seq)) return false;
        } else if (!sync_wait(F, jdone + (size_t)b * F.G + g, (unsigned)(F.CH * F.JQ))) {
            return false;
        }
        const unsigned long long t_ready = F.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
        cont_chain_walk<MIXED, true, kFusedChainBatch, PAIRS>(K, lds_tab, i, lane & 7, t0, t0, T, b == F.B - 1, &carry, b == 0);
        fused_trace(F, 0, b, g, h, t_start, t_ready);
    }
    return true;
}

// A worker's next ticket.  The workgroup takes tickets from the launch's counter sixteen at a time and hands them to its
// waves through one 64-bit LDS word (low half: the next ticket, high half: the end of the batch): one device-scope atomic
// per sixteen items (tens of thousands of items a pass on ONE word serialise otherwise).  The wave that finds the batch
// just used up fetches the next one; the order argument of the header holds: a workgroup hands its tickets out in
// rising order, so the launch's lowest unfinished ticket is always held by a wave that runs.
__device__ __forceinline__ unsigned fused_claim(const FusedArgs& F, unsigned long long* lds_tk) {
    const int lane = threadIdx.x & 63;
    for (;;) {
        unsigned long long old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(lds_tk, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned next = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)old);
        const unsigned end = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(old >> 32));
        if (next < end) return next;
        if (next == end) {
            unsigned base = 0;
            if (lane == 0) base = __hip_atomic_fetch_add(F.sync + kSyncTicket, (unsigned)kFusedWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            if (lane == 0)
                __hip_atomic_store(lds_tk, (unsigned long long)(base + 1u) | ((unsigned long long)(base + (unsigned)kFusedWaves) << 32), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            return base;
        }
        __builtin_amdgcn_s_sleep(2);  // (another wave is fetching the batch)
    }
}

template <bool MIXED, bool PLANE>
__global__ __launch_bounds__(kFusedThreads) void cont_fused_kernel(const FusedArgs F_) {
    __shared__ SharedTables lds_tab;
    // a worker workgroup's sixteen row-staging slabs (cont_joints_chunk) and a theta workgroup's ring share the space
    __shared__ __attribute__((aligned(16))) unsigned char lds_big[sizeof(ThetaRing)];
    static_assert(sizeof(ThetaRing) >= sizeof(double) * kFusedWaves * 64 * 7, "the ring is the larger of the two");
    __shared__ unsigned lds_role;
    __shared__ unsigned long long lds_tk;
    const int wave = threadIdx.x >> 6;
    const FusedArgsK fk = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const ThetaRingPtr ring = (ThetaRingPtr)lds_big;
    if (threadIdx.x == 0) {
        lds_role = __hip_atomic_fetch_add(F_.sync + kSyncRole, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_tk = 0ull;  // (an empty batch: the first wave to ask fetches one)
    }
    if (threadIdx.x < 13) (&ring->filled[0])[threadIdx.x] = 0u;  // filled, done, written, abort
    stage_tables<MIXED, 0, kFusedThreads>(lds_tab, F_.R.arms);  // (ends with the workgroup's one barrier)
    const unsigned role = lds_role;

    if ((RSIK_FUSED_ROLES & 8) && role < (unsigned)F_.theta_wgs) {
        fused_theta_wave<MIXED>(fk, (int)role, wave, ring);
        return;
    }

    // ---- a worker wave; the last few of a workgroup look for a chain sub-group first
    const LdsTabPtr tab3 = (LdsTabPtr)&lds_tab;
    if ((RSIK_FUSED_ROLES & 4) && wave >= kFusedWaves - F_.chain_waves) {
        unsigned sg = 0;
        if ((threadIdx.x & 63) == 0) sg = __hip_atomic_fetch_add(F_.sync + kSyncChain, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sg = (unsigned)__builtin_amdgcn_readfirstlane((int)sg);
        if (sg < (unsigned)F_.G * 8u) {
            __builtin_amdgcn_s_setprio(2);
            if (!fused_chain_wave<MIXED>(fk, (int)sg, tab3)) return;
        }
    }
    const LdsF64Ptr lw3 = (LdsF64Ptr)lds_big + wave * (64 * 7);
    for (;;) {
        const FusedArgs& F = fused_args(fk);
        const unsigned k = fused_claim(F, &lds_tk);
        if (k >= F.tickets) return;
        const FusedItem it = fused_decode(F, k);
        const unsigned long long t_start = F.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
        unsigned long long t_ready = t_start;
        // issue priority: what other work waits for goes first — the chain items (the next block's chain items wait for them), then
        // the joints items, then the prepare items, which have nobody waiting but the theta waves; except at the start of a run,
        // where the first blocks' goals are what everything waits for
        if ((RSIK_FUSED_ROLES & 1) && it.kind == 2) {
            if (it.b == 0) __builtin_amdgcn_s_setprio(3);
            else if (it.b == 1) __builtin_amdgcn_s_setprio(2);
            else if (it.b == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            fused_item_prepare<MIXED, PLANE>(fk, it.b, it.g, it.sub, tab3);
        } else if ((RSIK_FUSED_ROLES & 2) && it.kind == 1) {
            __builtin_amdgcn_s_setprio(1);
            const unsigned long long t = fused_item_joints<MIXED>(fk, it.b, it.g, it.sub, tab3, lw3);
            if (t == ~0ull) return;
            if (t != 0) t_ready = t;
        }
        fused_trace(fused_args(fk), it.kind, it.b, it.g, it.sub, t_start, t_ready);
    }
}

}  // namespace rsik
