// rsik_kernel_pipeline.hpp — rsik_control_continuous_run: the four phases of the trajectory pipeline
// (one translation unit: included by rsik_lib.hip, in this order, inside nothing)
#pragma once

namespace rsik {

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
// analysis builds: the chip-filling kernels at a fixed occupancy
#ifdef RSIK_PHASED_OCC
#define RSIK_PHASED_OCC_ATTR __attribute__((amdgpu_waves_per_eu(RSIK_PHASED_OCC, RSIK_PHASED_OCC)))
#else
#define RSIK_PHASED_OCC_ATTR
#endif
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
// -DRSIK_PIPE_TIMING (diagnostic builds, scripts/probes/c5_untraced_timeline.py): every phase kernel records when its
// first sampled workgroup starts and its last one ends (100 MHz counter, atomicMin / atomicMax by one thread of every 32nd
// workgroup): the pipeline's timeline WITHOUT a profiler in the way.
#ifdef RSIK_PIPE_TIMING
struct PipeStamp {
    unsigned long long* t;
    bool on;
    __device__ PipeStamp(unsigned long long* tmin, unsigned long long* tmax, int slot)
        : t(tmax ? tmax + slot : nullptr), on(tmin && threadIdx.x == 0 && (blockIdx.x & 31) == 0 && (blockIdx.y & 3) == 0) {
        if (on) atomicMin(tmin + slot, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
    __device__ ~PipeStamp() {
        if (on) atomicMax(t, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
};
#define RSIK_PIPE_STAMP(K, phase) PipeStamp pipe_stamp_((K).tmin, (K).tmax, (K).tslot * 5 + (phase))
#define RSIK_PIPE_STAMP_AT(K, slot, phase) PipeStamp pipe_stamp_((K).tmin, (K).tmax, ((slot) < 64 ? (int)(slot) : 63) * 5 + (phase))
#else
#define RSIK_PIPE_STAMP(K, phase)
#define RSIK_PIPE_STAMP_AT(K, slot, phase)
#endif
struct ContRunArgs {
#ifdef RSIK_PIPE_TIMING
    unsigned long long *tmin, *tmax;
    int tslot;
#endif
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
    double snap_tdag;             // phase 2, single-arm launches: see continuous_next_theta_lean (the kind is a template argument)
    double* theta_carry;          // [2][n]: previous_theta between the blocks of one run (phase 2's own state); row 1: see cont_theta_kernel
    int no_turn_hint;             // (timing experiments: RSIK_OPT_CONT_PHASED_VARIANT bit 64 — the hint is written as "no turns")
    unsigned* turn_hint;          // [n]: whole turns previous_sol sits away from [-pi, pi], joints 0, 2, 4, 6 as four biased bytes: what this block's
                                  // joints kernel reads (cont_joints_chunk) — the run's own array, written by the first block's theta kernel from the
                                  // state the run begins with, or, from the ninth block on, the workspace slot's, which the chain kernel of the block
                                  // that used the slot before (eight blocks earlier) left there
    unsigned* run_turn_hint;      // the run's own array (the first block's theta kernel writes it)
    uint8_t* slot_turn_hint;      // [n][4]: the slot's array (this block's chain kernel writes it at its end)
    int first_block, last_block;
    unsigned* started_word;       // phased pipeline, launch by launch: the theta kernel of a block writes started_seq here when it starts
    unsigned started_seq;         // (the host holds the joints kernel of the block BEFORE on it, see rsik_control_continuous_run), or NULL
    const unsigned* wait_word;    // theta kernel, launch by launch: the block's "prepared" word, which the kernel itself waits for (wait_seq) —
    unsigned wait_seq;            // see cont_theta_kernel — or NULL: the stream has waited
    unsigned* timeout_word;       // ... raised if that wait gives up (rsik_sync reports it)
    unsigned* chain_started_word; // the same for the chain kernel of a block (the NEXT run's first prepare kernel is held on the last one's), or NULL
    double* st;                   // cont_state
    double* joints;               // [n_steps][n][7]
    uint8_t* reachable;           // [n_steps][n] or NULL
    uint8_t* state;               // [n_steps][n] or NULL
    ArmC arms[2];
};
#define RSIK_WS(K, t, i) (K).ws[(int64_t)(t) * (K).n + (i)]
// The pipeline's bulk stores are written through to system scope (st_stream<kStoreThrough>, rsik_kernel_solve.hpp): the end of each of
// its kernels is what a hand-over waits for, and nothing of its output is then left dirty in the L2s for the write-back at that end
// (-1.3 ... -1.5 % per pass in every launch form, bit-identical; non-temporal LOADS of the goal matrices cost 1-3 %: the joints phase reads
// what the prepare phase read).  kRowStoreAux: the same policy for the raw buffer stores (sc0 | sc1).
constexpr int kRowStoreAux = 17;
template <class T>
__device__ __forceinline__ void pipe_store(T* p, T v) { st_stream<kStoreThrough>(p, v); }

// phase 1, one (step, trajectory): `m` the step's twelve matrix entries, `t` the step's row in the workspace arrays, `t_abs`
// its row in the run's outputs.
template <bool MIXED, bool PLANE>
__device__ __forceinline__ void cont_prepare_step(const ContRunArgs& K, const Acc<MIXED>& A, int slot, const double (&m)[12], int64_t t,
                                                  int64_t t_abs, int64_t i, bool live) {
    const bool invalid = !all_finite(m);  // (judged while the twelve values are at hand)
    Rot Rg;
    V3 pos;
    bool special;
    goal_from_m12(m, Rg, pos, K.euler_roundtrip, &special);
    const Goal G = make_goal(A, Rg);
    Reach r;
    ThetaTarget T = continuous_target<PLANE, false>(A, pos, G.woff, K.pref_self[slot], K.pref_self_cs[slot], K.pref_self_sn[slot], r);
    if (!live) return;
    // rsik.h "Rows that are not numbers": for the theta phase a step whose search found nothing ("stay"); the joints phase writes no
    // joints for it and the chain phase steps over it (flag bit 4; the paired layout carries it as the state code)
    if (RSIK_RARE(invalid)) { T.ok_limits = true; T.found = false; T.code = RSIK_STATE_INVALID_INPUT; }
    // the step's goal for the theta phase: the search's theta, NaN = nothing found, stay (U:252-264 with goal =
    // previous_theta), or the preferred theta of an unreachable pose (U:115-127)
    const double goal = T.ok_limits ? (T.found ? T.theta : __builtin_nan("")) : K.pref_arg[slot];
    pipe_store(&RSIK_WS(K, t, i), goal);
    // what limit_theta_to_interval makes of theta = goal before it looks at the interval (U:93-97): this phase has the
    // issue slots for it, the theta phase (a lone wave per SIMD) has not
    pipe_store(&K.gw[t * K.n + i], wrap_theta_to_pi(goal));
    pipe_store(&K.flags[t * K.n + i], (uint8_t)((T.ok_limits ? 1 : 0) | (T.found ? 2 : 0) | (special ? 8 : 0) | (invalid ? 16 : 0)));
    if (K.state) pipe_store(&K.state[t_abs * K.n + i], (uint8_t)T.code);
    if (K.reachable) pipe_store(&K.reachable[t_abs * K.n + i], (uint8_t)((T.ok_limits && T.found) ? 1 : 0));
}

// phase 1: one thread per (trajectory, step of the block)
template <bool MIXED, bool PLANE>
__global__ __launch_bounds__(kBlock) RSIK_PHASED_OCC_ATTR void cont_prepare_kernel(const ContRunArgs K) {
    RSIK_PIPE_STAMP(K, 0);
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
    cont_prepare_step<MIXED, PLANE>(K, A, slot, m, t, K.t0 + t, i, live);
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
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(RowWords2, v), buf, lane, row, kRowStoreAux);
}
__device__ __forceinline__ int ld_row_u8(__amdgpu_buffer_rsrc_t buf, unsigned lane, unsigned row) {
    return (int)__builtin_amdgcn_raw_buffer_load_b8(buf, lane, row, 0);
}

// phase 2: one thread per trajectory walks the block's steps: the recurrence on previous_theta.
// KIND: kSnapInner / kSnapWrap = the step specialised for the launch's control interval (continuous_next_theta_lean;
// single-arm launches), kSnapGeneric = the reference's own sequence of operations for any interval.
// cont_theta_walk: T steps of one trajectory per lane; `ws` / `gw` = the rows of the first of them, `first_generic`: the first
// step goes through the generic form (the state a run starts from is the caller's), `scratch`: n doubles nothing reads.
// BATCH steps' operands are fetched at once, a batch ahead.  Returns previous_theta after the last step.
template <int KIND, int BATCH>
__device__ __forceinline__ double cont_theta_walk(const ContRunArgs& K, int64_t i, double l0, double l1, double* ws, double* gw, int64_t T,
                                                  bool first_generic, double prev_theta, double* scratch) {
    // A lone wave per SIMD: every instruction of a step is paid in full (~4.5 cycles each, rsik_device.hpp `opaque`), and
    // the memory round trip of a step's operands would double a step, so they are fetched BATCH steps at a time,
    // one batch ahead of the one being computed, into two register sets that take turns (no copies); what is left of the
    // block after the last full batch goes step by step.
    const int64_t n = K.n;
    // this trajectory's goal / theta of the step the wave is at: (wbuf, off, row), its wrapped goal (gbuf, off, row); a step
    // further is `stride` bytes further
    const __amdgpu_buffer_rsrc_t wbuf = row_buffer(ws), gbuf = row_buffer(gw);
    const unsigned off = (unsigned)(i * sizeof(double)), stride = (unsigned)(n * sizeof(double));
    unsigned row = 0;
    // launch constants that a select or a sign transfer needs as a vector operand: pinned in vector registers once
    const double dmax_v = opaque(K.d_theta_max), l0v = opaque(l0), l1v = opaque(l1), tdag_v = opaque(K.snap_tdag);
    // `g` is the step's goal as the prepare phase encoded it: the search's theta, the preferred theta for an unreachable
    // pose, NaN = "stay".  Straight-line arithmetic only.
    auto generic = [&](double g) {
        return continuous_next_theta_goal((g != g) ? prev_theta : g, prev_theta, K.d_theta_max, l0, l1, dmax_v, l1v);
    };
    auto one = [&](double g, double gw_) {
        if constexpr (KIND == kSnapGeneric) prev_theta = generic(g);
        else prev_theta = continuous_next_theta_lean<KIND>(g, gw_, prev_theta, dmax_v, l0v, l1v, tdag_v);
        return prev_theta;
    };
    int64_t left = T;
    if (KIND != kSnapGeneric && first_generic && left > 0) {
        // the state a run starts from is the caller's: only from the first result on is previous_theta known to lie in
        // [-pi, pi], which the specialised step relies on
        prev_theta = generic(ld_row_f64(wbuf, off, row));
        st_row_f64(wbuf, off, row, prev_theta);
        row += stride; left -= 1;
    }
    struct Operands { double g[BATCH], gw[BATCH]; };
    // (`valid` < BATCH: the block's last, partial batch — the steps past its end repeat the last one and are skipped)
    auto fetch = [&](Operands& o, int ahead, int valid) {
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const unsigned at = row + (unsigned)(ahead + (u < valid ? u : valid - 1)) * stride;
            o.g[u] = ld_row_f64(wbuf, off, at);
            if constexpr (KIND != kSnapGeneric) o.gw[u] = ld_row_f64(gbuf, off, at);
        }
    };
    auto compute = [&](const Operands& o, auto partial, int valid) {  // the batch at `row`; leaves `row` at the next one
        constexpr bool kPartial = decltype(partial)::value;
        const unsigned r0 = row;
        row += (unsigned)(kPartial ? valid : BATCH) * stride;
        // one wait for the whole set (it was fetched a batch ago) instead of one per operand: a wait is an issue slot too
        asm volatile("" : : "v"(o.g[BATCH - 1]), "v"(o.gw[KIND != kSnapGeneric ? BATCH - 1 : 0]));
        // the batch's results leave together, behind its last step: a store between two steps would sit between the loads of
        // the batch after this one and those of the batch after that in the wave's one in-order memory counter, and the wait
        // for the former would wait for its acknowledgement too
        double res[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            if (!kPartial || u < valid) res[u] = one(o.g[u], o.gw[u]);  // (launch-uniform: a scalar branch)
        }
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            if (!kPartial || u < valid) st_row_f64(wbuf, off, r0 + (unsigned)u * stride, res[u]);
        }
    };
    int64_t batches = left / BATCH;
    left -= batches * BATCH;
    Operands a, b;
    if (batches > 0) fetch(a, 0, BATCH);
    // Two batches per turn, each fetched while the one before it is computed.  The turn runs only while a third full batch
    // exists, so that its second fetch needs no branch around it: where a path with and a path without that fetch meet, the
    // compiler can only wait for the smaller number of operations in flight — on the path with the fetch that is the batch
    // about to be computed AND half of the one just requested, a memory round trip every 32 steps (~30 % of this phase).
    if (batches >= 3) {
        // (and the turn is entered the way it is re-entered — a batch in flight, then a batch's worth of stores — or the wait
        // at its top, where the two ways in meet, is again for the smaller count: the stores of the batch just computed)
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            scratch[i] = 0.0;  // (a row that nothing reads; a constant: no load to wait for)
            asm volatile("" ::: "memory");
        }
    }
#pragma unroll 1
    while (batches >= 3) {
        fetch(b, BATCH, BATCH);
        compute(a, std::false_type{}, BATCH);
        fetch(a, BATCH, BATCH);
        compute(b, std::false_type{}, BATCH);
        batches -= 2;
    }
    if (batches == 2) {
        fetch(b, BATCH, BATCH);
        compute(a, std::false_type{}, BATCH);
        if (left > 0) fetch(a, BATCH, (int)left);
        compute(b, std::false_type{}, BATCH);
        if (left > 0) compute(a, std::true_type{}, (int)left);
    } else if (batches == 1) {
        if (left > 0) fetch(b, BATCH, (int)left);
        compute(a, std::false_type{}, BATCH);
        if (left > 0) compute(b, std::true_type{}, (int)left);
    } else if (left > 0) {
        fetch(a, 0, (int)left);
        compute(a, std::true_type{}, (int)left);
    }
    return prev_theta;
}

template <bool MIXED, int KIND>
__global__ __launch_bounds__(kThetaBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void cont_theta_kernel(const ContRunArgs K) {
    RSIK_PIPE_STAMP(K, 1);
    static_assert(!MIXED || KIND == kSnapGeneric, "a mixed launch has an interval per lane");
    if (K.started_word != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)  // (the last workgroup runs: all of them have been placed)
        __hip_atomic_store(K.started_word, K.started_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // Launch by launch the kernel itself waits for its block's prepare kernel (round 6): a stream wait ahead of it costs 8-17 us
    // between two theta kernels — the run's critical path once its prepare phase overlaps the run before — and a theta kernel that
    // only becomes ready while a chip-filling kernel holds the chip cannot get its 276-register waves onto it before that one drains
    // (docs/experiments.md A.4); launched right behind the theta kernel of the block before, its waves are resident when the word
    // comes.  Nothing this wait depends on can be behind this kernel in any queue: the prepare kernel, the stream write that raises the
    // word and everything THEY wait for were issued before this launch (rsik_control_continuous_run's issue order), and what the
    // waiting waves hold — a wave slot and 276 registers on 64 SIMDs — none of those needs.  Bounded all the same: a second on the
    // 100 MHz counter, then the walk goes on with whatever is there and the context's timeout word says so.
    if (K.wait_word != nullptr) {
        const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
        while ((int)(__hip_atomic_load(K.wait_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - K.wait_seq) < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memrealtime() - t_begin > 100000000ull) {
                if (threadIdx.x == 0) __hip_atomic_store(K.timeout_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        // (what the prepare kernel wrote was released to memory when it ended, before the word was raised: no line of it may be
        // served from this XCD's L2 as an earlier user of the workspace slot left it)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // a serial phase: its few waves share their SIMDs with the chip-filling phases of the neighbouring blocks (other
    // streams) and must win the issue arbitration, or every instruction waits behind throughput work
    __builtin_amdgcn_s_setprio(3);
    const int64_t i = (int64_t)blockIdx.x * kThetaBlock + threadIdx.x;
    if (i >= K.n) return;
    const int slot = MIXED ? (K.arm[i] != 0 ? 1 : 0) : 0;
    // previous_theta travels from block to block in theta_carry: this phase runs ahead of phase 4 (other streams), which
    // alone decides what ends up in the state's row 0 — the theta of the last step, or of the step that latched the
    // emergency stop (C:205-210; what this phase computes for a latched trajectory is never looked at).
    if (K.first_block) {
        // Round 6: where previous_sol stands when the run begins (the start-up kernel, ahead of this one on its stream, has had its say),
        // in whole turns, for the four joints that can wind (shoulder pitch, elbow yaw, wrist roll, wrist yaw): the joints phase writes
        // its rows that many turns up, so that a trajectory that ARRIVES wound — the later chunks of a streamed trajectory, the survivors
        // of an emergency stop — finds its chunks standing where the chain phase looks for them instead of having every element of every
        // chunk moved there by an atomic add (a run that continues config 5's batch, 95 % of whose steps have a joint beyond pi by then:
        // 0.50 -> 0.43 ms, profiles/r06/config5_continuing_runs_s1.txt).  A hint only: the chain phase judges every chunk against
        // previous_sol as before, and adds what is still missing.
        unsigned packed = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double p = K.st[(1 + 2 * q) * K.n + i];
            const double h = (fabs(p) < 600.0 && !K.no_turn_hint) ? rint(p * 0.15915494309189535) : 0.0;  // (|turns| <= 100 is all the chain phase takes as quiet)
            packed |= (unsigned)((int)h + 128) << (8 * q);
        }
        K.run_turn_hint[i] = packed;
    }
    const double prev_theta = K.first_block ? K.st[0 * K.n + i] : K.theta_carry[i];
    K.theta_carry[i] = cont_theta_walk<KIND, kThetaBatch>(K, i, K.lim[slot][0], K.lim[slot][1], K.ws, K.gw, K.T, K.first_block != 0,
                                                                 prev_theta, K.theta_carry + K.n);
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
// sequential phase's business (phase 4 finds them from the chunks' first and last rows and adds them in): they are
// not zero often enough to guess — shoulder pitch and elbow yaw swing by more than pi within a few hundred steps when the
// arm passes its shoulder singularity (8 % of config 5's steps have them outside [-pi, pi]).  What the reference decides
// step by step — the continuity thresholds (U:571-589, C:398), the +-6 pi limit (U:535-568), an exact singularity that
// needs previous_sol (S:751-753, 782-784) — is only DETECTED here, with a margin of 1e-9: the chunk's event byte tells
// phase 4 to walk that chunk with the reference's own sequence of operations.  So the joints make ONE trip to HBM but
// for the shifted elements (phase 4 used to read and rewrite all of them, 112 of the 412 bytes a control step moved),
// and a quiet step's value is its raw joint plus whole turns: within 2 ulp of the reference's previous + angle_diff(raw,
// previous), no accumulation.
// One wave: chunk `c` (steps 8 c ... 8 c + 7 of the K arrays) of the trajectories 8 grp ... 8 grp + 7; `lw`: 64 x 7 doubles of LDS
// of the wave's own.  The workgroup's tables are staged here, behind the chunk's loads (their latency overlaps the staging's round trip).
template <bool MIXED>
__device__ __forceinline__ void cont_joints_chunk(const ContRunArgs& K, SharedTables& lds_tab, double* lw, int64_t grp_in, int64_t c) {
    static_assert(kJointChunk == 8, "lane = 8 * step + trajectory");
    // the group is the same for the whole wave, but derived from threadIdx (a vector register): said so, or the buffer descriptor of
    // the rows written below sits in vector registers and each of its seven stores becomes a loop over "the lanes that agree"
    // (4 v_readfirstlane + 2 compares + mask juggling: ~70 instructions a wave-step, 7 % of this phase).  (< 2^31 groups: n <= 30 Mi)
    const int64_t grp = (int64_t)__builtin_amdgcn_readfirstlane((int)grp_in);
    const int lane = threadIdx.x & 63;
    const int tl = lane & 7, sl = lane >> 3;
    const int64_t n = K.n;
    const int64_t i = grp * 8 + tl;
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
    const unsigned hint = K.turn_hint[ii];  // (whole turns previous_sol stood away from the raw angles when the run began: cont_theta_kernel)
    int flag = (int)K.flags[tt * n + ii];
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
    stage_tables<MIXED, (int)offsetof(ContRunArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, lane_isl, lds_tab);
    Reach r;
    Goal G;
    step_geometry(A, m, K.euler_roundtrip, (flag & 1) == 0, r, G, !special);
    const double zeros[7] = {0, 0, 0, 0, 0, 0, 0};
    double jv[7];
    bool sing;
    step_joints(A, K, r, G, theta, zeros, jv, sing);
    // a step without joints: singular (phase 4 recomputes it with previous_sol), or its goal is not numbers (flag bit 4: it stays
    // without, and phase 4 steps over it — rsik.h "Rows that are not numbers")
    const bool dead = sing || (flag & 16) != 0;
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
    const unsigned long long sing_mask = __ballot(dead);
    const bool sing_below = sl > 0 && ((sing_mask >> (lane - 8)) & 1ull) != 0;
    const bool ev = dead || sing_below || !(worst_a <= 0.5 - 1e-9) || !(worst_b <= 1.0 - 1e-9) || !(fabs(packed) < 4.0e9);
    unsigned word = (unsigned)packed;  // (garbage for a NaN: the chunk is an event then)
#pragma unroll
    for (int step = 1; step < 8; step *= 2) {  // inclusive prefix sum over the chunk's steps (lane stride 8)
        const unsigned w = (unsigned)__builtin_amdgcn_ds_bpermute((lane - 8 * step) << 2, (int)word);
        if (sl >= step) word += w;
    }
#pragma unroll
    for (int k = 0; k < 7; k++) turn[k] = 0.0;
    {
        const int bias = 8 * (sl + 1) + 128;  // (+ the hint's own bias)
        turn[0] = (double)((int)(word & 0xffu) + (int)(hint & 0xffu) - bias);
        turn[2] = (double)((int)((word >> 8) & 0xffu) + (int)((hint >> 8) & 0xffu) - bias);
        turn[4] = (double)((int)((word >> 16) & 0xffu) + (int)((hint >> 16) & 0xffu) - bias);
        turn[6] = (double)((int)(word >> 24) + (int)(hint >> 24) - bias);
    }
    double out[7];
#pragma unroll
    for (int k = 0; k < 7; k++) out[k] = (k == 0 || k == 2 || k == 4 || k == 6) ? fma(turn[k], kTwoPi, jv[k]) : jv[k];
    if (RSIK_RARE(sing_mask != 0)) {  // (wave-uniform: no wave of an ordinary run has such a step, and fourteen selects are 2 % of this phase)
#pragma unroll
        for (int k = 0; k < 7; k++) out[k] = dead ? __builtin_nan("") : out[k];  // (singular: needs previous_sol, phase 4 recomputes the step — flag bit 2)
    }
    if (live && sing) K.flags[t * n + i] = (uint8_t)(flag | 4);
    // one event byte per (chunk, trajectory): OR over the chunk's steps
    const unsigned long long evm = __ballot(ev && live);
    if (live && sl == 0) pipe_store(&K.chunk_event[c * n + i], (uint8_t)(((evm >> tl) & 0x0101010101010101ull) != 0 ? 1 : 0));
    // rows out: the wave's 64 rows are 8 runs (one per step) of 8 x 7 consecutive doubles; 32-bit offsets from the chunk's
    // first row (a block's joints stay below 2 GB, see rsik_control_continuous_run)
#pragma unroll
    for (int k = 0; k < 7; k++) lw[lane * 7 + k] = out[k];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    const int traj_left = (int)((n - grp * 8) < 8 ? (n - grp * 8) : 8);  // trajectories of this group that exist (<= 0 past the end)
    const int steps_left = (int)((K.T - c * kJointChunk) < kJointChunk ? (K.T - c * kJointChunk) : kJointChunk);
    const __amdgpu_buffer_rsrc_t obuf = row_buffer(K.joints + ((K.t0 + c * kJointChunk) * n + grp * 8) * 7);
    const unsigned row_bytes = (unsigned)(n * 7 * sizeof(double));
    // one store per step: its 8 x 7 doubles are consecutive in the slab (lane = 8 step + trajectory) and in memory, so lane l < 56
    // takes element l of every row — the slab offset is a constant per store, the memory offset a scalar: no address arithmetic per
    // store (seven 64-lane stores across the row boundaries cost an integer division by 56 and two compares each: ~50 instructions)
    if (lane < traj_left * 7) {
#pragma unroll
        for (int s_ = 0; s_ < kJointChunk; s_++)
            if (s_ < steps_left) st_row_f64(obuf, (unsigned)lane * 8u, (unsigned)s_ * row_bytes, lw[s_ * 56 + lane]);
    }
    __builtin_amdgcn_wave_barrier();  // (the wave's slab is free again)
}

template <bool MIXED>
__global__ __launch_bounds__(kBlock) RSIK_PHASED_OCC_ATTR void cont_joints_kernel(const ContRunArgs K) {
    RSIK_PIPE_STAMP(K, 2);
    __shared__ double lds_out[kBlock / 64][64 * 7];
    __shared__ SharedTables lds_tab;
    const int wave = threadIdx.x >> 6;
    cont_joints_chunk<MIXED>(K, lds_tab, lds_out[wave], (int64_t)blockIdx.x * (kBlock / 64) + wave, (int64_t)blockIdx.y);
}

// phase 4: eight lanes per trajectory, lane j < 7 owns joint j; sequential over the block's steps: the recurrence on
// previous_sol (allow_multiturn U:493-505, multiturn_safety_check U:535-568, continuity_check U:571-589, the emergency
// latch C:205-210, C:398-405).
// cont_chain_walk: trajectory i, lane j of its eight (gid = 8 i + j), over T steps: rows tw0 ... of the workspace arrays (chunk rows
// tw0 / 8 ...), steps t_abs0 ... of the run.  BATCH chunks' operands are fetched at once.  `last`: the run ends with these steps.
template <bool MIXED, int BATCH>
__device__ __forceinline__ void cont_chain_walk(const ContRunArgs& K, SharedTables& lds_tab, int64_t i, int j, int64_t tw0, int64_t t_abs0,
                                                int64_t T, bool last) {
    // a step's theta (behind the joints phase, which has read it: it is there)
    auto theta_at = [&](int64_t t_ws, int64_t ix) { return RSIK_WS(K, t_ws, ix); };
    const int lane = threadIdx.x & 63;
    const int gshift = lane & ~7;
    const bool live = i < K.n;
    const int64_t ii = live ? i : (K.n - 1);
    const int jj = j < 7 ? j : 6;
    const bool owner = live && j < 7;
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, MIXED ? (K.arm[ii] != 0) : false, lds_tab);
    const int64_t n = K.n;
    const int64_t tm0 = t_abs0 - K.t0;  // (load_step_m12 counts from K.t0)
    double prev = K.st[(1 + jj) * n + ii];
    bool init = K.st[8 * n + ii] != 0.0;
    bool emergency = K.st[9 * n + ii] != 0.0;
    // A latched trajectory (C:205-210: previous_sol, not reachable, the emergency state for every goal until "unfreeze") is not walked:
    // its steps of the block are filled in at once (fill_rest below) — on entry where it was latched before this block, behind the
    // chunk in which it latches otherwise — and `filled` says that nothing of it is left to judge in this block: its chunks count as
    // standing.  An ordinary run pays one OR per chunk for it.
    bool filled = false;
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
    // Operands of BATCH steps are fetched together, one batch ahead (see cont_theta_kernel).  A step is straight-line
    // code: a latched trajectory (rare) goes through the same arithmetic and only its selects differ.
    // this lane's joint of the step the wave is at: jrow[joff]; its flag byte: frow[foff] (ld_row); a step further is
    // step_stride doubles / n bytes further
    const __amdgpu_buffer_rsrc_t jbuf = row_buffer(K.joints + t_abs0 * n * 7), fbuf = row_buffer(K.flags + tw0 * n);
    const unsigned joff = (unsigned)((ii * 7 + jj) * sizeof(double)), foff = (unsigned)ii;
    const unsigned jstride = (unsigned)(n * 7 * sizeof(double)), fstride = (unsigned)n;
    int64_t t_abs = t_abs0;
    auto one = [&](double cur, int f, int64_t t, unsigned jrow) {  // step t of the block; this lane's joint of it at (jbuf, joff, jrow)
        const bool inv = (f & 16) != 0;  // the goal is not numbers: NaN joints out, the trajectory's state as it was
        if (RSIK_RARE((f & 4) != 0 && !emergency && !inv)) {  // the same byte in all 8 lanes of the trajectory
            // exact singularity in get_joints: the step is recomputed with the real previous_sol (every lane of the
            // group computes all seven joints and keeps its own)
            double pv[7];
#pragma unroll
            for (int k = 0; k < 7; k++) pv[k] = __shfl(prev, gshift + k);
            Reach r;
            Goal G;
            double m[12];
            load_step_m12(K, tm0 + t, ii, m);
            step_geometry(A, m, K.euler_roundtrip, (f & 1) == 0, r, G);
            double jv[7];
            bool sing;
            step_joints(A, K, r, G, theta_at(tw0 + t, ii), pv, jv, sing);
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
        const bool trips = cause != 0 && !emergency && !inv;
        const double result = emergency ? prev : (inv ? __builtin_nan("") : accepted);  // latched (C:205-210): previous_sol
        if (owner) st_row_f64(jbuf, joff, jrow, result);
        if (RSIK_RARE(emergency || trips) && live) {
            if (emergency) {
                if (j == 7) {
                    if (K.state) K.state[t_abs * n + i] = (uint8_t)RSIK_STATE_EMERGENCY;
                    if (K.reachable) K.reachable[t_abs * n + i] = 0;
                }
            } else if (j == 7) {
                K.st[11 * n + i] = (double)cause;
                K.st[0 * n + i] = theta_at(tw0 + t, i);  // previous_theta of the step that tripped (phase 2 ran ahead)
            } else if (disc) {
                K.st[(12 + j) * n + i] = clamped;       // the joints that failed the check
            }
        }
        prev = (emergency || trips || inv) ? prev : accepted;
        init = (emergency || inv) ? init : false;
        emergency = emergency || trips;
        t_abs += 1;
    };
    // Phase 3 has already done the quiet part of the recurrence (see cont_joints_kernel): this phase walks the block CHUNK
    // by chunk.  A chunk stands as phase 3 wrote it when its event byte is clear, the trajectory is neither latched nor at
    // its first step after a (re)initialisation, and its first step lies within the continuity threshold (less 1e-9) of
    // previous_sol — which also says that phase 3 picked the right turn; previous_sol then becomes the chunk's last row.
    // Otherwise the chunk's steps go through `one`, the reference's own sequence of operations, in place (it re-bases
    // whatever representative phase 3 wrote).  Per chunk this reads two rows of the joints and a byte instead of
    // reading and rewriting every row; the first / last rows and event bytes of BATCH chunks are fetched at once.
    const double thr_short = thr - 1e-9;
    const __amdgpu_buffer_rsrc_t ebuf = row_buffer(K.chunk_event + (tw0 / kJointChunk) * n);
    // step by step with `one` (the only copy of it), the operands of the next three steps in flight meanwhile
    auto stepwise = [&](int64_t t_blk, int64_t count) {  // the steps [t_blk, t_blk + count) of the block
        unsigned jrow = (unsigned)t_blk * jstride, frow = (unsigned)t_blk * fstride;
        t_abs = t_abs0 + t_blk;
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
    // A trajectory that is latched (C:205-210) answers previous_sol with the emergency state whatever the goal: the steps [t_from, T)
    // of the block written without reading anything — what `one` would store for them, step by step — for the trajectories of the
    // wave that are latched and not yet filled in.  Called where a trajectory can have become latched: on entry (it latched in an
    // earlier block of the run) and behind a chunk that went through `one`.  An ordinary run never gets here.
    auto fill_rest = [&](int64_t t_from) {
        if (emergency && !filled) {
            unsigned jrow = (unsigned)t_from * jstride;
            for (int64_t t = t_from; t < T; ++t) {
                if (owner) st_row_f64(jbuf, joff, jrow, prev);
                if (live && j == 7) {
                    if (K.state) K.state[(t_abs0 + t) * n + i] = (uint8_t)RSIK_STATE_EMERGENCY;
                    if (K.reachable) K.reachable[(t_abs0 + t) * n + i] = 0;
                }
                jrow += jstride;
            }
            filled = true;
        }
    };
    const int64_t n_chunks = (T + kJointChunk - 1) / kJointChunk;
    struct Operands { double first[BATCH], last[BATCH]; int ev[BATCH]; };
    auto chunk_len = [&](int64_t c) { return (T - c * kJointChunk) < kJointChunk ? (T - c * kJointChunk) : (int64_t)kJointChunk; };
    auto fetch = [&](Operands& o, int64_t c0) {
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const int64_t c = (c0 + u) < n_chunks ? (c0 + u) : (n_chunks - 1);  // (past the end: the last chunk again, skipped)
            const unsigned r_first = (unsigned)(c * kJointChunk) * jstride;
            o.first[u] = ld_row_f64(jbuf, joff, r_first);
            o.last[u] = ld_row_f64(jbuf, joff, r_first + (unsigned)(chunk_len(c) - 1) * jstride);
            o.ev[u] = ld_row_u8(ebuf, foff, (unsigned)c * fstride);
        }
    };
    // Walks the fetched chunks until one does not stand: returns its index in the batch (BATCH: all stood).
    // Phase 3 left each chunk on the turn of its first step's raw joints; `turns` (this lane's joint, almost always 0) is
    // how many whole turns that is away from previous_sol; they are added to the chunk's rows below — only the elements
    // that need it, and without waiting for them.  The limits (U:535-568): phase 3 cannot test
    // them without the turn, so they are tested here on the chunk's first step with the slack its other steps can use
    // up — they lie within (chunk - 1) continuity thresholds of it.
    const bool limited = jj == 0 || jj == 2 || jj == 6;
    const double clear_of_limit = 6 * kPi - (kJointChunk - 1) * 1.0 - 1e-6;
    // this lane's joint in the first row of the block: a quiet chunk that sits `turns` whole turns away gets them added to
    // its (up to) eight rows right here, by fire-and-forget fp64 atomic adds (v + turns * 2 pi, the one rounding a read-
    // modify-write would do; nothing reads those elements again in this launch) — no fifth phase, no hand-over to it
    // What they cost this phase is their number, not how they are issued: ~520 element updates per wave and 1000 steps, each a
    // miss in L2, ~40 ns apiece through the compute unit's memory pipeline, which the next batch's operand loads share (4096 x
    // 1000 steps, this phase alone: 45 us without them, 69 with; the fifth phase was 25 us and a hand-over).  Measured and not
    // kept: the eight lanes of a trajectory sharing the rows of the joint that turned — one atomic instruction instead of
    // eight, seven shuffles to find out which: 97 us; noted in LDS per lane and issued in one go after the walk: 80 us; noted
    // in LDS per wave (ballot + prefix count) and added with all 64 lanes, one instruction per eight notes: 66 us, no faster
    // within a pass; workgroups of 64 / 128 threads instead of 256, so that more compute units share them: 62 / 59 us, no
    // faster within a pass either.
    double* const jcol = K.joints + (t_abs0 * n + ii) * 7 + jj;
    const int64_t row_doubles = n * 7;
    // (the whole turns of a quiet chunk, added to its rows)
    auto add_turns = [&](int64_t c, double sh) {
        double* p = jcol + c * kJointChunk * row_doubles;
        const int len = (int)chunk_len(c);
        // (written as an instruction: the compiler counts the memory operations a wave has in flight — one counter for
        // loads, stores and atomics, in issue order — to wait for exactly the loads it needs, and a data-dependent number
        // of atomics among them would make it wait for everything, the operands just requested for the next batch
        // included.  Atomics it does not see only prolong a wait where they really are still in flight.)
#pragma unroll
        for (int q = 0; q < kJointChunk; q++)
            if (q < len) asm volatile("global_atomic_add_f64 %0, %1, off" : : "v"(p + q * row_doubles), "v"(sh) : "memory");
    };
    // Where the walk stops, the chunk that did not stand for the WAVE may well stand for this lane's trajectory (eight lanes): that
    // trajectory takes it like any quiet chunk, only the eventful ones go step by step — so what a trajectory gets never depends on
    // the seven others that happen to share its wave (rsik.h "Rows that are not numbers": neighbours bit for bit).
    // (judged again from the chunk's operands, fetched again: nothing of this is carried through the walk, whose registers are scarce)
    auto stands_alone = [&](int64_t c, double& sh, double& last) -> bool {
        const unsigned r_first = (unsigned)(c * kJointChunk) * jstride;
        const double first = ld_row_f64(jbuf, joff, r_first);
        last = ld_row_f64(jbuf, joff, r_first + (unsigned)(chunk_len(c) - 1) * jstride);
        const int ev = ld_row_u8(ebuf, foff, (unsigned)c * fstride);
        const double turns = -rint((first - prev) * 0.15915494309189535);
        sh = turns * kTwoPi;
        const double f2 = first + sh;
        const bool quiet = filled || (!emergency && !init && ev == 0 && (fabs(f2 - prev) <= thr_short) && (fabs(turns) <= 100.0) &&
                                      (!limited || fabs(f2) <= clear_of_limit));
        return group_or(quiet ? 0 : 1) == 0;
    };
    auto walk = [&](const Operands& o, int64_t c0) -> int {
        int stop = BATCH;
#pragma unroll
        for (int u = 0; u < BATCH; u++) {
            const double turns = -rint((o.first[u] - prev) * 0.15915494309189535);
            const double sh = turns * kTwoPi;
            const double f2 = o.first[u] + sh;
            const bool quiet = filled || (!emergency && !init && o.ev[u] == 0 && (fabs(f2 - prev) <= thr_short) && (fabs(turns) <= 100.0) &&
                                          (!limited || fabs(f2) <= clear_of_limit));
            const bool inside = c0 + u < n_chunks;
            const bool stands = !__any(!quiet) && inside;  // (wave-uniform)
            const bool taken = stop == BATCH && stands;
            if (stop == BATCH && !stands) stop = u;
            // (a filled-in trajectory keeps its previous_sol whatever its rows held when they were fetched)
            if (taken && !filled) prev = o.last[u] + sh;
            // (a chunk that goes through `one` instead is rewritten there: no turns to add)
            if (RSIK_RARE(taken && turns != 0.0 && !filled) && owner) add_turns(c0 + u, sh);
        }
        return stop;
    };
    {
        Operands oa, ob;
        int64_t c0 = 0;
        if (RSIK_RARE(__any(emergency && !filled))) fill_rest(0);
        fetch(oa, 0);
#pragma unroll 1
        while (c0 < n_chunks) {
            // the next batch's operands, always (past the end `fetch` repeats the last chunk): with a branch around it the two
            // paths meet with different numbers of loads in flight and the compiler waits for all of them before the walk —
            // the batch just requested included, a memory round trip per batch (this phase alone 52 -> 45 us without the
            // atomics, 74 -> 69 with them)
            fetch(ob, c0 + BATCH);
            const int stop = walk(oa, c0);
            if (RSIK_RARE(c0 + stop < n_chunks && stop < BATCH)) {
                // an eventful chunk: the reference's own sequence of operations for its steps, then the walk resumes behind it
                double sh, last_row;
                if (stands_alone(c0 + stop, sh, last_row)) {
                    if (!filled) prev = last_row + sh;
                    if (RSIK_RARE(sh != 0.0 && !filled) && owner) add_turns(c0 + stop, sh);
                } else {
                    stepwise((c0 + stop) * kJointChunk, chunk_len(c0 + stop));
                    if (RSIK_RARE(__any(emergency && !filled))) fill_rest((c0 + stop + 1) * kJointChunk);
                }
                c0 += stop + 1;
                if (c0 < n_chunks) fetch(oa, c0);
            } else {
                c0 += BATCH;
                oa = ob;
            }
        }
    }
    if (owner) K.st[(1 + j) * n + i] = prev;
    // (where previous_sol stands now, in whole turns: the hint of the joints kernel that takes this workspace slot next)
    if (owner && (j & 1) == 0)
        K.slot_turn_hint[4 * i + (j >> 1)] = (uint8_t)((int)((fabs(prev) < 600.0 && !K.no_turn_hint) ? rint(prev * 0.15915494309189535) : 0.0) + 128);
    if (live && j == 7) {
        K.st[8 * n + i] = init ? 1.0 : 0.0;
        K.st[9 * n + i] = emergency ? 1.0 : 0.0;
        if (last && !emergency) K.st[0 * n + i] = theta_at(tw0 + T - 1, i);  // previous_theta after the last step
    }
}

// phase 4: the chain walk over the block
template <bool MIXED>
__global__ __launch_bounds__(kChainBlock) __attribute__((amdgpu_waves_per_eu(1, 1))) void cont_chain_kernel(const ContRunArgs K) {
    RSIK_PIPE_STAMP(K, 3);
    if (K.chain_started_word != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
        __hip_atomic_store(K.chain_started_word, K.started_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // a serial phase beside throughput phases (see cont_theta_kernel), one step below the theta phase, which is the
    // critical path where the two share a SIMD (0.544 -> 0.536 ms per 4096 x 1000 pass)
    __builtin_amdgcn_s_setprio(2);
    const int64_t gid = (int64_t)blockIdx.x * kChainBlock + threadIdx.x;
    __shared__ SharedTables lds_tab;
        stage_tables<MIXED, (int)offsetof(ContRunArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC)), kChainBlock>(lds_tab, K.arms);
    cont_chain_walk<MIXED, kChainBatch>(K, lds_tab, gid >> 3, (int)(gid & 7), 0, K.t0, K.T, K.last_block != 0);
}

}  // namespace rsik
