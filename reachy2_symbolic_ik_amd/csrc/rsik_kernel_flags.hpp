// rsik_kernel_flags.hpp — rsik_control_continuous_run as chip-filling kernels the hardware schedules, plus two persistent
// kernels for the sequential phases, tied together by flags in device memory instead of events between launches
// (one translation unit: included by rsik_lib.hip behind rsik_kernel_fused.hpp, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// What was measured on the way here.  The phased pipeline's chip-filling kernels keep the vector units
// ~88 % busy, but its sequential kernels (lone waves with 220-280 registers) cannot START while a chip-filling kernel holds
// every SIMD's register file: theta(b + 1) sat behind joints(b) whatever the streams and events said, a third of a pass.
// The single self-scheduling launch (rsik_kernel_fused.hpp) has no such waits, but its worker waves — items claimed from a
// queue, dependencies polled, completions signalled behind drained stores, four waves per SIMD — keep the vector units a
// third busy; per-workgroup polls and drains cost an ordinary kernel a quarter to a half of its throughput, too.
// This form keeps what works of each:
//   * prepare: ONE ordinary kernel over the whole run (a workgroup per step and tile of trajectories, dispatched in step
//     order).  It hands each step's goal to the theta waves as a tagged pair (cont_prepare_step, PAIRS): no counter, no
//     drain, the workgroup ends like the phased kernel's.
//   * theta: persistent — the theta workgroups of the single launch (walker / loader / writer around an LDS ring, a compute
//     unit each), started first and resident for the whole run, so they never have to get in again.  They follow the prepare
//     kernel by the tags, and hand each theta on as a tagged pair with the step's flags and state code in the tag.
//   * joints: an ordinary kernel per BLOCK of steps, on two streams in turn, each launch held by its stream
//     (hipStreamWaitValue32) until every theta wave has counted itself through the block — a word in device memory, 1-4 us
//     from the last count to the kernel's first wave — and validating what it loads by the tags (a stale pair: poll a hint,
//     load again).  No per-workgroup poll, no drain: it writes rows, flags and states like the phased kernel.
//   * chain: an ordinary, small kernel per couple of blocks on a stream of its own, each held until the words that its
//     blocks' joints launches' streams write behind them (hipStreamWriteValue32) say they have completed: kernel
//     boundaries, so the rows are in memory without anybody waiting for stores.
// ------------------------------------------------------------------------------------------

// prepare, all steps of the run: blockIdx.y = step, blockIdx.x = tile of kBlock trajectories
template <bool MIXED, bool PLANE>
__global__ __launch_bounds__(kBlock) void flags_prepare_kernel(const FusedArgs F) {
    const ContRunArgs& K = F.R;
    RSIK_PIPE_STAMP_AT(K, blockIdx.y / F.S, 0);
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t t = blockIdx.y;
    const bool live = i < K.n;
    const int64_t ii = live ? i : (K.n - 1);
    double m[12];  // loads first: their latency overlaps the table staging
    const double* src = K.m12_steps + t * 12 * K.n + ii;
#pragma unroll
    for (int k = 0; k < 12; k++) m[k] = src[k * K.n];
    const bool lane_isl = MIXED ? (K.arm[ii] != 0) : false;
    __shared__ SharedTables lds_tab;
    stage_tables<MIXED, (int)offsetof(ContRunArgs, arms) + (MIXED ? 0 : (int)sizeof(ArmC))>(lds_tab, K.arms);
    const Acc<MIXED> A = make_acc<MIXED>(K.arms, lane_isl, lds_tab);
    const int slot = MIXED ? (A.isl ? 1 : 0) : 0;
    cont_prepare_step<MIXED, PLANE, true, true>(K, A, slot, m, t, t, i, live);
    // a count that tells a theta loader that found an old tag how far the block's steps have got: issued behind the stores,
    // not waiting for them (a hint: the tags decide)
    constexpr int kGroups = kBlock / 64;
    if ((threadIdx.x & 63) == 0) {
        const int g = (int)blockIdx.x * kGroups + (int)(threadIdx.x >> 6);
        if (g < F.G) sync_add(F.sync + kSyncArrays + (size_t)(t / F.S) * F.G + g, 1u);
    }
}

// joints, one block of steps: blockIdx.y = chunk of the block (8 steps), blockIdx.x = tile of 32 trajectories (a wave: 8)
template <bool MIXED>
__global__ __launch_bounds__(kBlock) void flags_joints_kernel(const FusedArgs F, const int64_t first_chunk) {
    RSIK_PIPE_STAMP_AT(F.R, (first_chunk * kJointChunk) / F.S, 2);
    __shared__ double lds_out[kBlock / 64][64 * 7];
    __shared__ SharedTables lds_tab;
    const int wave = threadIdx.x >> 6;
    // (thetas, flags and state codes as tagged pairs; rows, flag / state / reachable bytes out as ordinary stores: the chain
    // kernel of the block starts behind this launch's end)
    cont_joints_chunk<MIXED, false, true, true>(F.R, lds_tab, lds_out[wave], (int64_t)blockIdx.x * (kBlock / 64) + wave, first_chunk + blockIdx.y);
}

// chain, a few blocks of steps: eight lanes per trajectory; at most 128 registers, so that its waves find room beside the joints
// kernels of the blocks after it.  (Persistent chain waves polling for their blocks were measured instead: 512 waves that sit on
// registers and poll one word cost the prepare kernel 130 us.)
template <bool MIXED>
__global__ __launch_bounds__(kChainBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) void flags_chain_kernel(const FusedArgs F, const int64_t first_step,
                                                                                                          const int64_t steps, const int last) {
    RSIK_PIPE_STAMP_AT(F.R, first_step / F.S, 3);
    const int64_t gid = (int64_t)blockIdx.x * kChainBlock + threadIdx.x;
    __shared__ SharedTables lds_tab;
    stage_tables<MIXED, 0, kChainBlock>(lds_tab, F.R.arms);
    cont_chain_walk<MIXED, false, kFusedChainBatch, true>(F.R, lds_tab, gid >> 3, (int)(gid & 7), first_step, first_step, steps, last != 0);
}

// the theta workgroups as a kernel: four groups per workgroup, twelve waves (walker / loader / writer per group); enough
// registers that nothing else fits beside them: the walkers have their SIMDs to themselves
template <bool MIXED>
__global__ __launch_bounds__(768) void flags_theta_kernel(const FusedArgs F) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_big[sizeof(ThetaRing)];
    const ThetaRingPtr ring = (ThetaRingPtr)lds_big;
    const FusedArgsK fk = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    if (threadIdx.x < 13) (&ring->filled[0])[threadIdx.x] = 0u;  // filled, done, written, abort
    if (threadIdx.x == 0) sync_add(F.sync + kSyncAlive, 1u);  // (this workgroup runs: the host lets the chip-filling kernels go once all do)
    __syncthreads();
    asm volatile("v_mov_b32 v167, 0" ::: "v167");  // (168 registers x 12 waves: the compute unit's register file, nothing else fits)
    fused_theta_wave<MIXED, true>(fk, (int)blockIdx.x, (int)(threadIdx.x >> 6), ring);
}

}  // namespace rsik
