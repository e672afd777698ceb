// rsik_kernel_flags.hpp — rsik_control_continuous_run as chip-filling kernels the hardware schedules, plus two persistent
// kernels for the sequential phases, tied together by flags in device memory instead of events between launches
// (one translation unit: included by rsik_lib.hip behind rsik_kernel_fused.hpp, inside nothing)
#pragma once

namespace rsik {

// ------------------------------------------------------------------------------------------
// What was measured on the way here: the phased pipeline's kernels keep the vector units ~88 % busy,
// but every dependency between two launches on different streams costs 15-55 us, a third of a pass; the single
// self-scheduling launch (rsik_kernel_fused.hpp) has no such hand-overs, but its worker waves — items claimed from a queue,
// dependencies polled, completions signalled behind drained stores, four waves per SIMD — keep the vector units a third
// busy.  This form keeps what works of each:
//   * prepare and joints are ordinary kernels over the WHOLE run (one workgroup per step / chunk and tile of trajectories,
//     dispatched by the hardware in step order, six waves per SIMD), back to back on the caller's stream;
//   * the two recurrences are persistent: the theta workgroups of the single launch (walker / loader / writer around an LDS
//     ring, a compute unit each) as a kernel of their own, and one chain wave per eight trajectories; both follow the
//     chip-filling kernels block by block (S steps) through counters in device memory — pdone[b][g]: prepare workgroups of
//     block b that have written group g's goals; tprog[g]: blocks whose thetas are in memory; jdone[b][g]: joints
//     workgroups of block b done with group g — with the coherent (written-through / L2-bypassing) accesses of the shared
//     bodies for everything that crosses from one kernel to another while both run.
// Stream-level dependencies left: the fork of the two persistent kernels at the start and their join at the end.  The joints
// kernel starts when the prepare kernel has finished (stream order), by which time the theta waves — fed as the prepare
// kernel goes — are blocks ahead; a joints wave whose thetas are not there yet polls tprog (bounded, like every wait).
// ------------------------------------------------------------------------------------------

// prepare, all steps of the run: blockIdx.y = step, blockIdx.x = tile of kBlock trajectories
template <bool MIXED, bool PLANE>
__global__ __launch_bounds__(kBlock) void flags_prepare_kernel(const FusedArgs F) {
    const ContRunArgs& K = F.R;
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
    // (goals as tagged pairs: nothing to wait for — the workgroup ends like the phased kernel's, but for a count that tells the
    // theta loaders how far the steps of the block have got: issued behind the stores, not waiting for them)
    cont_prepare_step<MIXED, PLANE, true, true>(K, A, slot, m, t, t, i, live);
    constexpr int kGroups = kBlock / 64;
    if ((threadIdx.x & 63) == 0) {
        const int g = (int)blockIdx.x * kGroups + (int)(threadIdx.x >> 6);
        if (g < F.G) sync_add(F.sync + kSyncArrays + (size_t)(t / F.S) * F.G + g, 1u);
    }
}

// joints, all chunks of the run: blockIdx.y = chunk (8 steps), blockIdx.x = tile of 32 trajectories (a wave: 8 of them)
template <bool MIXED>
__global__ __launch_bounds__(kBlock) void flags_joints_kernel(const FusedArgs F) {
    const ContRunArgs& K = F.R;
    __shared__ double lds_out[kBlock / 64][64 * 7];
    __shared__ SharedTables lds_tab;
    const int wave = threadIdx.x >> 6;
    const int64_t c = blockIdx.y;
    const int b = (int)((c * kJointChunk) / F.S);
    const int g = (int)(blockIdx.x >> 1);  // 32 trajectories per workgroup: two workgroups per group of 64
    unsigned* const jdone = F.sync + kSyncArrays + (size_t)F.B * F.G;
    // (thetas as tagged pairs: the chunk's loads wait for the theta wave only if it really is behind)
    cont_joints_chunk<MIXED, true, true, true>(K, lds_tab, lds_out[wave], (int64_t)blockIdx.x * (kBlock / 64) + wave, c);
    // the chain wave reads and updates the rows in place: they must be in memory before it is told
    stores_done();
    __syncthreads();
    if (threadIdx.x == 0) sync_add(jdone + (size_t)b * F.G + g, 1u);
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

// the chain waves as a kernel: one wave (eight trajectories, eight lanes each) per workgroup
// (at most 128 registers: a chain wave shares its SIMD with the chip-filling kernels' waves — with the 266 the compiler would
// take, half of the chip's SIMDs held two of those instead of six: the prepare kernel 135 -> 215 us, the joints kernel 220 -> 405)
template <bool MIXED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void flags_chain_kernel(const FusedArgs F) {
    __shared__ SharedTables lds_tab;
    const FusedArgsK fk = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    stage_tables<MIXED, 0, 64>(lds_tab, F.R.arms);
    __builtin_amdgcn_s_setprio(2);
    (void)fused_chain_wave<MIXED, true>(fk, (int)blockIdx.x, (LdsTabPtr)&lds_tab);
}

}  // namespace rsik
