// Issue-rate probe for a LONE wave on a gfx950 SIMD (the situation of the serial phases of the continuous pipeline):
// cycles per instruction for dependent / independent chains of the instruction kinds those phases are made of.
// Build: hipcc -O2 --offload-arch=gfx950 scripts/probes/issue_probe.hip -o build/issue_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ void probe(double* out, uint64_t* cycles, int iters) {
    double a = out[threadIdx.x], b = a + 1.0, c = a + 2.0, d = a + 3.0, k = 1.0000001, m = 0.5;
    unsigned u = threadIdx.x, v = u + 1, w = u + 2, x = u + 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {  // dependent v_fma_f64
            asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a) : "v"(k), "v"(m));
        } else if constexpr (KIND == 1) {  // 4 independent v_fma_f64 chains
            asm volatile(REP16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k), "v"(m));
        } else if constexpr (KIND == 2) {  // dependent v_add_f64
            asm volatile(REP64("v_add_f64 %0, %0, %1\n") : "+v"(a) : "v"(m));
        } else if constexpr (KIND == 3) {  // dependent v_add_u32
            asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(u) : "v"(v));
        } else if constexpr (KIND == 4) {  // 4 independent v_add_u32
            asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(u), "+v"(v), "+v"(w), "+v"(x) : "v"(threadIdx.x));
        } else if constexpr (KIND == 5) {  // v_cmp_lt_f64 -> v_cndmask_b32 x2 (a 64-bit select), dependent through a
            asm volatile(REP16("v_cmp_lt_f64 vcc, %0, %3\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_f64 %0, %0, %3\n")
                         : "+v"(a), "+v"(u), "+v"(w) : "v"(m), "v"(v) : "vcc");
        } else if constexpr (KIND == 6) {  // dependent s_add_u32
            unsigned s = iters;
            asm volatile(REP64("s_add_u32 %0, %0, 1\n") : "+s"(s) : : "scc");
            u += s;
        } else if constexpr (KIND == 7) {  // alternating v_add_f64 / s_add_u32 (independent of each other)
            unsigned s = iters;
            asm volatile(REP16("v_add_f64 %0, %0, %2\n s_add_u32 %1, %1, 1\n v_add_f64 %0, %0, %2\n s_add_u32 %1, %1, 1\n")
                         : "+v"(a), "+s"(s) : "v"(m) : "scc");
            u += s;
        } else if constexpr (KIND == 8) {  // dependent v_floor_f64 + v_fma (the modulo's core)
            asm volatile(REP16("v_mul_f64 %1, %0, %2\n v_floor_f64 %1, %1\n v_fma_f64 %0, -%1, %3, %0\n v_add_f64 %0, %0, %2\n")
                         : "+v"(a), "+v"(b) : "v"(m), "v"(k));
        } else if constexpr (KIND == 9) {  // v_cmp to an SGPR pair + s_or_b64 accumulate + independent v_add_f64
            uint64_t acc = 0, tmp;
            asm volatile(REP16("v_cmp_gt_f64 %2, %0, %3\n s_or_b64 %1, %1, %2\n v_add_f64 %0, %0, %3\n v_add_f64 %0, %0, %3\n")
                         : "+v"(a), "+s"(acc), "=&s"(tmp) : "v"(m) : "scc");
            u += (unsigned)acc;
        } else if constexpr (KIND == 10) {  // dependent v_max_f64 / v_min_f64 (clamp)
            asm volatile(REP16("v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2\n v_max_f64 %0, %0, %1\n v_min_f64 %0, %0, %2\n") : "+v"(a) : "v"(m), "v"(k));
        } else if constexpr (KIND == 11) {  // v_mov_b32 pairs (register copies)
            asm volatile(REP16("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %2, %0\n v_mov_b32 %3, %1\n") : "+v"(u), "+v"(v), "+v"(w), "+v"(x));
        } else if constexpr (KIND == 12) {  // v_bfi_b32 (copysign) + v_add_f64 dependent
            asm volatile(REP16("v_bfi_b32 %1, %2, %1, %4\n v_add_f64 %0, %0, %3\n v_bfi_b32 %1, %2, %1, %4\n v_add_f64 %0, %0, %3\n")
                         : "+v"(a), "+v"(u) : "v"(0x7fffffffu), "v"(m), "v"(v));
        } else if constexpr (KIND == 13) {  // v_cmp -> SGPR pair (not vcc) -> v_cndmask reading it, + v_add_f64 (no scalar ALU)
            uint64_t tmp;
            asm volatile(REP16("v_cmp_lt_f64 %3, %0, %4\n v_cndmask_b32 %1, %1, %5, %3\n v_cndmask_b32 %2, %2, %5, %3\n v_add_f64 %0, %0, %4\n")
                         : "+v"(a), "+v"(u), "+v"(w), "=&s"(tmp) : "v"(m), "v"(v));
        } else if constexpr (KIND == 14) {  // v_cmp -> vcc -> s_and_b64 with a constant mask -> v_cndmask (VALU -> SALU -> VALU)
            uint64_t msk = ~0ull;
            asm volatile(REP16("v_cmp_lt_f64 vcc, %0, %3\n s_and_b64 vcc, vcc, %5\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_f64 %0, %0, %3\n")
                         : "+v"(a), "+v"(u), "+v"(w) : "v"(m), "v"(v), "s"(msk) : "vcc", "scc");
        } else if constexpr (KIND == 15) {  // v_cmp -> vcc -> s_cbranch_vccnz (never taken) + 2 v_add_f64
            asm volatile(REP16("v_cmp_gt_f64 vcc, %0, %1\n s_cbranch_vccnz 1f\n 1:\n v_add_f64 %0, %0, %2\n v_add_f64 %0, %0, %2\n")
                         : "+v"(a) : "v"(1e300), "v"(m) : "vcc");
        } else if constexpr (KIND == 16) {  // event accumulation on the vector side: v_cmp -> vcc -> v_cndmask_b32 ev, ev, 1
            asm volatile(REP16("v_cmp_gt_f64 vcc, %0, %2\n v_cndmask_b32 %1, %1, %3, vcc\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n")
                         : "+v"(a), "+v"(u) : "v"(1e300), "v"(v), "v"(m) : "vcc");
        } else if constexpr (KIND == 17) {  // v_readfirstlane -> s_add (VALU -> SALU through a lane read)
            unsigned sacc = 0, st;
            asm volatile(REP16("v_readfirstlane_b32 %2, %1\n s_add_u32 %3, %3, %2\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n")
                         : "+v"(a), "+v"(u), "=&s"(st), "+s"(sacc) : "v"(m) : "scc");
            v += sacc;
        } else if constexpr (KIND == 18) {  // the 8-lane group OR of the chain phase: three DPP v_or
            asm volatile(REP16("v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_or_b32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %0, %1\n")
                         : "+v"(u) : "v"(v));
        } else if constexpr (KIND == 19) {  // quarter-rate: v_rcp_f64 dependent
            asm volatile(REP64("v_rcp_f64 %0, %0\n") : "+v"(a));
        } else if constexpr (KIND == 20) {  // v_cmp_class / v_cmp with |abs| modifier then v_cndmask: same as 5 (check)
            asm volatile(REP16("v_cmp_lt_f64 vcc, |%0|, %3\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_f64 vcc, %0, %3\n v_cndmask_b32 %2, %2, %4, vcc\n")
                         : "+v"(a), "+v"(u), "+v"(w) : "v"(m), "v"(v) : "vcc");
        } else if constexpr (KIND == 21) {  // two v_cmp to SGPR pairs, s_or_b64 of them, v_cndmask on the result (the mask algebra as compiled today)
            uint64_t t1, t2;
            asm volatile(REP16("v_cmp_lt_f64 %3, %0, %5\n v_cmp_gt_f64 %4, %0, %6\n s_or_b64 %3, %3, %4\n v_cndmask_b32 %1, %1, %7, %3\n")
                         : "+v"(a), "+v"(u), "+v"(w), "=&s"(t1), "=&s"(t2) : "v"(m), "v"(k), "v"(v) : "scc");
        } else if constexpr (KIND == 22) {  // the same decision with chained v_cndmask (no scalar ALU): 2 v_cmp + 2 v_cndmask
            asm volatile(REP16("v_cmp_lt_f64 vcc, %0, %3\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_gt_f64 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n")
                         : "+v"(a), "+v"(u), "+v"(w) : "v"(m), "v"(k), "v"(v) : "vcc");
        } else if constexpr (KIND == 23) {  // s_nop 0 between dependent v_add_f64 (does an idle slot cost a full turn?)
            asm volatile(REP16("v_add_f64 %0, %0, %1\n s_nop 0\n v_add_f64 %0, %0, %1\n s_nop 0\n") : "+v"(a) : "v"(m));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[threadIdx.x + 64 * blockIdx.x] = a + b + c + d + (double)(u + v + w + x);
}

template <int KIND>
static void run(const char* name, int per_iter, int blocks, int waves) {
    double* out;
    uint64_t* cyc;
    hipMalloc(&out, sizeof(double) * 64 * 8 * 4096);
    hipMalloc(&cyc, sizeof(uint64_t) * 4096);
    hipMemset(out, 0, sizeof(double) * 64 * 8 * 4096);
    const int iters = 256;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(uint64_t) * blocks, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : h) mean += (double)c;
    mean /= blocks;
    std::printf("%-58s blocks %4d waves/block %d: %7.2f cycles (s_memtime) per instruction\n", name, blocks, waves, mean / ((double)iters * per_iter));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int cfg = 0; cfg < 3; cfg += 2) {
        const int blocks = cfg == 0 ? 1 : 256, waves = cfg == 2 ? 8 : 1;  // lone wave; one wave per CU; two waves per SIMD
        run<0>("v_fma_f64 dependent", 64, blocks, waves);
        run<1>("v_fma_f64 four independent chains", 64, blocks, waves);
        run<2>("v_add_f64 dependent", 64, blocks, waves);
        run<3>("v_add_u32 dependent", 64, blocks, waves);
        run<4>("v_add_u32 four independent", 64, blocks, waves);
        run<5>("v_cmp_lt_f64 + 2 v_cndmask + v_add_f64", 64, blocks, waves);
        run<6>("s_add_u32 dependent", 64, blocks, waves);
        run<7>("v_add_f64 / s_add_u32 alternating", 64, blocks, waves);
        run<8>("v_mul, v_floor, v_fma, v_add f64 dependent", 64, blocks, waves);
        run<9>("v_cmp->sgpr, s_or_b64, 2 v_add_f64", 64, blocks, waves);
        run<10>("v_max_f64 / v_min_f64 dependent", 64, blocks, waves);
        run<11>("v_mov_b32", 64, blocks, waves);
        run<12>("v_bfi_b32 + v_add_f64", 64, blocks, waves);
        run<13>("v_cmp->sgpr, 2 v_cndmask(sgpr), v_add_f64", 64, blocks, waves);
        run<14>("v_cmp->vcc, s_and_b64 vcc, v_cndmask, v_add_f64", 64, blocks, waves);
        run<15>("v_cmp->vcc, s_cbranch_vccnz (not taken), 2 v_add_f64", 64, blocks, waves);
        run<16>("v_cmp->vcc, v_cndmask ev, 2 v_add_f64", 64, blocks, waves);
        run<17>("v_readfirstlane, s_add_u32, 2 v_add_f64", 64, blocks, waves);
        run<18>("3 v_or_b32_dpp + v_add_u32", 64, blocks, waves);
        run<19>("v_rcp_f64 dependent", 64, blocks, waves);
        run<20>("v_cmp |a|, v_cndmask, v_cmp, v_cndmask", 64, blocks, waves);
        run<21>("2 v_cmp->sgpr, s_or_b64, v_cndmask(sgpr)", 64, blocks, waves);
        run<22>("2 x (v_cmp->vcc, v_cndmask)", 64, blocks, waves);
        run<23>("v_add_f64, s_nop 0", 64, blocks, waves);
    }
    return 0;
}
