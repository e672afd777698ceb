// edge_probe.hip — what a dependency between two kernels on DIFFERENT streams costs on the device, with the host out of the way
// (everything is enqueued behind a blocker kernel first): the child's first instruction minus the parent's last, from the 100 MHz
// clock, for three ways of tying the streams: a hipEvent, hipStreamWriteValue32 -> hipStreamWaitValue32, and a flag the parent
// kernel itself writes (hipStreamWaitValue32 on it).  Build: hipcc -O2 --offload-arch=gfx950 edge_probe.hip -o build/edge_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; }    \
    } while (0)

__global__ void stamp_kernel(unsigned long long* t, int slot, int spin_us, unsigned* flag, unsigned flag_value) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        t[2 * slot] = t0;
        t[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
        if (flag) __hip_atomic_store(flag, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int main() {
    const int links = 16, reps = 5;
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long* t;
    unsigned* flags;
    CK(hipMalloc(&t, sizeof(unsigned long long) * 2 * (links + 2)));
    CK(hipMalloc(&flags, sizeof(unsigned) * (links + 2)));
    std::vector<hipEvent_t> ev(links + 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::vector<unsigned long long> h(2 * (links + 2));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    for (int mode = 0; mode < 3; mode++) {
        double sum = 0, worst = 0, best = 1e9;
        for (int r = 0; r < reps; r++) {
            CK(hipMemset(flags, 0, sizeof(unsigned) * (links + 2)));
            CK(hipDeviceSynchronize());
            // a blocker on stream a: 2 ms, so that the whole chain is enqueued before anything of it runs
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, a, t, links + 1, 2000, (unsigned*)nullptr, 0u);
            CK(hipEventRecord(ev[links + 1], a));
            CK(hipStreamWaitEvent(b, ev[links + 1], 0));
            for (int k = 0; k < links; k++) {  // kernel k on stream (k even: a, odd: b), 20 us each, waits for kernel k - 1
                hipStream_t s = (k % 2 == 0) ? a : b, o = (k % 2 == 0) ? b : a;
                (void)o;
                if (k > 0) {
                    if (mode == 0) CK(hipStreamWaitEvent(s, ev[k - 1], 0));
                    else CK(hipStreamWaitValue32(s, flags + (k - 1), 1u, hipStreamWaitValueGte, 0xffffffffu));
                }
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, t, k, 20, mode == 2 ? flags + k : (unsigned*)nullptr, 1u);
                if (mode == 0) CK(hipEventRecord(ev[k], s));
                else if (mode == 1) CK(hipStreamWriteValue32(s, flags + k, 1u, 0));
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), t, sizeof(unsigned long long) * 2 * (links + 2), hipMemcpyDeviceToHost));
            for (int k = 1; k < links; k++) {
                const double gap = (double)(h[2 * k] - h[2 * (k - 1) + 1]) / 100.0;
                sum += gap;
                if (gap > worst) worst = gap;
                if (gap < best) best = gap;
            }
        }
        static const char* names[3] = {"hipEventRecord -> hipStreamWaitEvent", "hipStreamWriteValue32 -> hipStreamWaitValue32",
                                       "flag written by the parent kernel -> hipStreamWaitValue32"};
        printf("%-62s: parent's end -> child's start %6.2f us mean (%.2f ... %.2f)\n", names[mode], sum / (reps * (links - 1)), best, worst);
    }
    return 0;
}
