// anyorder_probe.hip — does hipExtLaunchKernel's hipExtAnyOrderLaunch let a kernel overlap the one before it on the SAME stream on
// gfx950 (hip_ext.h says "not supported on AMD GFX9xx boards" of the module-launch variant)?  Two one-workgroup kernels that spin 200 us
// each, back to back on one stream: ~400 us if they serialise, ~200 us if the second one is let in.  Then a third launched normally
// behind an any-order one: it must wait for BOTH.  Build: hipcc -O2 --offload-arch=gfx950 anyorder_probe.hip -o build/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

__global__ void spin_kernel(unsigned long long* t, int slot, int spin_us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) { t[2 * slot] = t0; t[2 * slot + 1] = __builtin_amdgcn_s_memrealtime(); }
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* t;
    CK(hipMalloc(&t, 64 * sizeof(unsigned long long)));
    unsigned long long h[64];
    for (int rep = 0; rep < 3; rep++) {
        for (int mode = 0; mode < 2; mode++) {
            CK(hipMemset(t, 0, 64 * sizeof(unsigned long long)));
            CK(hipDeviceSynchronize());
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, t, 0, 200);
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, t, 1, 200);
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, t, 2, 200);
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, t, 3, 50);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
            const double base = (double)h[0];
            printf("%s: ", mode ? "kernels 1, 2 any-order" : "all in order          ");
            for (int k = 0; k < 4; k++) printf(" k%d %7.1f -> %7.1f us", k, (h[2 * k] - base) / 100.0, (h[2 * k + 1] - base) / 100.0);
            printf("\n");
        }
    }
    return 0;
}
