// micro-benchmark: do the two branches of a forked hipGraph run CONCURRENTLY on gfx950 / ROCm 7.2?
//   stream s0: K0 (all CUs, short) -> fork -> A (32 workgroups x 256, spins ~20 us)   -> join -> K1
//                                          -> B (256 workgroups x 1024, records its start)
// Prints, per replay, when B's first / last workgroup started relative to A's start and A's end (s_memrealtime, 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_short(unsigned long long* t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = __builtin_amdgcn_s_memrealtime(); }
__global__ void k_spin(unsigned long long* t, int ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
}
__global__ __launch_bounds__(1024) void k_mark(unsigned long long* t) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = 1.f;
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}
int main() {
    unsigned long long *ta, *tb, *t0; hipMalloc(&ta, 64 * 8); hipMalloc(&tb, 256 * 8); hipMalloc(&t0, 64);
    hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t ef, ej; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal);
    hipLaunchKernelGGL(k_short, dim3(256), dim3(256), 0, s0, t0);
    hipEventRecord(ef, s0); hipStreamWaitEvent(s1, ef, 0);
    hipLaunchKernelGGL(k_spin, dim3(32), dim3(256), 0, s0, ta, 2000);          // A: 20 us
    hipLaunchKernelGGL(k_mark, dim3(256), dim3(1024), 32768, s1, tb);          // B
    hipEventRecord(ej, s1); hipStreamWaitEvent(s0, ej, 0);
    hipLaunchKernelGGL(k_short, dim3(256), dim3(256), 0, s0, t0 + 1);
    hipStreamEndCapture(s0, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
    unsigned long long ha[64], hb[256];
    for (int r = 0; r < 6; ++r) {
        hipGraphLaunch(ge, s0); hipStreamSynchronize(s0);
        hipMemcpy(ha, ta, sizeof ha, hipMemcpyDeviceToHost); hipMemcpy(hb, tb, sizeof hb, hipMemcpyDeviceToHost);
        unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
        for (int i = 0; i < 32; ++i) { if (ha[2 * i] < a0) a0 = ha[2 * i]; if (ha[2 * i + 1] > a1) a1 = ha[2 * i + 1]; }
        for (int i = 0; i < 256; ++i) { if (hb[i] < b0) b0 = hb[i]; if (hb[i] > b1) b1 = hb[i]; }
        printf("replay %d: A ran %.2f us; B's workgroups started %+.2f .. %+.2f us after A's start (%s)\n", r, (a1 - a0) / 100.0,
               ((long long)b0 - (long long)a0) / 100.0, ((long long)b1 - (long long)a0) / 100.0, b1 < a1 ? "CONCURRENT" : (b0 >= a1 ? "serialized after A" : "partly"));
    }
    return 0;
}
