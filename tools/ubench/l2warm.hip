// micro-benchmark: does data touched by kernel P (one dword per 128-byte line) stay close (the XCD's L2, or the MALL) for the
// kernel R that follows it in the stream?  R = 256 workgroups x 1024 threads, workgroup i reads its own slab of S bytes.
//   cold      : R alone on memory not touched for > 2 GiB of traffic
//   same-wg   : P's workgroup i touched R's slab i            (same XCD if workgroups are dealt round-robin: L2 hit expected)
//   other-xcd : P's workgroup i touched R's slab (i+1)%256    (a different XCD: only the memory-side cache can help)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k_touch(const unsigned* p, size_t slab_bytes, int shift, unsigned* sink) {
    const size_t slab = ((size_t)blockIdx.x + shift) % gridDim.x;
    const unsigned* q = p + slab * (slab_bytes / 4);
    unsigned acc = 0;
    for (size_t line = threadIdx.x; line < slab_bytes / 128; line += blockDim.x) acc ^= q[line * 32];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(1024) void k_read(const v4u* p, size_t slab_bytes, unsigned* sink, unsigned long long* ticks) {
    const v4u* q = p + (size_t)blockIdx.x * (slab_bytes / 16);
    const size_t n16 = slab_bytes / 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < n16; i += 4 * blockDim.x) {
        v4u v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const size_t k = i + u * blockDim.x; v[u] = k < n16 ? __builtin_nontemporal_load(q + k) : v4u{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ticks[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

int main() {
    const size_t total = 3ull << 30;
    char* buf; unsigned* sink; unsigned long long* ticks;
    hipMalloc(&buf, total); hipMalloc(&sink, 64); hipMalloc(&ticks, 256 * 8);
    hipMemset(buf, 1, total);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long h[256];
    size_t cursor = 0;
    for (size_t slab : {65536ul, 131072ul, 262144ul}) {
        const size_t region = slab * 256;
        for (int mode = 0; mode < 3; ++mode) {
            double us_sum = 0, tick_sum = 0; const int reps = 10;
            for (int r = 0; r < reps; ++r) {
                cursor = (cursor + region) % (total - region); cursor = cursor / 4096 * 4096;
                char* p = buf + cursor;
                if (mode) hipLaunchKernelGGL(k_touch, dim3(256), dim3(1024), 0, st, (const unsigned*)p, slab, mode == 2 ? 1 : 0, sink);
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, st, (const v4u*)p, slab, sink, ticks);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); us_sum += ms * 1000;
                hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
                double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i]; tick_sum += m / 256;
            }
            printf("slab %4zu KiB/workgroup  %-9s  kernel %6.2f us   in-kernel read %7.0f ticks (mean over workgroups)\n", slab >> 10,
                   mode == 0 ? "cold" : mode == 1 ? "same-wg" : "other-xcd", us_sum / reps, tick_sum / reps);
        }
    }
    return 0;
}
