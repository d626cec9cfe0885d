// micro-benchmark: what a plain streaming read reaches on gfx950 at the sizes of the decode GEMVs (one launch reads `bytes` once, every
// launch a different region of a 3 GiB buffer so that neither L2 nor the 256 MiB MALL helps), for the launch shapes the GEMV could use.
// Gives the practical ceiling to hold roofline.frac against: usage  stream [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(1024) void k_read(const v4u* __restrict__ p, size_t n16, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { v4u v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the GEMV's own order: a workgroup owns a contiguous slab (its rows), a wave reads 1 KiB per instruction
template <int U>
__global__ __launch_bounds__(1024) void k_read_slab(const v4u* __restrict__ p, size_t n16, unsigned* sink) {
    const size_t per = n16 / gridDim.x;
    const v4u* q = p + per * blockIdx.x;
    unsigned acc = 0;
    size_t i = threadIdx.x;
    for (; i + (U - 1) * blockDim.x < per; i += U * blockDim.x) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(q + i + u * blockDim.x);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const size_t total = 3ull << 30;
    v4u* buf; unsigned* sink;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, total);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double sizes_mb[] = {16.8, 45.1, 50.3, 90.2, 131.1, 1024};
    auto run = [&](const char* name, double mb, auto launch) {
        const size_t bytes = (size_t)(mb * 1e6) / 4096 * 4096, n16 = bytes / 16;
        const size_t slots = total / bytes;
        size_t slot = 0;
        auto next = [&] { const v4u* p = buf + (slot % slots) * n16; ++slot; return p; };
        for (int w = 0; w < 3; ++w) launch(next(), n16);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) launch(next(), n16);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000 / reps;
        printf("%-34s %7.1f MB  %7.2f us  %5.2f TB/s\n", name, mb, us, bytes / us * 1e-6);
    };
    for (double mb : sizes_mb) {
        run("grid-stride 256x1024 U4", mb, [&](const v4u* p, size_t n) { hipLaunchKernelGGL(k_read<4>, dim3(256), dim3(1024), 0, st, p, n, sink); });
        run("grid-stride 256x1024 U8", mb, [&](const v4u* p, size_t n) { hipLaunchKernelGGL(k_read<8>, dim3(256), dim3(1024), 0, st, p, n, sink); });
        run("grid-stride 512x1024 U4", mb, [&](const v4u* p, size_t n) { hipLaunchKernelGGL(k_read<4>, dim3(512), dim3(1024), 0, st, p, n, sink); });
        run("grid-stride 2048x256 U4", mb, [&](const v4u* p, size_t n) { hipLaunchKernelGGL(k_read<4>, dim3(2048), dim3(256), 0, st, p, n, sink); });
        run("grid-stride 8192x256 U4", mb, [&](const v4u* p, size_t n) { hipLaunchKernelGGL(k_read<4>, dim3(8192), dim3(256), 0, st, p, n, sink); });
        run("slab/WG     256x1024 U8", mb, [&](const v4u* p, size_t n) { hipLaunchKernelGGL(k_read_slab<8>, dim3(256), dim3(1024), 0, st, p, n, sink); });
        run("slab/WG     1024x1024 U4", mb, [&](const v4u* p, size_t n) { hipLaunchKernelGGL(k_read_slab<4>, dim3(1024), dim3(1024), 0, st, p, n, sink); });
        printf("\n");
    }
    return 0;
}
