// micro-benchmark: the engine's dot loop (eng_run of flm_engine.h) on data that already sits in LDS -- what ONE consumer wave sustains, alone and next to
// 3 / 7 / 11 others, and what each part of a piece costs (variants drop the chain, the dots or the LDS reads).   usage: engloop [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "flm_kernels.h"
using namespace flm;

template <bool TWO, int VAR>
__device__ __forceinline__ void run_var(const char* wq, const char* sq, const char* xa, const char* xsa, const int n, float& acc, float& acc3) {
    if constexpr (VAR == 0) eng_run<TWO>(wq, sq, xa, xsa, n, acc, acc3);
    else {
        // the same loop with parts removed: VAR 1 no chain, 2 no dots (chain on constants), 3 neither (LDS reads only)
        for (int i = 0; i < n; ++i) {
            const v4i xv = *reinterpret_cast<const v4i*>(xa + i * 256);
            const float4 sx = *reinterpret_cast<const float4*>(xsa + i * 16);
            const v4i w1 = *reinterpret_cast<const v4i*>(wq + (TWO ? 2 * i : i) * 1024);
            const float4 s1 = *reinterpret_cast<const float4*>(sq + (TWO ? 2 * i : i) * 64);
            float f1 = __int_as_float(w1.x ^ xv.x);
            if (VAR == 1) f1 = (float)quad_sum(dot16_i8(w1, xv, 0));
            const float4 g = make_float4(__fmul_rn(s1.x, sx.x), __fmul_rn(s1.y, sx.y), __fmul_rn(s1.z, sx.z), __fmul_rn(s1.w, sx.w));
            if (VAR == 2) eng_chain4(acc, f1, g); else acc += f1 + g.x + g.y + g.z + g.w;
            if constexpr (TWO) {
                const v4i w3 = *reinterpret_cast<const v4i*>(wq + (2 * i + 1) * 1024);
                const float4 s3 = *reinterpret_cast<const float4*>(sq + (2 * i + 1) * 64);
                float f3 = __int_as_float(w3.x ^ xv.y);
                if (VAR == 1) f3 = (float)quad_sum(dot16_i8(w3, xv, 0));
                const float4 g3 = make_float4(__fmul_rn(s3.x, sx.x), __fmul_rn(s3.y, sx.y), __fmul_rn(s3.z, sx.z), __fmul_rn(s3.w, sx.w));
                if (VAR == 2) eng_chain4(acc3, f3, g3); else acc3 += f3 + g3.x + g3.y + g3.z + g3.w;
            }
        }
    }
}

template <bool TWO, int VAR>
__global__ void __launch_bounds__(1024) k_loop(float* out, unsigned long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 150 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* xq = lds + 14 * kEngSlotBytes + 256;
    const char* xs = xq + 11008;
    float acc = 0.f, acc3 = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const char* sl = lds + ((it + wave) % 14) * kEngSlotBytes;
        run_var<TWO, VAR>(sl + lane * 16, sl + kEngSlotW + (lane >> 4) * 16, xq + (it % 5) * 2048 + (lane & 15) * 16, xs + (it % 5) * 128, TWO ? 4 : 8, acc, acc3);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) ticks[blockIdx.x * 16 + wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc3;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 256 * 16 * 8);
    auto run = [&](const char* name, auto kern, int waves) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipMemset(ticks, 0, 256 * 16 * 8);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 155 * 1024, 0, out, ticks, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
        unsigned long long h[256 * 16]; hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        double tot = 0; int n = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { tot += (double)h[b * 16 + w]; ++n; }
        const double ns_piece = tot / n * 10.0 / ((double)iters * 8);
        printf("%-44s %2d waves/CU: %6.1f ns per piece and wave -> %6.1f KB/us per CU\n", name, waves, ns_piece, waves * 1.024 / ns_piece * 1000.0);
    };
    for (int waves : {1, 4, 8, 12, 16}) {
        run("one matrix, full", k_loop<false, 0>, waves);
        run("W1/W3 pairs, full", k_loop<true, 0>, waves);
    }
    for (int waves : {4, 8}) {
        run("one matrix, plain loop, no chain", k_loop<false, 1>, waves);
        run("one matrix, plain loop, no dots", k_loop<false, 2>, waves);
        run("one matrix, plain loop, LDS reads only", k_loop<false, 3>, waves);
        run("pairs, plain loop, no chain", k_loop<true, 1>, waves);
        run("pairs, plain loop, no dots", k_loop<true, 2>, waves);
        run("pairs, plain loop, LDS reads only", k_loop<true, 3>, waves);
    }
    return 0;
}
