// micro-benchmark: the prompt path's int8 GEMMs on their own -- k_gemm_q8_ring (loader / consumer waves around an LDS ring) against the tile kernel
// k_gemm_q8_mfma, 7B shapes, with the ring kernel's per-wave wait accounting and ablations (FLM_ABLATE build).
//   usage: gemm_ring [tokens] [iters]          build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DFLM_ABLATE=1
//                                                      -Iinclude -Ifast-llama_amd/csrc -Itools/ubench tools/ubench/gemm_ring.hip -o tools/ubench/bin/gemm_ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "flm_kernels.h"
#include "gemm_ring_kernel.h"
using namespace flm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Prob { const char* name; int rows, n; };

template <int EPI, int WT, int WR, int NB>
float run_ring(const GemmArgs& g, int iters, int ablate, unsigned long long* trace, bool report) {
    using G = GrTile<WT, WR, NB>;
    constexpr int TRH = EPI == EPI_SWIGLU ? G::TR / 2 : G::TR;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_ring<EPI, WT, WR, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds));
    const int nblk = ((g.B + G::TT - 1) / G::TT) * ((g.rows + TRH - 1) / TRH), grid = nblk < 256 ? nblk : 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm_q8_ring<EPI, WT, WR, NB>), dim3(grid), dim3(kGrBlock), G::kLds, 0, g, (int*)nullptr, (unsigned long long*)nullptr, ablate);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_gemm_q8_ring<EPI, WT, WR, NB>), dim3(grid), dim3(kGrBlock), G::kLds, 0, g, (int*)nullptr, (unsigned long long*)nullptr, ablate);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1000.f / iters;
    if (report) {
        CK(hipMemset(trace, 0, (256 * 16 * 4 + 64 * 8 + 64) * 8));
        hipLaunchKernelGGL((k_gemm_q8_ring<EPI, WT, WR, NB>), dim3(grid), dim3(kGrBlock), G::kLds, 0, g, (int*)nullptr, trace, ablate);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> t(256 * 16 * 4 + 64 * 8 + 64);
        CK(hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost));
        double lt = 0, lw = 0, ln = 0, ct = 0, cw = 0, cn = 0; int nl = 0, nc = 0;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < kGrLoaders + G::NC; ++w) {
                const unsigned long long* q = &t[((size_t)b * 16 + w) * 4];
                if (q[1] <= q[0]) continue;
                if (w < kGrLoaders) { lt += q[1] - q[0]; lw += q[2]; ln += q[3]; ++nl; } else { ct += q[1] - q[0]; cw += q[2]; cn += q[3]; ++nc; }
            }
        printf("      [ticks per wave: loaders run %.0f, waiting for a free slot %.0f in %.0f waits | consumers run %.0f, waiting for a fill %.0f in %.0f waits]\n",
               lt / std::max(nl, 1), lw / std::max(nl, 1), ln / std::max(nl, 1), ct / std::max(nc, 1), cw / std::max(nc, 1), cn / std::max(nc, 1));
        { const unsigned long long* q = &t[256 * 16 * 4 + 64 * 8 + 48]; if (q[3] > q[1]) printf("      [clock: %.0f s_memtime ticks in %.2f us (100 MHz s_memrealtime) = %.0f MHz]\n", (double)(q[2] - q[0]), (q[3] - q[1]) / 100.0, (double)(q[2] - q[0]) / ((q[3] - q[1]) / 100.0)); }
        if (getenv("STAMPS")) {
            const unsigned long long* q = &t[256 * 16 * 4]; const unsigned long long z = q[0];
            printf("      stage: loader 0 [slot free, issued, previous published] | consumer 0 [fill seen, reads done]   (ticks from loader 0's first stage)\n");
            printf("      consumers at stages 0 / 16 / 32 / 48:");
            for (int w = 0; w < 12; ++w) printf("  [%d] %lld %lld %lld %lld", w, (long long)(q[64 * 8 + w * 4] - z), (long long)(q[64 * 8 + w * 4 + 1] - z), (long long)(q[64 * 8 + w * 4 + 2] - z), (long long)(q[64 * 8 + w * 4 + 3] - z));
            printf("\n");
            for (int s2 = 0; s2 < (getenv("STAMPS")[0] == '2' ? 40 : 0); ++s2) printf("      %2d: %7lld %7lld %7lld | %7lld %7lld\n", s2, (long long)(q[s2 * 8] - z), (long long)(q[s2 * 8 + 1] - z), (long long)(q[s2 * 8 + 2] - z), (long long)(q[s2 * 8 + 4] - z), (long long)(q[s2 * 8 + 5] - z));
        }
    }
    return us;
}
template <int EPI, int WT, int WR, int NB>
float run_tile(const GemmArgs& g, int iters) {
    using T = GemmTile<WT, WR, NB>;
    constexpr int TRH = EPI == EPI_SWIGLU ? T::TR / 2 : T::TR;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_q8_mfma<EPI, WT, WR, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, T::kLds));
    const int tiles = ((g.rows + TRH - 1) / TRH) * ((g.B + T::TT - 1) / T::TT);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm_q8_mfma<EPI, WT, WR, NB>), dim3(tiles), dim3(T::NT), T::kLds, 0, g);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_gemm_q8_mfma<EPI, WT, WR, NB>), dim3(tiles), dim3(T::NT), T::kLds, 0, g);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 20;
    const Prob probs[] = {{"qkv  ", 12288, 4096}, {"wo   ", 4096, 4096}, {"ffn13", 22016, 4096}, {"ffn2 ", 4096, 11008}};
    unsigned long long* trace; CK(hipMalloc(&trace, (256 * 16 * 4 + 64 * 8 + 64) * 8));
    for (const Prob& p : probs) {
        if (getenv("STAMPS") && p.rows != 12288) continue;
        const size_t wn = (size_t)p.rows * p.n, xn = (size_t)B * p.n; const int sn = p.n / 64;
        char *W, *X; float *sW, *sX, *sWT, *sXT, *out;
        CK(hipMalloc(&W, wn)); CK(hipMalloc(&X, xn)); CK(hipMalloc(&sW, (size_t)p.rows * sn * 4)); CK(hipMalloc(&sX, (size_t)B * sn * 4));
        CK(hipMalloc(&sWT, (size_t)p.rows * sn * 4)); CK(hipMalloc(&sXT, (size_t)B * sn * 4 + 64)); CK(hipMalloc(&out, (size_t)B * p.rows * 4));
        {
            std::vector<char> h(std::max(wn, xn)); unsigned s = 12345u;
            for (auto& c : h) { s = s * 1664525u + 1013904223u; c = (char)((int)(s >> 24) % 128); }
            CK(hipMemcpy(W, h.data(), wn, hipMemcpyHostToDevice)); CK(hipMemcpy(X, h.data(), xn, hipMemcpyHostToDevice));
            std::vector<float> f(std::max((size_t)p.rows, (size_t)B) * sn);
            for (auto& v : f) { s = s * 1664525u + 1013904223u; v = 1e-3f + (s >> 8) * 1e-10f; }
            CK(hipMemcpy(sW, f.data(), (size_t)p.rows * sn * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sWT, f.data(), (size_t)p.rows * sn * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(sX, f.data(), (size_t)B * sn * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sXT, f.data(), (size_t)B * sn * 4, hipMemcpyHostToDevice));
        }
        GemmArgs g{W, sW, X, sX, out, p.rows, p.n, p.rows, B, sXT, sWT};
        const double gmac = (double)B * p.rows * p.n * 1e-9;
        auto line = [&](const char* what, float us) { printf("  %-44s %8.1f us  %7.1f TMAC/s  (%4.1f %% of 1972)\n", what, us, gmac / us * 1e-3 * 1e3, gmac / us * 1e3 / 1972 * 100 * 1e-3); };
        printf("%s  %d tokens x %d rows x K %d  (%.1f GMAC)\n", p.name, B, p.rows, p.n, gmac);
        line("tile kernel 64 x 64", run_tile<EPI_STORE, 2, 2, 1>(g, iters));
        line("tile kernel 128 x 128", run_tile<EPI_STORE, 4, 2, 2>(g, iters));
        line("ring 128 x 192 <4,3,2>", run_ring<EPI_STORE, 4, 3, 2>(g, iters, 0, trace, true));
        line("ring 128 x 96  <4,3,1>", run_ring<EPI_STORE, 4, 3, 1>(g, iters, 0, trace, true));
        line("ring 128 x 64  <4,2,1>", run_ring<EPI_STORE, 4, 2, 1>(g, iters, 0, trace, true));
        line("ring <4,3,2> no chain", run_ring<EPI_STORE, 4, 3, 2>(g, iters, 1, trace, true));
        line("ring <4,3,2> no sync (consumers' bare loop)", run_ring<EPI_STORE, 4, 3, 2>(g, iters, 16, trace, true));
        line("ring <4,3,2> no sync, no chain", run_ring<EPI_STORE, 4, 3, 2>(g, iters, 17, trace, true));
        line("ring <4,3,1> no sync", run_ring<EPI_STORE, 4, 3, 1>(g, iters, 16, trace, true));
        line("ring <4,2,1> no sync", run_ring<EPI_STORE, 4, 2, 1>(g, iters, 16, trace, true));
        line("ring <4,3,2> no DMA (consumers alone)", run_ring<EPI_STORE, 4, 3, 2>(g, iters, 8, trace, true));
        line("ring <4,3,1> no DMA (consumers alone)", run_ring<EPI_STORE, 4, 3, 1>(g, iters, 8, trace, true));
        CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(sW)); CK(hipFree(sX)); CK(hipFree(sWT)); CK(hipFree(sXT)); CK(hipFree(out));
    }
    return 0;
}
