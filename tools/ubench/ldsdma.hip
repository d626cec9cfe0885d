// micro-benchmark: a weight stream through an LDS ring filled by ONE loader wave per CU with LDS-DMA (buffer_load_dwordx4 ... lds) and drained by
// four consumer waves -- the skeleton of the engine kernel (flm_engine.h).  Measures what the loader alone sustains (consumers release at once),
// what loader + LDS-reading consumers sustain, and checks every byte that went through the ring (sum of all dwords against the host's).
//   usage: ldsdma [reps] [K] [rows]         (int8 matrix [rows][K] + fp32 scales [rows][K/64]; a piece = 4 rows x 256 B, a slot = 16 pieces + 1 scale piece)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

constexpr int kSlotW = 16 * 1024, kSlotBytes = 17 * 1024;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// one 1 KiB piece: lane's 16 bytes from (rsrc, voff + soff) to LDS at dst + 16 * lane
template <bool NT>
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned dst) {
    if constexpr (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen nt lds" :: "v"(voff), "s"(r), "s"(dst), "s"(soff) : "memory");
    else              asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(r), "s"(dst), "s"(soff) : "memory");
}

// NLOAD loader waves (fill F by loader F % NLOAD), 4 consumer waves; a piece = RP rows x (1024 / RP) bytes; a "unit" = RP rows
// ctl: fill_seq[NSLOT] (slot holds fill number fill_seq * NSLOT + slot - NSLOT ... i.e. value F / NSLOT + 1 once fill F has landed), free_seq[NSLOT]
template <int NSLOT, int DEPTH, int NLOAD, int RP, bool NT>
__global__ void __launch_bounds__(64 * (4 + NLOAD)) k_ring(const char* W, const float* S, int K, int rows, int mode, unsigned* sums, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned* fill_seq = reinterpret_cast<unsigned*>(lds + NSLOT * kSlotBytes);
    unsigned* free_seq = fill_seq + NSLOT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 2 * NSLOT) fill_seq[threadIdx.x] = 0;
    __syncthreads();
    constexpr int SEG = 1024 / RP, LPR = 64 / RP;                                 // bytes per row per piece, lanes per row
    const int sn = K / 64, PPU = K / SEG, U = rows / RP, c = blockIdx.x, ncu = gridDim.x;
    const int upc = c < U % ncu ? U / ncu + 1 : U / ncu;                          // units of this CU: c, c + ncu, ...
    const int npieces = upc * PPU, NF = (npieces + 15) / 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (wave < NLOAD) {
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(W), 0, rows * K, 0x00020000);
        const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, rows * sn * 4, 0x00020000);
        const unsigned voff = (lane / LPR) * K + (lane % LPR) * 16;
        int nissued = 0;
        for (int F = wave; F < NF; F += NLOAD) {
            const int slot = F % NSLOT; const unsigned need = F / NSLOT;
            while (lds_ld(free_seq + slot) < need) __builtin_amdgcn_s_sleep(1);
            const unsigned dst = (unsigned)(uintptr_t)(lds + slot * kSlotBytes);
            {   // the scale bytes of the slot (1 KiB per 16 KiB of weights; here simply the F-th KiB of this CU's share: the volume is what matters)
                const size_t so = ((size_t)c * NF + F) * 1024 + lane * 16;
                dma16<NT>(rS, so < (size_t)rows * sn * 4 ? (unsigned)so : 0x80000000u, 0, dst + kSlotW);
            }
            int i = (F * 16) / PPU, cb = (F * 16) % PPU;                          // unit index of this CU, column block: the fill's first piece
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                // (a piece past the end: the out-of-range offset sits in the VGPR operand, the only one the hardware bounds-checks)
                const bool live = i < upc;
                dma16<NT>(rW, live ? voff : 0x80000000u, live ? (unsigned)((c + ncu * i) * RP * K + cb * SEG) : 0u, dst + p * 1024);
                if (++cb == PPU) { cb = 0; ++i; }
            }
            ++nissued;
            // this loader's fill number (nissued - DEPTH) has landed once at most (DEPTH - 1) of its fills are outstanding
            if (nissued >= DEPTH) {
                if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                else if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(51)" ::: "memory");
                const int Fd = F - (DEPTH - 1) * NLOAD;
                if (lane == 0) lds_st(fill_seq + Fd % NSLOT, (unsigned)(Fd / NSLOT + 1));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int F = wave; F < NF; F += NLOAD) if (F + (DEPTH - 1) * NLOAD >= NF && lane == 0) lds_st(fill_seq + F % NSLOT, (unsigned)(F / NSLOT + 1));
        if (lane == 0 && ticks && wave == 0) ticks[c * 2] = __builtin_amdgcn_s_memrealtime() - t0;
    } else {
        const int w = wave - NLOAD;
        unsigned acc = 0;
        for (int F = w; F < NF; F += 4) {
            const int slot = F % NSLOT;
            while (lds_ld(fill_seq + slot) < (unsigned)(F / NSLOT + 1)) __builtin_amdgcn_s_sleep(1);
            if (mode == 0) {
                const v4u* s = reinterpret_cast<const v4u*>(lds + slot * kSlotBytes);
#pragma unroll
                for (int p = 0; p < 16; ++p) { const v4u v = s[p * 64 + lane]; acc += v.x + v.y + v.z + v.w; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_st(free_seq + slot, (unsigned)(F / NSLOT + 1));
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) sums[c * 4 + w] = acc;
        if (lane == 0 && ticks && w == 0) ticks[c * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0;
    }
}


// the engine's stream shape: the CU's fills go round robin over NSTREAM consumer streams (stream w: units c + ncu (w + NSTREAM m)), PIECES pieces
// per fill (+ PIECES / 8 scale KiB-eighths), TWO: a unit's pieces alternate between two matrices `rows` apart
template <int NSLOT, int DEPTH, int NLOAD, int NSTREAM, int PIECES, bool TWO>
__global__ void __launch_bounds__(64 * (4 + NLOAD)) k_ring2(const char* W, const float* S, int K, int rows, int mode, unsigned* sums, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SLOTB = PIECES * 1024 + PIECES * 64;
    unsigned* fill_seq = reinterpret_cast<unsigned*>(lds + NSLOT * SLOTB);
    unsigned* free_seq = fill_seq + NSLOT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 2 * NSLOT) fill_seq[threadIdx.x] = 0;
    __syncthreads();
    const int R = TWO ? rows / 2 : rows;                                          // rows per matrix
    const int PPU = K / 256, PPUV = TWO ? 2 * PPU : PPU, U = R / 4, c = blockIdx.x, ncu = gridDim.x;
    const int upc = c < U % ncu ? U / ncu + 1 : U / ncu;                          // units of this CU
    const int ups = upc / NSTREAM;                                                // units per stream (remainder dropped: a benchmark)
    const int fps = ups * PPUV / PIECES, NF = fps * NSTREAM;                      // fills per stream, fills of the CU
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (wave < NLOAD) {
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(W), 0, rows * K, 0x00020000);
        const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S), 0, rows * (K / 64) * 4, 0x00020000);
        const unsigned voff = (lane >> 4) * K + (lane & 15) * 16;
        int nissued = 0;
        for (int F = wave; F < NF; F += NLOAD) {
            const int slot = F % NSLOT; const unsigned need = F / NSLOT;
            while (lds_ld(free_seq + slot) < need) __builtin_amdgcn_s_sleep(1);
            const unsigned dst = (unsigned)(uintptr_t)(lds + slot * SLOTB);
            const int w = F % NSTREAM, sidx = F / NSTREAM;                        // stream, its fill number
            int j = sidx * PIECES, m = j / PPUV, rem = j - m * PPUV;
            {   // scale bytes: PIECES * 64 bytes per fill
                const size_t so = ((size_t)c * NF + F) * (PIECES * 64) + lane * 16;
                if (lane * 16 < PIECES * 64) dma16<true>(rS, so < (size_t)rows * (K / 64) * 4 ? (unsigned)so : 0x80000000u, 0, dst + PIECES * 1024);
            }
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const int cb = TWO ? rem >> 1 : rem, mat = TWO ? rem & 1 : 0;
                const unsigned base = (unsigned)((mat * R + 4 * (c + ncu * (w + NSTREAM * m))) * K + cb * 256);
                dma16<true>(rW, voff + base, 0, dst + p * 1024);
                if (++rem == PPUV) { rem = 0; ++m; }
            }
            ++nissued;
            if (nissued >= DEPTH) {
                constexpr int IPF = PIECES + 1;
                if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr ((DEPTH - 1) * IPF == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else if constexpr ((DEPTH - 1) * IPF == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                else if constexpr ((DEPTH - 1) * IPF == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else if constexpr ((DEPTH - 1) * IPF == 27) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
                const int Fd = F - (DEPTH - 1) * NLOAD;
                if (lane == 0) lds_st(fill_seq + Fd % NSLOT, (unsigned)(Fd / NSLOT + 1));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int F = wave; F < NF; F += NLOAD) if (F + (DEPTH - 1) * NLOAD >= NF && lane == 0) lds_st(fill_seq + F % NSLOT, (unsigned)(F / NSLOT + 1));
        if (lane == 0 && ticks && wave == 0) ticks[c * 2] = __builtin_amdgcn_s_memrealtime() - t0;
    } else {
        const int w = wave - NLOAD;
        unsigned acc = 0;
        for (int F = w; F < NF; F += 4) {
            const int slot = F % NSLOT;
            while (lds_ld(fill_seq + slot) < (unsigned)(F / NSLOT + 1)) __builtin_amdgcn_s_sleep(1);
            if (mode == 0) {
                const v4u* s = reinterpret_cast<const v4u*>(lds + slot * SLOTB);
#pragma unroll
                for (int p = 0; p < PIECES; ++p) { const v4u v = s[p * 64 + lane]; acc += v.x + v.y + v.z + v.w; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_st(free_seq + slot, (unsigned)(F / NSLOT + 1));
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) sums[c * 4 + w] = acc;
        if (lane == 0 && ticks && w == 0) ticks[c * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0;
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int K = argc > 2 ? atoi(argv[2]) : 4096;
    const int rows = argc > 3 ? atoi(argv[3]) : 22016;
    const size_t wbytes = (size_t)rows * K, sbytes = (size_t)rows * (K / 64) * 4;
    const int nbuf = (int)((3ull << 30) / wbytes);
    char* W; float* S; unsigned* sums; unsigned long long* ticks;
    if (hipMalloc(&W, wbytes * nbuf) != hipSuccess || hipMalloc(&S, sbytes * nbuf) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sums, 256 * 4 * 4); hipMalloc(&ticks, 256 * 2 * 8);
    std::vector<unsigned> hw(wbytes / 4), hs(sbytes / 4);
    unsigned x = 12345, ref = 0;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = x; ref += v; }
    for (auto& v : hs) { x = x * 1664525u + 1013904223u; v = x; }
    for (int b = 0; b < nbuf; ++b) { hipMemcpy(W + wbytes * b, hw.data(), wbytes, hipMemcpyHostToDevice); hipMemcpy((char*)S + sbytes * b, hs.data(), sbytes, hipMemcpyHostToDevice); }
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("matrix [%d][%d] int8 + scales: %.1f MB per launch, %d copies\n", rows, K, (wbytes + sbytes) * 1e-6, nbuf);
    auto run = [&](const char* name, int mode, auto kern, int nslot, int nload, int slotb = kSlotBytes) {
        const size_t ldsb = (size_t)nslot * slotb + 128;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        int b = 0;
        auto launch = [&] { hipLaunchKernelGGL(kern, dim3(256), dim3(64 * (4 + nload)), ldsb, st, W + wbytes * (b % nbuf), (const float*)((char*)S + sbytes * (b % nbuf)), K, rows, mode, sums, ticks); ++b; };
        for (int w = 0; w < 3; ++w) launch();
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000 / reps;
        std::vector<unsigned> h(1024); hipMemcpy(h.data(), sums, 4096, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> tk(512); hipMemcpy(tk.data(), ticks, 4096, hipMemcpyDeviceToHost);
        unsigned tot = 0; for (unsigned v : h) tot += v;
        unsigned long long lmax = 0, cmax = 0; for (int i = 0; i < 256; ++i) { if (tk[2 * i] > lmax) lmax = tk[2 * i]; if (tk[2 * i + 1] > cmax) cmax = tk[2 * i + 1]; }
        printf("%-44s %8.2f us/launch  %5.2f TB/s   in-kernel: loader %.2f us, consumers %.2f us (100 MHz ticks x 10 ns)%s\n", name, us, (wbytes + sbytes) / us * 1e-6,
               lmax * 0.01, cmax * 0.01, mode == 0 ? (tot == ref ? "   sum OK" : "   SUM MISMATCH") : "");
    };
    run("2 loaders x 2 deep, 1 stream, 16-piece fills (round 3's first ubench)", 1, k_ring<7, 2, 2, 4, true>, 7, 2);
    run("2 loaders x 2 deep, 1 stream, 16-piece fills", 1, k_ring2<7, 2, 2, 1, 16, false>, 7, 2, 17408);
    run("2 loaders x 2 deep, 8 streams, 16-piece fills", 1, k_ring2<7, 2, 2, 8, 16, false>, 7, 2, 17408);
    run("4 loaders x 2 deep, 1 stream, 8-piece fills", 1, k_ring2<14, 2, 4, 1, 8, false>, 14, 4, 8704);
    run("4 loaders x 2 deep, 8 streams, 8-piece fills", 1, k_ring2<14, 2, 4, 8, 8, false>, 14, 4, 8704);
    run("4 loaders x 2 deep, 8 streams, 8-piece fills, two matrices", 1, k_ring2<14, 2, 4, 8, 8, true>, 14, 4, 8704);
    run("4 loaders x 3 deep, 8 streams, 8-piece fills, two matrices", 1, k_ring2<14, 3, 4, 8, 8, true>, 14, 4, 8704);
    run("4 loaders x 2 deep, 2 streams, 8-piece fills", 1, k_ring2<14, 2, 4, 2, 8, false>, 14, 4, 8704);
    run("4 loaders x 2 deep, 4 streams, 8-piece fills", 1, k_ring2<14, 2, 4, 4, 8, false>, 14, 4, 8704);
    return 0;
}
