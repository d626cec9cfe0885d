// micro-benchmark: issue cost (shader cycles per wave64 instruction, per SIMD) of the VALU instructions of the prefill GEMM's fp32
// chain step -- v_cvt_f32_i32, v_mul_f32, v_fma_f32 and their packed forms -- alone and next to int8 MFMAs issued by the same wave,
// at 1 and 4 waves per SIMD.  Decides whether the chain is written with packed or scalar fp32 instructions.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE, int MFMA>
__global__ void k(unsigned long long* out, float* sink, int iters) {
    float a[16], b = 1.0001f, c = 0.5f;
    f2 p[16]; f2 pb = {1.0001f, 0.9999f};
    int q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; p[i] = f2{(float)i, (float)threadIdx.x}; q[i] = threadIdx.x * 7 + i; }
    v16i d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i ma = {1, 2, 3, 4}, mb = {5, 6, 7, 8};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MFMA) { asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d) : "v"(ma), "v"(mb)); }
        if (MFMA == 2) { asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(d) : "v"(ma), "v"(mb)); }
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define CVT(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(q[i]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
        if (MODE == 0) { REP16(FMA) }
        if (MODE == 1) { REP16(MUL) }
        if (MODE == 2) { REP16(CVT) }
        if (MODE == 3) { REP16(PKFMA) }
        if (MODE == 4) { REP16(PKMUL) }
        if (MODE == 5) { REP16(PKADD) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 1.2345f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int MFMA>
void run(const char* name, unsigned long long* dout, float* sink) {
    for (int threads : {256, 1024}) {
        const int iters = 2000;
        hipLaunchKernelGGL((k<MODE, MFMA>), dim3(256), dim3(threads), 0, 0, dout, sink, iters);
        hipLaunchKernelGGL((k<MODE, MFMA>), dim3(256), dim3(threads), 0, 0, dout, sink, iters);
        hipDeviceSynchronize();
        unsigned long long h[256];
        hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += (double)h[i];
        m /= 256;
        const int wps = threads / 256;
        // s_memtime ticks at 100 MHz on gfx950?  report raw ticks per iteration AND per (instruction x waves on the SIMD)
        printf("%-10s mfma=%d waves/SIMD=%d: %8.2f ticks/iter  -> %6.3f ticks per VALU instr per SIMD\n", name, MFMA, wps, m / iters, m / iters / (16.0 * wps));
    }
}

int main() {
    unsigned long long* dout; float* sink;
    hipMalloc(&dout, 256 * 8); hipMalloc(&sink, 4);
    run<0, 0>("fma", dout, sink); run<1, 0>("mul", dout, sink); run<2, 0>("cvt", dout, sink);
    run<3, 0>("pk_fma", dout, sink); run<4, 0>("pk_mul", dout, sink); run<5, 0>("pk_add", dout, sink);
    run<0, 1>("fma", dout, sink); run<3, 1>("pk_fma", dout, sink); run<2, 1>("cvt", dout, sink);
    run<0, 2>("fma", dout, sink); run<3, 2>("pk_fma", dout, sink);
    return 0;
}
