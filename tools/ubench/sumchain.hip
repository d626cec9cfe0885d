// micro-benchmark: the softmax's sequential sum  s += e[t], t ascending (the reference's chain, tf_operators.cpp:180-183)  over T values that lie in LDS,
// one 1024-thread workgroup per CU, the other 15 waves waiting at a barrier -- as inside attn_head:
//   MODE 0  the kernel's form: a lone lane, 4 dependent adds per 16-byte LDS read, reads 28 adds ahead
//   MODE 1  blocks of 64 along the lanes of ONE wave: lane l holds e[64 b + l], step k: every lane adds its own term to its left neighbour's running sum (v_add with DPP
//           wave_shr:1) -- after 63 steps lane 63 holds the block's exact prefix; the carry enters through lane 0's term (fl(carry + e_0) is the chain's own first step)
// Every result is checked against the host's sequential sum, bit for bit.   hipcc --offload-arch=gfx950 -O3 -o sumchain sumchain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(1024) k_sum(const float* src, int T, int reps, float* out) {
    __shared__ __attribute__((aligned(16))) float sc[1024 + 64];
    __shared__ float red[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < 1024 + 64; t += 1024) sc[t] = t < T ? src[t] : 0.f;
    __syncthreads();
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {
            if (tid == 0) {
                float sum = 0.f; int t = 0;
#define ADD4(q) sum = __fadd_rn(sum, q.x); sum = __fadd_rn(sum, q.y); sum = __fadd_rn(sum, q.z); sum = __fadd_rn(sum, q.w);
#define ASTEP(q, off) ADD4(q) q = *reinterpret_cast<const float4*>(pp + (off)); __builtin_amdgcn_sched_barrier(0);
                if (T >= 32) {
                    const float* pp = sc;
                    float4 q0 = *reinterpret_cast<const float4*>(pp), q1 = *reinterpret_cast<const float4*>(pp + 4), q2 = *reinterpret_cast<const float4*>(pp + 8), q3 = *reinterpret_cast<const float4*>(pp + 12);
                    float4 q4 = *reinterpret_cast<const float4*>(pp + 16), q5 = *reinterpret_cast<const float4*>(pp + 20), q6 = *reinterpret_cast<const float4*>(pp + 24), q7 = *reinterpret_cast<const float4*>(pp + 28);
                    __builtin_amdgcn_sched_barrier(0);
                    for (; t + 32 <= T; t += 32, pp += 32) { ASTEP(q0, 32) ASTEP(q1, 36) ASTEP(q2, 40) ASTEP(q3, 44) ASTEP(q4, 48) ASTEP(q5, 52) ASTEP(q6, 56) ASTEP(q7, 60) }
                    if (t + 4 <= T) { ADD4(q0) t += 4; } if (t + 4 <= T) { ADD4(q1) t += 4; } if (t + 4 <= T) { ADD4(q2) t += 4; } if (t + 4 <= T) { ADD4(q3) t += 4; }
                    if (t + 4 <= T) { ADD4(q4) t += 4; } if (t + 4 <= T) { ADD4(q5) t += 4; } if (t + 4 <= T) { ADD4(q6) t += 4; }
                }
                for (; t < T; ++t) sum = __fadd_rn(sum, sc[t]);
                red[0] = sum;
            }
        } else if (MODE == 2) {
            // the lone lane with its LDS reads as inline assembly: four 16-byte reads per instruction group, ONE s_waitcnt per 16 elements (the compiler's own reads get a
            // wait in front of every register's first use: 6 instructions per 4 elements; here 21 per 16)
            if (tid == 0) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                float sum = 0.f; int t = 0;
                const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)sc;
                v4f q0, q1, q2, q3, q4, q5, q6, q7;
#define RD4(a0, a1, a2, a3, addr) asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(addr) : "memory");
#define WAIT4(a0, a1, a2, a3, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "memory");
#define ADDV(q) sum = __fadd_rn(sum, q.x); sum = __fadd_rn(sum, q.y); sum = __fadd_rn(sum, q.z); sum = __fadd_rn(sum, q.w);
                if (T >= 32) {
                    unsigned ad = base;
                    RD4(q0, q1, q2, q3, ad) { const unsigned a2 = ad + 64; RD4(q4, q5, q6, q7, a2) }
                    for (; t + 32 <= T; t += 32) {
                        ad += 128;
                        WAIT4(q0, q1, q2, q3, 4) ADDV(q0) ADDV(q1) ADDV(q2) ADDV(q3) RD4(q0, q1, q2, q3, ad)
                        { const unsigned a2 = ad + 64; WAIT4(q4, q5, q6, q7, 4) ADDV(q4) ADDV(q5) ADDV(q6) ADDV(q7) RD4(q4, q5, q6, q7, a2) }
                    }
                    WAIT4(q0, q1, q2, q3, 0) WAIT4(q4, q5, q6, q7, 0)
                    if (t + 4 <= T) { ADDV(q0) t += 4; } if (t + 4 <= T) { ADDV(q1) t += 4; } if (t + 4 <= T) { ADDV(q2) t += 4; } if (t + 4 <= T) { ADDV(q3) t += 4; }
                    if (t + 4 <= T) { ADDV(q4) t += 4; } if (t + 4 <= T) { ADDV(q5) t += 4; } if (t + 4 <= T) { ADDV(q6) t += 4; }
                }
                for (; t < T; ++t) sum = __fadd_rn(sum, sc[t]);
                red[0] = sum;
            }
        } else {
            if (wave == 0) {
                float carry = 0.f, pre = 0.f;
                const int nb = (T + 63) >> 6;
                float e = sc[lane];
                for (int b = 0; b < nb; ++b) {
                    const float en = sc[64 * (b + 1) + lane];                 // (the next block's terms; zeros past T: s + 0 = s)
                    if (lane == 0) e = __fadd_rn(carry, e);                     // the chain's step over e[64 b]: lane 0's term carries the prefix in
                    pre = e;
#pragma unroll
                    for (int k = 1; k < 64; ++k)
                        pre = __fadd_rn(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(pre), 0x138 /* wave_shr:1 */, 0xF, 0xF, true)), e);
                    carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pre), 63));
                    e = en;
                }
                if (lane == 0) red[0] = carry;
            }
        }
        __syncthreads();
        acc += red[0];
        __syncthreads();
    }
    if (tid == 0) { out[blockIdx.x * 2] = red[0]; out[blockIdx.x * 2 + 1] = acc; }
}
int main() {
    float *src, *out; hipMalloc(&src, 1088 * 4); hipMalloc(&out, 256 * 2 * 4);
    std::vector<float> h(1088);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = expf(-(float)(s >> 8) / 16777216.f * 12.f); }      // exp(-d), d in [0, 12): what a softmax sums
    h[3] = 1.f;
    hipMemcpy(src, h.data(), 1088 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    for (int T : {130, 301, 517, 901, 1024}) {
        float want = 0.f; for (int t = 0; t < T; ++t) want += h[t];
        for (int mode = 0; mode < 3; ++mode) {
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_sum<0>, dim3(256), dim3(1024), 0, 0, src, T, reps, out); else if (mode == 1) hipLaunchKernelGGL(k_sum<1>, dim3(256), dim3(1024), 0, 0, src, T, reps, out); else hipLaunchKernelGGL(k_sum<2>, dim3(256), dim3(1024), 0, 0, src, T, reps, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float got[2]; hipMemcpy(got, out, 8, hipMemcpyDeviceToHost);
            printf("T %4d  %-40s %7.3f us per sum   %s\n", T, mode == 2 ? "lone lane, asm reads, a wait per 16" : mode ? "64 lanes, DPP wave_shr:1 per step" : "lone lane, 16-byte reads (the kernel's)", ms * 1000 / reps, memcmp(&got[0], &want, 4) == 0 ? "bits ok" : "MISMATCH");
        }
    }
    return 0;
}
