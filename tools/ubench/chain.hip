// micro-benchmark: how fast can ONE wave run a dependent fp32 FMA chain on gfx950, alone / next to other waves?
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang fp contract(off)
__global__ void k_chain(float* out, const float* in, int n, int active_lanes) {
    float l = in[threadIdx.x];
    float x = in[64 + threadIdx.x];
    if ((threadIdx.x & 63) < active_lanes) {
        for (int i = 0; i < n; i += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) l = __builtin_fmaf(x, x, l);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = l;
}
__global__ void k_chain_lds(float* out, const float* in, int n) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < n * 4; i += blockDim.x) sm[i] = in[i & 127];
    __syncthreads();
    float l = 0.f;
    if (threadIdx.x < 4) {
        const float* p = sm + threadIdx.x * n;
        for (int k = 0; k < n; k += 4) { float4 v = *(const float4*)(p + k); l = __builtin_fmaf(v.x, v.x, l); l = __builtin_fmaf(v.y, v.y, l); l = __builtin_fmaf(v.z, v.z, l); l = __builtin_fmaf(v.w, v.w, l); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = l;
}
int main() {
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 1 << 22); hipMemset(in, 0, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-50s %8.2f us per launch\n", name, ms * 1000 / 20);
    };
    run("empty-ish: 1 WG x 64 thr, n=16", [&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, in, 16, 4); });
    run("1 WG x 64 thr, 4 lanes, 1024 dep FMAs", [&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, in, 1024, 4); });
    run("1 WG x 64 thr, 4 lanes, 16384 dep FMAs", [&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, in, 16384, 4); });
    run("1 WG x 64 thr, 64 lanes, 16384 dep FMAs", [&] { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, in, 16384, 64); });
    run("512 WG x 512 thr, 4 lanes/wave, 16384 dep FMAs", [&] { hipLaunchKernelGGL(k_chain, dim3(512), dim3(512), 0, 0, out, in, 16384, 4); });
    run("512 WG x 512 thr: LDS-fed chain n=1024/lane", [&] { hipLaunchKernelGGL(k_chain_lds, dim3(512), dim3(512), 16384, 0, out, in, 1024); });
    run("512 WG x 512 thr: LDS-fed chain n=8192/lane", [&] { hipLaunchKernelGGL(k_chain_lds, dim3(512), dim3(512), 131072, 0, out, in, 8192); });
    return 0;
}
