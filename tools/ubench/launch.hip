// micro-benchmark: cost of a dependent kernel boundary on gfx950 for different launch shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { char pad[256]; };
__global__ void k_empty(float* p, Big b) { if (p == nullptr) p[0] = b.pad[0]; }
__global__ void k_touch(float* p, Big b) { extern __shared__ float sm[]; sm[threadIdx.x] = threadIdx.x; __syncthreads(); if (threadIdx.x == 0 && sm[1] < 0) p[blockIdx.x] = sm[1]; }
int main() {
    float* p; hipMalloc(&p, 1 << 20); Big b{};
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto bench = [&](const char* name, int n, auto launch) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) launch();
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(e0, st); for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-60s %6.2f us per kernel\n", name, ms * 1000 / (10 * n));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
    bench("empty 1 WG x 64", 200, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, p, b); });
    bench("empty 256 WG x 256", 200, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, p, b); });
    bench("empty 256 WG x 1024", 200, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, st, p, b); });
    bench("empty 512 WG x 512", 200, [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, st, p, b); });
    bench("empty 2048 WG x 256", 200, [&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st, p, b); });
    bench("lds-touch 256 WG x 1024, 32 KB LDS", 200, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(1024), 32768, st, p, b); });
    hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    bench("lds-touch 256 WG x 1024, 120 KB LDS", 200, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(1024), 120 * 1024, st, p, b); });
    bench("lds-touch 256 WG x 256, 8 KB LDS", 200, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 8192, st, p, b); });
    bench("lds-touch 32 WG x 256, 8 KB LDS", 200, [&] { hipLaunchKernelGGL(k_touch, dim3(32), dim3(256), 8192, st, p, b); });
    return 0;
}
