// micro-benchmark: latency of a grid barrier over 256 one-per-CU workgroups on MI355X, several implementations
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int kAux = 17;   // sc0 | sc1

template <int MODE>
__global__ void __launch_bounds__(1024) k_bar(unsigned* bar, int iters, int sleep) {
    extern __shared__ char lds[];
    unsigned epoch = 0;
    const unsigned nwg = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        epoch += 1;
        if (MODE == 0) {            // single counter, agent atomics
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nwg) if (sleep) __builtin_amdgcn_s_sleep(1);
            }
        } else if (MODE == 1) {     // flag array, one wave polls 1 KiB
            if (threadIdx.x < 64) {
                if (threadIdx.x == 0) __hip_atomic_store(bar + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(bar, 0, (int)(nwg * 4), 0x00020000);
                while (true) {
                    const v4u f = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(threadIdx.x * 16), 0, kAux);
                    const bool ok = f.x >= epoch && f.y >= epoch && f.z >= epoch && f.w >= epoch;
                    if (__all(ok)) break;
                    if (sleep) __builtin_amdgcn_s_sleep(1);
                }
            }
        } else if (MODE == 2) {     // flag array with 64-byte spacing (one line per workgroup), 4 waves poll
            if (threadIdx.x < 256) {
                if (threadIdx.x == 0) __hip_atomic_store(bar + blockIdx.x * 16, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __shared__ int done;
                if (threadIdx.x == 0) done = 0;
                while (true) {
                    const unsigned f = __hip_atomic_load(bar + threadIdx.x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all(f >= epoch)) break;
                    if (sleep) __builtin_amdgcn_s_sleep(1);
                }
            }
        } else if (MODE == 3) {     // two-level: 16 group counters (16 workgroups each, 4 KiB apart) + root counter; everyone polls the root
            if (threadIdx.x == 0) {
                unsigned* g = bar + 1024 + (blockIdx.x & 15) * 1024;
                const unsigned mine = __hip_atomic_fetch_add(g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (mine + 1 == epoch * (nwg / 16)) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * 16) if (sleep) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

int main() {
    unsigned* bar; hipMalloc(&bar, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](const char* name, auto kern, int sleep) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024);
        hipMemset(bar, 0, 1 << 20);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 90 * 1024, 0, bar, 10, sleep); hipDeviceSynchronize();
        hipMemset(bar, 0, 1 << 20);
        hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 90 * 1024, 0, bar, iters, sleep); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-70s %8.2f us per barrier\n", name, ms * 1000 / iters);
    };
    run("single counter, sleep", k_bar<0>, 1);
    run("single counter, no sleep", k_bar<0>, 0);
    run("flag array 1 KiB, one wave polls, sleep", k_bar<1>, 1);
    run("flag array 1 KiB, one wave polls, no sleep", k_bar<1>, 0);
    run("flag array 64 B apart, 4 waves poll, sleep", k_bar<2>, 1);
    run("flag array 64 B apart, 4 waves poll, no sleep", k_bar<2>, 0);
    run("two-level counters, root polled by all, sleep", k_bar<3>, 1);
    run("two-level counters, root polled by all, no sleep", k_bar<3>, 0);
    return 0;
}
