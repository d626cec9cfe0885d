// spoll.hip -- can a flag poll avoid the CU's in-order vector-memory pipeline?  Every workgroup (one per CU, 16 waves) requests `depth` KiB of weights (nt) and
// then polls a flag line that ANOTHER workgroup (the next block: another XCD) raises `delay` us after the launch's start with a write-through store.
// Polls: vector (relaxed agent-scope atomic load: global_load sc1 -- queues behind the weight requests) against SCALAR (s_load_dword glc: the scalar cache's own
// path to L2 / the fabric).  Flag memory: hipMalloc (L2-cacheable: a scalar load may be served a stale line by this XCD's L2) and uncached (hipDeviceMallocUncached).
// Reported: observation latency = (time the poller saw the flag) - (time the writer's store was issued), median / max over workgroups, 100 MHz clock for all XCDs.
//   hipcc --offload-arch=gfx950 -O3 -o bin/spoll spoll.hip && bin/spoll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned sload(const unsigned* p) {
    unsigned v;
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long ua = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ua) : "memory");
    return v;
}
template <int MODE>   // 0 vector poll, 1 scalar poll, 2 scalar poll of the flag then vector read of a 1 KiB payload
__global__ void __launch_bounds__(1024) k(const char* buf, unsigned slice, int depth_kib, int delay_ticks, unsigned* flags, const float* payload, unsigned long long* out, int* sinkp) {
    const char* base = buf + (size_t)blockIdx.x * slice;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)slice, 0x00020000);
    const int tid = threadIdx.x, n = depth_kib / 16;
    v4i acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) acc += __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, i * 16384 + tid * 16, 0, 2));
    const int G = gridDim.x;
    unsigned long long tw = 0, ts = 0, tp = 0;
    if (tid >= 960) {   // wave 15 writes the NEXT block's flag after the delay
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)delay_ticks) __builtin_amdgcn_s_sleep(1);
        tw = __builtin_amdgcn_s_memrealtime();
        if (tid == 960) { __hip_atomic_store(flags + ((blockIdx.x + 1) % G) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); out[blockIdx.x * 4 + 0] = tw; }
    }
    if (tid < 64) {     // wave 0 polls this block's flag
        const unsigned* line = flags + blockIdx.x * 16;
        const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
        bool seen = false;
        while (!seen) {
            unsigned f;
            if (MODE == 0) f = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else f = sload(line);
            seen = __all(f != 0u);
            if (__builtin_amdgcn_s_memrealtime() - tb > 200000ull) break;      // 2 ms: never seen (a stale line)
        }
        ts = __builtin_amdgcn_s_memrealtime();
        float4 pv = make_float4(0, 0, 0, 0);
        if (MODE == 2) { const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(payload), 0, (int)(gridDim.x * 1024), 0x00020000);
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rp, (blockIdx.x * 256 + tid * 4) * 4, 0, 17));
            pv = make_float4(t.x, t.y, t.z, t.w); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        tp = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) { out[blockIdx.x * 4 + 1] = seen ? ts : 0ull; out[blockIdx.x * 4 + 2] = tp; }
        if (pv.x == 1.2345f) *sinkp = 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    if (tid == 0) out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime() - t0;
    if (acc.x == 0x12345678) *sinkp = acc.y + acc.z + acc.w;
}
int main() {
    const int G = 256; const unsigned slice = 4u << 20;
    char* buf; hipMalloc(&buf, (size_t)G * slice);
    unsigned long long* out; hipMalloc(&out, G * 4 * 8); int* sp; hipMalloc(&sp, 4);
    float* payload; hipMalloc(&payload, G * 1024);
    unsigned* fl_c; hipMalloc(&fl_c, G * 64);
    unsigned* fl_u = nullptr;
    if (hipExtMallocWithFlags((void**)&fl_u, G * 64, hipDeviceMallocUncached) != hipSuccess) { printf("no uncached allocation\n"); fl_u = nullptr; }
    std::vector<unsigned long long> h(G * 4);
    for (int mem = 0; mem < 2; ++mem) {
        unsigned* fl = mem ? fl_u : fl_c; if (!fl) continue;
        for (int mode = 0; mode < 3; ++mode)
        for (int depth : {0, 64, 128, 192})
        for (int delay : {100, 400}) {            // ticks of 10 ns: 1 us, 4 us after the start
            std::vector<double> lat, pay, tot; int never = 0;
            for (int rep = 0; rep < 6; ++rep) {
                hipMemset(buf, rep + 1, (size_t)G * slice); hipMemset(fl, 0, G * 64); hipMemset(out, 0, G * 32); hipDeviceSynchronize();
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(1024), 0, 0, buf, slice, depth, delay, fl, payload, out, sp);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(1024), 0, 0, buf, slice, depth, delay, fl, payload, out, sp);
                else hipLaunchKernelGGL(k<2>, dim3(G), dim3(1024), 0, 0, buf, slice, depth, delay, fl, payload, out, sp);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                hipMemcpy(h.data(), out, G * 4 * 8, hipMemcpyDeviceToHost);
                if (rep == 0) continue;
                for (int g = 0; g < G; ++g) {
                    const unsigned long long tw = h[((g + G - 1) % G) * 4 + 0], ts = h[g * 4 + 1], tp = h[g * 4 + 2];
                    if (!ts) { ++never; continue; }
                    lat.push_back((double)((long long)ts - (long long)tw) * 0.01); pay.push_back((double)(tp - ts) * 0.01); tot.push_back(h[g * 4 + 3] * 0.01);
                }
            }
            std::sort(lat.begin(), lat.end()); std::sort(pay.begin(), pay.end()); std::sort(tot.begin(), tot.end());
            if (lat.empty()) { printf("%-8s %-22s depth %3d KiB delay %3.1f us: NEVER seen (%d)\n", mem ? "uncached" : "hipMalloc", mode == 0 ? "vector poll" : mode == 1 ? "scalar poll" : "scalar poll + payload", depth, delay * 0.01, never); continue; }
            printf("%-8s %-22s depth %3d KiB/CU, flag %3.1f us after start: seen %5.2f us after the store (median; p90 %5.2f, max %5.2f) never %d | payload read %5.2f | kernel %5.2f us\n",
                   mem ? "uncached" : "hipMalloc", mode == 0 ? "vector poll" : mode == 1 ? "scalar poll" : "scalar poll + payload", depth, delay * 0.01,
                   lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat.back(), never, pay[pay.size() / 2], tot[tot.size() / 2]);
        }
    }
    return 0;
}
