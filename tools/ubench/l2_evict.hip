// l2_evict.hip -- L2 as a landing space INSIDE one launch: a workgroup (one per CU) reads `warm` KiB of its slab and drops them (ordinary policy), then streams
// `mid` KiB of other bytes with cache policy `aux` (the GEMV's weight loads: nt), then reads the warm bytes again (nt) -- how long does that last read take, against
// the same amount of cold bytes?  Answers whether lines parked in the XCD's L2 survive the weight stream that passes between parking and use.
//   hipcc --offload-arch=gfx950 -O3 -o l2_evict l2_evict.hip && ./l2_evict
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int AUX>
__global__ void __launch_bounds__(1024) k(const char* buf, unsigned slice, int warm_kib, int mid_kib, unsigned long long* out, int* sinkp) {
    const char* base = buf + (size_t)blockIdx.x * slice;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)slice, 0x00020000);
    const int tid = threadIdx.x, n = warm_kib / 16, m = mid_kib / 16;               // 16 KiB per round of 1024 x 16 bytes
    v4i acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) acc += __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, i * 16384 + tid * 16, 0, 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < m; ++i) acc += __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (1 << 22) + i * 16384 + tid * 16, 0, AUX));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) acc += __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, i * 16384 + tid * 16, 0, 2));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) acc += __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (3 << 21) + i * 16384 + tid * 16, 0, 2));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
    const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = t3 - t2; out[blockIdx.x * 4 + 3] = t4 - t3; }
    if (acc.x == 0x12345678) *sinkp = acc.y + acc.z + acc.w;
}
int main() {
    const int G = 256; const unsigned slice = 8u << 20;
    char* buf; hipMalloc(&buf, (size_t)G * slice);
    unsigned long long* out; hipMalloc(&out, G * 4 * 8); int* sp; hipMalloc(&sp, 4);
    std::vector<unsigned long long> h(G * 4);
    const int auxs[] = {2, 0, 17, 19, 1, 16};
    const char* names[] = {"nt", "default", "sc0 sc1", "sc0 sc1 nt", "sc0", "sc1"};
    for (int ai = 0; ai < 6; ++ai)
    for (int warm : {32, 96})
    for (int mid : {0, 64, 128, 256}) {
        double a[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(buf, rep + 1, (size_t)G * slice);         // (2 GiB through the caches: nothing of the previous repetition stays)
            switch (auxs[ai]) {
            case 2:  hipLaunchKernelGGL(k<2>,  dim3(G), dim3(1024), 0, 0, buf, slice, warm, mid, out, sp); break;
            case 0:  hipLaunchKernelGGL(k<0>,  dim3(G), dim3(1024), 0, 0, buf, slice, warm, mid, out, sp); break;
            case 17: hipLaunchKernelGGL(k<17>, dim3(G), dim3(1024), 0, 0, buf, slice, warm, mid, out, sp); break;
            case 19: hipLaunchKernelGGL(k<19>, dim3(G), dim3(1024), 0, 0, buf, slice, warm, mid, out, sp); break;
            case 1:  hipLaunchKernelGGL(k<1>,  dim3(G), dim3(1024), 0, 0, buf, slice, warm, mid, out, sp); break;
            default: hipLaunchKernelGGL(k<16>, dim3(G), dim3(1024), 0, 0, buf, slice, warm, mid, out, sp); break;
            }
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, G * 4 * 8, hipMemcpyDeviceToHost);
            if (rep == 0) continue;
            for (int j = 0; j < 4; ++j) { std::vector<double> v; for (int g = 0; g < G; ++g) v.push_back(h[g * 4 + j] * 0.01); std::sort(v.begin(), v.end()); a[j] += v[G / 2] / 3; }
        }
        printf("stream policy %-10s warm %3d KiB/CU (%4.2f MB/XCD), stream %3d KiB/CU between: warm-up %5.2f us | stream %5.2f us | warm bytes again %5.2f us | as many cold bytes %5.2f us\n",
               names[ai], warm, warm * 32 / 1024.0, mid, a[0], a[1], a[2], a[3]);
    }
    return 0;
}
