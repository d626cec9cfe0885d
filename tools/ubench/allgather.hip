// micro-benchmark: one strict all-to-all edge of k_layers (x -> every workgroup's rmsnorm prologue) as back-to-back rounds:
//   256 workgroups x 1024 threads, workgroup b owns 16 elements of a 4096-float vector, every workgroup needs all of it in LDS.
//   MODE 0  the kernel's form: write-through stores -> s_waitcnt vmcnt(0) -> flag line; lane i of waves 0..3 polls line i -> barrier -> coherent 16-byte loads
//   MODE 1  data-tagged granules: one 8-byte {value, tag} write-through store per element, no drain, no flag; every thread sweeps ITS 4 granules until the tags match
//   MODE 2  as 1, but only the lanes whose tags are missing re-read (exec-masked retry instead of a whole-wave re-sweep)
// W = 16-byte nt weight loads per thread requested in front of the round and consumed behind it (0: bare latency; 4: the 64 KiB per CU k_layers keeps in front of a poll).
// Every value read is checked (hash of round and index).   hipcc --offload-arch=gfx950 -O3 -o allgather allgather.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kAux = 17;   // sc0 | sc1
constexpr int N = 4096, PER = 16, LINE = 16;   // dwords between two flag lines

__device__ __forceinline__ unsigned val(unsigned r, unsigned i) { unsigned h = r * 0x9E3779B1u ^ i * 0xC2B2AE3Du; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; return h; }

template <int MODE, int W>
__global__ void __launch_bounds__(1024) k_edge(unsigned* buf, unsigned* flags, const v4i* weights, size_t wmask, int rounds, unsigned* err, int* sink) {
    __shared__ unsigned xs[N];
    const unsigned b = blockIdx.x, tid = threadIdx.x;
    unsigned bad = 0; int acc = 0;
    size_t woff = ((size_t)b * 1024 + tid);
    for (unsigned r = 1; r <= (unsigned)rounds; ++r) {
        v4i wreg[W > 0 ? W : 1];
        if (W > 0) {
#pragma unroll
            for (int k = 0; k < W; ++k) { wreg[k] = __builtin_nontemporal_load(weights + (woff & wmask)); woff += 256 * 1024; }
        }
        if (MODE == 0) {
            unsigned* x = buf + (size_t)(r & 1u) * N;
            if (tid < PER) __hip_atomic_store(x + b * PER + tid, val(r, b * PER + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 64) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) __hip_atomic_store(flags + b * LINE, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < 256) {
                while (true) {
                    const unsigned f = __hip_atomic_load(flags + tid * LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all(f >= r)) break;
                }
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(x, 0, N * 4, 0x00020000);
            const v4u v = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(tid * 16), 0, kAux));
            xs[tid * 4] = v.x; xs[tid * 4 + 1] = v.y; xs[tid * 4 + 2] = v.z; xs[tid * 4 + 3] = v.w;
        } else {
            unsigned long long* g = (unsigned long long*)buf + (size_t)(r & 1u) * N;
            if (tid < PER) __hip_atomic_store(g + b * PER + tid, ((unsigned long long)r << 32) | val(r, b * PER + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(g, 0, N * 8, 0x00020000);
            v4u a, c;
            if (MODE == 1) {
                while (true) {
                    asm volatile("" ::: "memory");      // (the loads are re-issued every pass: without it the compiler deletes the loop)
                    a = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(tid * 32), 0, kAux));
                    c = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(tid * 32 + 16), 0, kAux));
                    const bool ok = a.y == r && a.w == r && c.y == r && c.w == r;
                    if (__all(ok)) break;
                }
            } else {
                bool ok = false;
                while (!ok) {
                    asm volatile("" ::: "memory");
                    a = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(tid * 32), 0, kAux));
                    c = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(tid * 32 + 16), 0, kAux));
                    ok = a.y == r && a.w == r && c.y == r && c.w == r;
                }
            }
            xs[tid * 4] = a.x; xs[tid * 4 + 1] = a.z; xs[tid * 4 + 2] = c.x; xs[tid * 4 + 3] = c.z;
        }
        __syncthreads();
        // every thread checks 4 elements of the assembled vector (another thread's), as the chain / quantizer would read them
        const unsigned j = ((tid + 257u) & 1023u) * 4;
        bad += (xs[j] != val(r, j)) + (xs[j + 1] != val(r, j + 1)) + (xs[j + 2] != val(r, j + 2)) + (xs[j + 3] != val(r, j + 3));
        if (W > 0) {
#pragma unroll
            for (int k = 0; k < W; ++k) acc += wreg[k].x ^ wreg[k].y ^ wreg[k].z ^ wreg[k].w;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(err, bad);
    if (acc == 0x12345) *sink = acc;
}

int main() {
    unsigned *buf, *flags, *err; int* sink; v4i* weights;
    const size_t wbytes = (size_t)1 << 32;     // 4 GiB of "weights": the stream comes from HBM, not from the caches
    hipMalloc(&buf, 2 * N * 8); hipMalloc(&flags, 256 * LINE * 4); hipMalloc(&err, 4); hipMalloc(&sink, 4); hipMalloc(&weights, wbytes);
    hipMemset(weights, 1, wbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 4000;
    auto run = [&](const char* name, auto kern) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(buf, 0, 2 * N * 8); hipMemset(flags, 0, 256 * LINE * 4); hipMemset(err, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, buf, flags, weights, wbytes / 16 - 1, rep ? rounds : 50, err, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            if (rep) { float ms; unsigned e; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
                printf("%-78s %7.2f us per round   wrong values %u\n", name, ms * 1000 / rounds, e); }
        }
    };
    run("flags: drain + flag line, 4 waves poll, coherent read          W=0", k_edge<0, 0>);
    run("granules {value, tag}: whole-wave sweep                        W=0", k_edge<1, 0>);
    run("granules {value, tag}: per-lane retry                          W=0", k_edge<2, 0>);
    run("flags                                                          W=2 (32 KiB per CU in front)", k_edge<0, 2>);
    run("granules, whole-wave sweep                                     W=2", k_edge<1, 2>);
    run("granules, per-lane retry                                       W=2", k_edge<2, 2>);
    run("flags                                                          W=4 (64 KiB per CU in front)", k_edge<0, 4>);
    run("granules, whole-wave sweep                                     W=4", k_edge<1, 4>);
    run("granules, per-lane retry                                       W=4", k_edge<2, 4>);
    return 0;
}
