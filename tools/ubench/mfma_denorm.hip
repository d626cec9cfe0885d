// does v_mfma_f32_32x32x2_f32 keep subnormal products / inputs like v_mul_f32 / v_fma_f32 do?  (the scale product s = sW * sX of the
// 64 x 64 prompt GEMM tiles comes from this instruction)      hipcc --offload-arch=gfx950 -O3 -o mfma_denorm mfma_denorm.hip && ./mfma_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(const float* a, const float* b, float* out_mfma, float* out_mul) {
    const int lane = threadIdx.x, h = lane >> 5;
    const v16f z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const v16f S = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? 0.f : a[lane & 31], h ? 0.f : b[lane & 31], z, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h, j = lane & 31;
        out_mfma[i * 32 + j] = S[r];
        out_mul[i * 32 + j] = __fmul_rn(a[i], b[j]);
    }
}
int main() {
    float ha[32], hb[32], hm[1024], hv[1024];
    const float as[8] = {1e-20f, 3e-25f, 1.5e-30f, 7e-39f /* subnormal input */, 0.f, 1.0f, 1e-38f, 2.5e-19f};
    const float bs[8] = {1e-20f, 1e-15f, 1e-10f, 0.5f, 1e-30f, 1.2e-38f, 1e-7f, 3e-22f};
    for (int i = 0; i < 32; ++i) { ha[i] = as[i % 8] * (1.f + i / 64.f); hb[i] = bs[(i * 3) % 8] * (1.f + i / 128.f); }
    float *da, *db, *dm, *dv;
    hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dm, 4096); hipMalloc(&dv, 4096);
    hipMemcpy(da, ha, 128, hipMemcpyHostToDevice); hipMemcpy(db, hb, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dm, dv);
    hipMemcpy(hm, dm, 4096, hipMemcpyDeviceToHost); hipMemcpy(hv, dv, 4096, hipMemcpyDeviceToHost);
    int diff = 0, sub = 0, subdiff = 0;
    for (int i = 0; i < 1024; ++i) {
        const float host = ha[i / 32] * hb[i % 32];
        const bool is_sub = host != 0.f && host > -1.17549435e-38f && host < 1.17549435e-38f;
        sub += is_sub;
        if (memcmp(&hm[i], &hv[i], 4)) { ++diff; subdiff += is_sub; if (diff <= 6) printf("  a %.9g b %.9g: mfma %.9g  v_mul %.9g  host %.9g\n", ha[i / 32], hb[i % 32], hm[i], hv[i], host); }
    }
    printf("1024 products, %d subnormal on the host: %d differ between the MFMA and v_mul_f32 (%d of them subnormal)\n", sub, diff, subdiff);
    return 0;
}
