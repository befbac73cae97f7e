// Diagnostic probe (not product): do f32 MFMA (16x16x4) and f32 VALU co-execute on one SIMD of gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((vector_size(16)));

// mode bit0: waves 0-3 run MFMA loop; bit1: waves 4-7 run VALU loop  (waves w and w+4 share a SIMD)
__global__ void __launch_bounds__(512) coexec(float* out, long long* cyc, int iters, int mode) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    float a = 1.0f + l * 1e-3f, b = 0.5f - l * 1e-3f;
    long long t0 = clock64();
    float res = 0.f;
    if (w < 4) {
        if (mode & 1) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            }
            res = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (mode & 2) {
            float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, v4 = a + 2, v5 = b + 2, v6 = a + 3, v7 = b + 3;
            for (int i = 0; i < iters * 8; ++i) {          // 8 x 8 = 64 v_fma per MFMA-loop iteration (4 MFMA = 128 cycles)
                v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
                v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
            }
            res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        }
    }
    long long t1 = clock64();
    if (blockIdx.x == 0 && l == 0) cyc[w] = t1 - t0;
    out[(blockIdx.x * 512 + threadIdx.x) % 4096] = res;
}

// one wave: per MFMA, K independent v_fma
template <int K>
__global__ void __launch_bounds__(64) mixed(float* out, long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    float a = 1.0f + l * 1e-3f, b = 0.5f - l * 1e-3f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k % 8] = fmaf(v[k % 8], a, b);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) v[(k + 4) % 8] = fmaf(v[(k + 4) % 8], a, b);
    }
    long long t1 = clock64();
    if (blockIdx.x == 0 && l == 0) cyc[0] = t1 - t0;
    out[(blockIdx.x * 64 + threadIdx.x) % 4096] = c0[0] + c1[1] + v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
}

template <int K>
void run_mixed(float* out, long long* cyc, int iters) {
    long long h;
    mixed<K><<<256 * 4, 64>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("one wave/SIMD, %2d v_fma per MFMA: %.1f cycles per MFMA\n", K, (double)h / (2.0 * iters));
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 64);
    const int iters = 20000;
    for (int mode = 1; mode <= 3; ++mode) {
        long long h[8];
        coexec<<<256, 512>>>(out, cyc, iters, mode);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("mode %d (1=MFMA waves only, 2=VALU waves only, 3=both): MFMA wave0 %.1f cyc/iter(4 MFMA), VALU wave4 %.1f cyc/iter(64 fma)\n",
               mode, (double)h[0] / iters, (double)h[4] / iters);
    }
    run_mixed<0>(out, cyc, iters); run_mixed<2>(out, cyc, iters); run_mixed<4>(out, cyc, iters); run_mixed<6>(out, cyc, iters);
    run_mixed<8>(out, cyc, iters); run_mixed<12>(out, cyc, iters); run_mixed<16>(out, cyc, iters);
    return 0;
}
