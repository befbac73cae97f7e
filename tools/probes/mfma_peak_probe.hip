// Diagnostic probe (not product): chip-wide sustained f32 MFMA rate (16x16x4) with every SIMD busy, random data, wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((vector_size(16)));
template <int LDS_READS>
__global__ void __launch_bounds__(1024) peak(float* out, int iters, float seed) {
    __shared__ float lds[4096];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed * (float)((i * 2654435761u) >> 8) * 1e-8f;
    __syncthreads();
    float a0 = 1.0f + l * 1.7e-3f * seed, b0 = 0.5f - l * 1.3e-3f * seed, a1 = a0 * 0.9f, b1 = b0 * 1.1f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        if (LDS_READS) {
            const float4 v = *reinterpret_cast<const float4*>(&lds[((i * 64 + l) * 4) & 4095]);
            a0 = v.x; b0 = v.y; a1 = v.z; b1 = v.w;
        }
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, c3, 0, 0, 0);
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) % 4096] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int LDS_READS>
void run(float* out, int threads, int blocks) {
    const int iters = 40000;
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    peak<LDS_READS><<<blocks, threads>>>(out, 1000, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    peak<LDS_READS><<<blocks, threads>>>(out, iters, 1.0f);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    const double flops = (double)blocks * (threads / 64) * iters * 4.0 * 2048.0;
    printf("lds_reads=%d blocks=%d threads=%d: %.3f ms, %.1f TFLOP/s\n", LDS_READS, blocks, threads, ms, flops / ms / 1e9);
}
int main() {
    float* out;
    (void)hipMalloc(&out, 4096 * 4);
    run<0>(out, 256, 256); run<0>(out, 512, 256); run<0>(out, 1024, 256);
    run<1>(out, 256, 256); run<1>(out, 512, 256); run<1>(out, 1024, 256);
    return 0;
}
