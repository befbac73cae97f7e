// Diagnostic probe (not product): lane maps and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((vector_size(16)));

__global__ void probe_map(const float* a, const float* b, float* d, int cbsz_mode) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (cbsz_mode == 0) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    else if (cbsz_mode == 1) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 2, 1, 0);   // cbsz=2: groups of 4 blocks share A of block abid=1 in group
    else c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 3, 0);                        // cbsz=4: all 16 blocks share A of block 3
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

template <int KIND>
__global__ void probe_rate(float* out, int iters) {
    const int l = threadIdx.x & 63;
    float a = 1.0f + l * 1e-3f, b = 0.5f - l * 1e-3f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);
        } else {   // dependent chain of 4x4x1 (latency)
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0) / (8.0f * iters); }
    out[1 + (blockIdx.x * blockDim.x + threadIdx.x) % 64] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    float *a, *b, *d;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024 + 1024);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int mode = 0; mode < 3; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            for (int l = 0; l < 64; ++l) { ha[l] = pass == 0 ? l + 1 : 1; hb[l] = pass == 0 ? 1 : l + 1; }
            hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
            hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
            probe_map<<<1, 64>>>(a, b, d, mode);
            hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
            printf("mode %d: source %s lane of D[lane][reg]:\n", mode, pass == 0 ? "A" : "B");
            for (int l = 0; l < 64; ++l) {
                printf(" l%02d:", l);
                for (int r = 0; r < 4; ++r) printf("%3d", (int)hd[l * 4 + r] - 1);
                if (l % 8 == 7) printf("\n");
            }
        }
    }
    for (int waves = 1; waves <= 2; ++waves) {
        float h;
        probe_rate<0><<<256 * 4, 64 * waves>>>(d, 20000); hipDeviceSynchronize(); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("4x4x1 independent x8, %d wave(s)/block, 4 blocks/CU: %.2f cycles per MFMA per wave\n", waves, h);
        probe_rate<1><<<256 * 4, 64 * waves>>>(d, 20000); hipDeviceSynchronize(); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("16x16x4 independent x8, %d wave(s)/block: %.2f cycles per MFMA per wave\n", waves, h);
        probe_rate<2><<<256 * 4, 64 * waves>>>(d, 20000); hipDeviceSynchronize(); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("4x4x1 dependent chain, %d wave(s)/block: %.2f cycles per MFMA per wave\n", waves, h);
    }
    return 0;
}
