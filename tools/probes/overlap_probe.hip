// Diagnostic probe (not product): can an LDS-fed MFMA GEMM phase and the VALU attention-core loop overlap on one CU when
// they run in DIFFERENT waves of the same work-group (wave specialisation)?  Times: MFMA waves alone, VALU waves alone, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((vector_size(16)));
constexpr int ROWS = 64, LDQ = 244, LDX = 68, H = 8, DH = 10, I = 80;
struct V10 { float v[DH]; };
__device__ __forceinline__ void ld(V10& x, const float* p) {
#pragma unroll
    for (int c = 0; c < DH; c += 2) { const float2 t = *reinterpret_cast<const float2*>(p + c); x.v[c] = t.x; x.v[c + 1] = t.y; }
}
__device__ __forceinline__ float dot(const V10& a, const V10& b) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 2) { s0 = fmaf(a.v[c], b.v[c], s0); s1 = fmaf(a.v[c + 1], b.v[c + 1], s1); }
    return s0 + s1;
}
// mode bit0: MFMA waves (wave ids < NM) work; bit1: VALU waves (ids >= NM) work.  NT threads, NM matrix waves.
template <int NT, int NM>
__global__ void __launch_bounds__(NT) overlap(float* out, long long* cyc, int L, int iters, int mode) {
    extern __shared__ float sm[];
    float* qkv = sm;                 // [64][244]
    float* xs = qkv + ROWS * LDQ;    // [64][68]
    float* ob = xs + ROWS * LDX;     // [64][84]
    for (int e = threadIdx.x; e < ROWS * (LDQ + LDX + 84); e += NT) sm[e] = 0.01f * (float)((e * 2654435761u) >> 20) - 20.f;
    __syncthreads();
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, n = l & 15, g = l >> 4;
    long long t0 = clock64();
    float res = 0.f;
    if (w < NM) {
        if (mode & 1) {
            // QKV-like: per iteration 960 MFMAs spread over the NM matrix waves: tasks = 15 column tiles x 4 row tiles x 4 k-blocks
            float4 bq[4];
            for (int kb = 0; kb < 4; ++kb) bq[kb] = make_float4(0.1f * l, 0.2f, 0.3f * kb, 0.4f);
            for (int it = 0; it < iters; ++it) {
                for (int nt = w; nt < 15; nt += NM) {
                    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        float4 af[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const float4*>(xs + (16 * i + n) * LDX + 16 * kb + 4 * g);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bq[kb].x, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bq[kb].y, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bq[kb].z, acc[i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bq[kb].w, acc[i], 0, 0, 0);
                    }
                    res += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
                }
            }
        }
    } else if (mode & 2) {
        __builtin_amdgcn_s_setprio(3);
        const int nv = NT - NM * 64, tv = threadIdx.x - NM * 64;
        const int nsq = ROWS / L, ntasks = nsq * H * L;
        for (int it = 0; it < iters; ++it) {
            for (int task = tv; task < ntasks; task += nv) {
                const int i = task % L, h = (task / L) % H, sq = task / (L * H);
                const int row_i = sq * L + i;
                V10 q, kv, o;
                ld(q, qkv + row_i * LDQ + h * DH);
#pragma unroll
                for (int c = 0; c < DH; ++c) o.v[c] = 0.f;
                float m = -1e30f, lsum = 0.f;
                const float* kbase = qkv + (sq * L) * LDQ + I + h * DH;
                for (int j = 0; j < L; ++j) {
                    const float* kp = kbase + j * LDQ;
                    ld(kv, kp);
                    const float s = dot(q, kv) * 0.4f;
                    const float mn = fmaxf(m, s);
                    const float corr = __builtin_amdgcn_exp2f(m - mn), p = __builtin_amdgcn_exp2f(s - mn);
                    lsum = lsum * corr + p;
                    ld(kv, kp + I);
#pragma unroll
                    for (int c = 0; c < DH; ++c) o.v[c] = fmaf(p, kv.v[c], o.v[c] * corr);
                    m = mn;
                }
                float* op = ob + row_i * 84 + h * DH;
#pragma unroll
                for (int c = 0; c < DH; c += 2) *reinterpret_cast<float2*>(op + c) = make_float2(o.v[c] / lsum, o.v[c + 1] / lsum);
                res += o.v[0];
            }
        }
    }
    long long t1 = clock64();
    if (blockIdx.x == 0 && l == 0) cyc[w] = (t1 - t0) / iters;
    out[(blockIdx.x * NT + threadIdx.x) % 4096] = res;
}
template <int NT, int NM>
void run(float* out, long long* cyc, int L) {
    const size_t smem = (size_t)ROWS * (LDQ + LDX + 84) * 4;
    long long h[16];
    for (int mode = 1; mode <= 3; ++mode) {
        overlap<NT, NM><<<256, NT, smem>>>(out, cyc, L, 100, mode);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("NT=%d matrix waves=%d L=%d mode %d: matrix wave0 %lld cyc/iter (960 MFMA per iter in total), core wave%d %lld cyc/iter (one forward core pass)\n",
               NT, NM, L, mode, h[0], NM, h[NM]);
    }
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 16 * 8);
    run<512, 4>(out, cyc, 21);      // 4 matrix waves (1 per SIMD) + 4 core waves
    run<1024, 8>(out, cyc, 21);     // 8 + 8
    run<1024, 4>(out, cyc, 21);     // 4 matrix + 12 core
    return 0;
}
