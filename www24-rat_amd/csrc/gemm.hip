// gemm.hip — plain fp32 GEMM on v_mfma_f32_16x16x4_f32 for the prediction head's nn.Linear layers
// (MLP_Layer, fuxictr/pytorch/layers/deep.py:126-141): forward (x W^T + b), dgrad (dy W) and wgrad (dy^T x).
// The head is 1.4 % of the model's FLOPs; this kernel is a straightforward 64x64x16 LDS-tiled
// design: 4 waves, each owning a 32x32 quadrant as 2x2 MFMA tiles.
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int GM_THREADS = 256;
constexpr int GM_TILE = 64;
constexpr int GM_K = 16;
constexpr int GM_LD = GM_K + 4;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K, lda, ldb, ldc;
    int ta, tb;
    float beta;
};

__global__ void __launch_bounds__(GM_THREADS) sgemm_kernel(GemmArgs g) {
    RAT_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);           // [64][GM_LD]  As[m][k] = op(A)[m0+m][k0+k]
    float* Bs = As + GM_TILE * GM_LD;                     // [64][GM_LD]  Bs[n][k] = op(B)[k0+k][n0+n]
    const int tiles_n = (g.N + GM_TILE - 1) / GM_TILE;
    const int m0 = (blockIdx.x / tiles_n) * GM_TILE;
    const int n0 = (blockIdx.x % tiles_n) * GM_TILE;
    const int wave = rat_wave();
    const int wm = (wave >> 1) * 2, wn = (wave & 1) * 2;  // first 16-row / 16-col tile of this wave's quadrant
    f32x4 acc[2][2];
    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = rat_zero4();
    const RatLdsRows Af{As, GM_LD};
    const RatLdsRows Bf{Bs, GM_LD};
    for (int k0 = 0; k0 < g.K; k0 += GM_K) {
        for (int e = threadIdx.x; e < GM_TILE * GM_K; e += GM_THREADS) {
            int m, k;
            if (g.ta) { m = e % GM_TILE; k = e / GM_TILE; } else { m = e / GM_K; k = e % GM_K; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < g.K) v = g.ta ? g.A[(size_t)gk * g.lda + gm] : g.A[(size_t)gm * g.lda + gk];
            As[m * GM_LD + k] = v;
            int n, kk;
            if (g.tb) { n = e / GM_K; kk = e % GM_K; } else { n = e % GM_TILE; kk = e / GM_TILE; }
            const int gn = n0 + n, gk2 = k0 + kk;
            float w = 0.f;
            if (gn < g.N && gk2 < g.K) w = g.tb ? g.B[(size_t)gn * g.ldb + gk2] : g.B[(size_t)gk2 * g.ldb + gn];
            Bs[n * GM_LD + kk] = w;
        }
        __syncthreads();
        rat_wave_gemm<2, 2>(acc, Af, Bf, wm, wn, 2, 2, 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + rat_acc_col(wn + j);
            if (col < g.N) {
                const float b = g.bias ? g.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + rat_acc_row(wm + i, r);
                    if (row < g.M) {
                        float v = acc[i][j][r] + b;
                        float* c = g.C + (size_t)row * g.ldc + col;
                        if (g.beta != 0.f) v += g.beta * (*c);
                        *c = v;
                    }
                }
            }
        }
}

}  // namespace

extern "C" int rat_sgemm(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, float beta, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && K > 0, "bad dims");
    RAT_REQUIRE(A && B && C, "null pointer");
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, beta};
    const int tiles = ((M + GM_TILE - 1) / GM_TILE) * ((N + GM_TILE - 1) / GM_TILE);
    RAT_LAUNCH(sgemm_kernel, tiles, GM_THREADS, (size_t)2 * GM_TILE * GM_LD * sizeof(float), stream, g);
    return rat_check_launch("rat_sgemm");
}
