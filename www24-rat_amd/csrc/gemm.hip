// gemm.hip — plain fp32 GEMM on v_mfma_f32_16x16x4_f32 for the prediction head's nn.Linear layers
// (MLP_Layer, fuxictr/pytorch/layers/deep.py:126-141): forward (x W^T + b), dgrad (dy W) and wgrad (dy^T x).
// The head is 1.4 % of the model's FLOPs.  64x64x32 LDS tiles, 4 waves each owning a 32x32 quadrant
// (2x2 MFMA tiles); the next k-tile is fetched into registers with 16-byte loads while the current one feeds the
// MFMAs (register-staged software pipeline: one barrier pair per 32-deep step).
#include "rat_device.h"
#include <stdlib.h>
#include "../../include/rat_hip.h"

namespace {

constexpr int GM_THREADS = 256;
constexpr int GM_TILE = 64;
constexpr int GM_K = 32;
constexpr int GM_LD = GM_K + 4;
constexpr int GM_VEC = GM_TILE * GM_K / 4 / GM_THREADS;      // float4 loads per thread per operand tile (= 2)

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K, lda, ldb, ldc;
    int ta, tb;
    int veca, vecb;       // 16-byte loads are legal for that operand (ld % 4 == 0, base 16-byte aligned)
    float beta;
    int slices, kper;     // split-K: the k range is cut into `slices` pieces of `kper` (multiple of GM_K); slice s of tile t is
    float* partial;       //          block t * slices + s and writes its raw tile to partial[s][M][N] (summed in fixed order later)
};

// One operand tile: T[r][k] (r = m or n index inside the tile, k inside the k-tile).  `kmajor` = the operand is stored
// with k as the slow index ([K][R], contiguous along r); otherwise [R][K], contiguous along k.
struct TileLoader {
    const float* base;
    int R, K, ld, r0;
    bool kmajor, vec;
    __device__ __forceinline__ void fetch(int k0, float4 (&v)[GM_VEC]) const {
#pragma unroll
        for (int u = 0; u < GM_VEC; ++u) {
            const int e = threadIdx.x + GM_THREADS * u;
            int r, k;
            if (kmajor) { k = e / (GM_TILE / 4); r = (e % (GM_TILE / 4)) * 4; }
            else        { r = e / (GM_K / 4);    k = (e % (GM_K / 4)) * 4; }
            const int gr = r0 + r, gk = k0 + k;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kmajor) {
                if (gk < K) {
                    const float* p = base + (size_t)gk * ld + gr;
                    if (vec && gr + 3 < R) t = *reinterpret_cast<const float4*>(p);
                    else {
                        if (gr + 0 < R) t.x = p[0];
                        if (gr + 1 < R) t.y = p[1];
                        if (gr + 2 < R) t.z = p[2];
                        if (gr + 3 < R) t.w = p[3];
                    }
                }
            } else if (gr < R) {
                const float* p = base + (size_t)gr * ld + gk;
                if (vec && gk + 3 < K) t = *reinterpret_cast<const float4*>(p);
                else {
                    if (gk + 0 < K) t.x = p[0];
                    if (gk + 1 < K) t.y = p[1];
                    if (gk + 2 < K) t.z = p[2];
                    if (gk + 3 < K) t.w = p[3];
                }
            }
            v[u] = t;
        }
    }
    __device__ __forceinline__ void stash(float* tile, const float4 (&v)[GM_VEC]) const {
#pragma unroll
        for (int u = 0; u < GM_VEC; ++u) {
            const int e = threadIdx.x + GM_THREADS * u;
            if (kmajor) {
                const int k = e / (GM_TILE / 4), r = (e % (GM_TILE / 4)) * 4;
                tile[(r + 0) * GM_LD + k] = v[u].x;
                tile[(r + 1) * GM_LD + k] = v[u].y;
                tile[(r + 2) * GM_LD + k] = v[u].z;
                tile[(r + 3) * GM_LD + k] = v[u].w;
            } else {
                const int r = e / (GM_K / 4), k = (e % (GM_K / 4)) * 4;
                *reinterpret_cast<float4*>(tile + r * GM_LD + k) = v[u];
            }
        }
    }
};

__device__ __forceinline__ void sgemm_epilogue(const GemmArgs& g, const f32x4 (&acc)[2][2], int m0, int n0, int wm, int wn, int slice) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + rat_acc_col(wn + j);
            if (g.slices > 1) {                            // raw partial tile; bias / beta are applied by the reduction
                if (col < g.N)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + rat_acc_row(wm + i, r);
                        if (row < g.M) g.partial[((size_t)slice * g.M + row) * g.N + col] = acc[i][j][r];
                    }
                continue;
            }
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

__global__ void __launch_bounds__(GM_THREADS) sgemm_kernel(GemmArgs g) {
    RAT_DYN_SMEM(smem);
    float* As = reinterpret_cast<float*>(smem);           // [64][GM_LD]  As[m][k] = op(A)[m0+m][k0+k]
    float* Bs = As + GM_TILE * GM_LD;                     // [64][GM_LD]  Bs[n][k] = op(B)[k0+k][n0+n]
    const int tiles_n = (g.N + GM_TILE - 1) / GM_TILE;
    const int tile = blockIdx.x / g.slices, slice = blockIdx.x - tile * g.slices;
    const int m0 = (tile / tiles_n) * GM_TILE;
    const int n0 = (tile % tiles_n) * GM_TILE;
    const int kbeg = slice * g.kper, kend = kbeg + g.kper < g.K ? kbeg + g.kper : g.K;
    const int wave = rat_wave();
    const int wm = (wave >> 1) * 2, wn = (wave & 1) * 2;  // first 16-row / 16-col tile of this wave's quadrant
    // op(A)[m][k]: ta=0 -> A[m*lda+k] (k contiguous); ta=1 -> A[k*lda+m] (k-major).  op(B)[k][n]: tb=1 -> B[n*ldb+k]; tb=0 -> B[k*ldb+n]
    const TileLoader la{g.A, g.M, g.K, g.lda, m0, g.ta != 0, g.veca != 0};
    const TileLoader lb{g.B, g.N, g.K, g.ldb, n0, g.tb == 0, g.vecb != 0};
    f32x4 acc[2][2];
    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = rat_zero4();
    const RatLdsRows Af{As, GM_LD};
    const RatLdsRows Bf{Bs, GM_LD};
    float4 ra[GM_VEC], rb[GM_VEC];
    la.fetch(kbeg, ra);
    lb.fetch(kbeg, rb);
    for (int k0 = kbeg; k0 < kend; k0 += GM_K) {
        la.stash(As, ra);
        lb.stash(Bs, rb);
        __syncthreads();
        if (k0 + GM_K < kend) {                            // next tile's loads fly while this one is multiplied
            la.fetch(k0 + GM_K, ra);
            lb.fetch(k0 + GM_K, rb);
        }
        rat_wave_gemm<2, 2>(acc, Af, Bf, wm, wn, 2, 2, GM_K / 16);
        __syncthreads();
    }
    sgemm_epilogue(g, acc, m0, n0, wm, wn, slice);
}

// C[m][n] = sum_s partial[s][m][n] (+ bias[n]) (+ beta * C[m][n]), slices summed in index order (deterministic)
__global__ void __launch_bounds__(256) sgemm_reduce_kernel(GemmArgs g) {
    const size_t total = (size_t)g.M * g.N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int row = (int)(e / g.N), col = (int)(e - (size_t)row * g.N);
        float v = 0.f;
        for (int s = 0; s < g.slices; ++s) v += g.partial[(size_t)s * total + e];
        if (g.bias) v += g.bias[col];
        float* c = g.C + (size_t)row * g.ldc + col;
        if (g.beta != 0.f) v += g.beta * (*c);
        *c = v;
    }
}

// ---- bf16x3 variant (round 3): the same 64 x 64 x 32 tiling on v_mfma_f32_16x16x32_bf16 with 3-way split operands (rat_device.h
// "bf16x3": six exact bf16 products per fp32 product, fp32 accumulation — the arithmetic of the encoder kernels).  Operand tiles are
// split ONCE, by the thread that fetched them, into LDS plane images:
//   * k contiguous in memory ([R][K]):  RatPlanes<80, 0>  [64 r][32 k] (+16 pad bytes per row: the 16 rows of a fragment read fall
//     on distinct banks), k stored in the slot order of col_frag so that both kinds of operand agree;   fragment = row_frag
//   * r contiguous in memory ([K][R]):  RatPlanes<128, 7> [32 k][64 r] in its natural orientation;      fragment = col_frag (two
//     transposed 4 x 16 block reads per plane, ds_read_b64_tr_b16) — no transposing scalar stores as in sgemm_kernel.
// One K = 32 step per tile pair: 4 x 6 MFMAs of 16 cycles per wave against 4 x 8 x 32 cycles on v_mfma_f32_16x16x4_f32.
typedef RatPlanes<80, 0, 64 * 80> GmPlanesR;           // [64 r][32 k]
typedef RatPlanes<128, 7, 32 * 128> GmPlanesK;         // [32 k][64 r]
constexpr int GM3_OPERAND = 3 * 64 * 80;               // bytes per operand (the larger of the two images)

template <bool KMAJOR>
struct Tile3 {
    char* base;
    // the thread's two float4 of TileLoader::fetch -> planes
    __device__ __forceinline__ void stash(const float4 (&v)[GM_VEC]) const {
        if (KMAJOR) {                                      // element e -> k = e / 16, r = 4 (e % 16): two fetches = rows k, k + 16
            const GmPlanesK pl{base};
#pragma unroll
            for (int u = 0; u < GM_VEC; ++u) {
                const int e = threadIdx.x + GM_THREADS * u;
                const int k = e / (GM_TILE / 4), q4 = e % (GM_TILE / 4);
                unsigned h0, h1, m0, m1, l0, l1;
                rat_split2(v[u].x, v[u].y, h0, m0, l0);
                rat_split2(v[u].z, v[u].w, h1, m1, l1);
                pl.store_half(k, q4, h0, h1, m0, m1, l0, l1);
            }
        } else {                                           // element e -> r = e / 8, k = 4 (e % 8): slot order of col_frag
            const GmPlanesR pl{base};
#pragma unroll
            for (int u = 0; u < GM_VEC; ++u) {
                const int e = threadIdx.x + GM_THREADS * u;
                const int r = e / (GM_K / 4), k = (e % (GM_K / 4)) * 4;
                unsigned h0, h1, m0, m1, l0, l1;
                rat_split2(v[u].x, v[u].y, h0, m0, l0);
                rat_split2(v[u].z, v[u].w, h1, m1, l1);
                pl.store_half(r, 2 * ((k & 15) >> 2) + (k >> 4), h0, h1, m0, m1, l0, l1);     // piece g = (k % 16) / 4, half = k >= 16
            }
        }
    }
    __device__ __forceinline__ RatB3 frag(int t) const {   // rows / columns 16 t .. 16 t + 15 of the tile, all 32 k
        if (KMAJOR) return GmPlanesK{base}.col_frag(t, 0);
        return GmPlanesR{base}.row_frag(t, 0);
    }
};

// KSUB: 32-wide k sub-tiles per trip.  KSUB = 2 (48 MFMAs per wave between a request and its use, half the barriers
// per k, 61 KB of LDS) was built in round 5 on the theory that the loads of the next sub-tile arrive late at the head's small grids —
// measured: no shape faster, the 1280-tile input-gradient product 0.047 -> 0.088 ms (occupancy), profiles/round5/r5_sgemm_ksub_ab.txt.
// The product runs KSUB = 1.
template <bool KA, bool KB, int KSUB = 1>
__global__ void __launch_bounds__(GM_THREADS) sgemm3_kernel(GemmArgs g) {
    RAT_DYN_SMEM(smem);
    const int tiles_n = (g.N + GM_TILE - 1) / GM_TILE;
    const int tile = blockIdx.x / g.slices, slice = blockIdx.x - tile * g.slices;
    const int m0 = (tile / tiles_n) * GM_TILE;
    const int n0 = (tile % tiles_n) * GM_TILE;
    const int kbeg = slice * g.kper, kend = kbeg + g.kper < g.K ? kbeg + g.kper : g.K;
    const int wave = rat_wave();
    const int wm = (wave >> 1) * 2, wn = (wave & 1) * 2;
    const TileLoader la{g.A, g.M, kend, g.lda, m0, KA, g.veca != 0};       // (K = this slice's end: a sub-tile past it reads zeros)
    const TileLoader lb{g.B, g.N, kend, g.ldb, n0, KB, g.vecb != 0};
    f32x4 acc[2][2];
    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = rat_zero4();
    float4 ra[KSUB][GM_VEC], rb[KSUB][GM_VEC];
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
        la.fetch(kbeg + s * GM_K, ra[s]);
        lb.fetch(kbeg + s * GM_K, rb[s]);
    }
    for (int k0 = kbeg; k0 < kend; k0 += KSUB * GM_K) {
#pragma unroll
        for (int s = 0; s < KSUB; ++s) {
            Tile3<KA>{smem + (size_t)s * 2 * GM3_OPERAND}.stash(ra[s]);
            Tile3<KB>{smem + (size_t)s * 2 * GM3_OPERAND + GM3_OPERAND}.stash(rb[s]);
        }
        __syncthreads();
        if (k0 + KSUB * GM_K < kend) {                     // next trip's loads fly while this one is multiplied
#pragma unroll
            for (int s = 0; s < KSUB; ++s) {
                la.fetch(k0 + (KSUB + s) * GM_K, ra[s]);
                lb.fetch(k0 + (KSUB + s) * GM_K, rb[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < KSUB; ++s) {
            const Tile3<KA> At{smem + (size_t)s * 2 * GM3_OPERAND};
            const Tile3<KB> Bt{smem + (size_t)s * 2 * GM3_OPERAND + GM3_OPERAND};
            const RatB3 a0 = At.frag(wm), a1 = At.frag(wm + 1), b0 = Bt.frag(wn), b1 = Bt.frag(wn + 1);
            {
                f32x4 c[2] = {acc[0][0], acc[1][0]};
                const RatB3 aa[2] = {a0, a1};
                rat_mfma3_block<2>(c, aa, b0);
                acc[0][0] = c[0];
                acc[1][0] = c[1];
            }
            {
                f32x4 c[2] = {acc[0][1], acc[1][1]};
                const RatB3 aa[2] = {a0, a1};
                rat_mfma3_block<2>(c, aa, b1);
                acc[0][1] = c[0];
                acc[1][1] = c[1];
            }
        }
        __syncthreads();
    }
    sgemm_epilogue(g, acc, m0, n0, wm, wn, slice);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// split-K plan: enough work-groups to cover the HBM latency of the k-loop by occupancy (the weight-gradient GEMMs of the
// head have 7..140 output tiles and k = batch = 4096)
// `target` work-groups: 1024 for the weight gradients (k = batch: long, few tiles), 512 for the forward / input-gradient products
// (tools/kbench.py sgemm, MI355X: at 448 tiles two slices beat three — 32 against 38 us at K = 400, 70 against 86 us at K = 1280 —
// while the weight gradients lose 20-40 % with fewer slices); RAT_SGEMM_SPLIT_TARGET overrides both (A/B knob)
void plan_split(int M, int N, int K, int& slices, int& kper, int target = 1024) {
    const int tiles = ((M + GM_TILE - 1) / GM_TILE) * ((N + GM_TILE - 1) / GM_TILE);
    slices = 1;
    kper = (K + GM_K - 1) / GM_K * GM_K;
    if (tiles >= 512 || K < 8 * GM_K) return;
    const int forced = rat_knob(RAT_KNOB_SGEMM_SPLIT_TARGET);
    if (forced) target = forced;
    int want = (target + tiles - 1) / tiles;
    const int max_slices = K / (2 * GM_K);
    if (want > max_slices) want = max_slices;
    if (want <= 1) return;
    kper = ((K + want - 1) / want + GM_K - 1) / GM_K * GM_K;
    slices = (K + kper - 1) / kper;
}

}  // namespace

extern "C" size_t rat_sgemm_workspace(int M, int N, int K) {
    int slices, kper;
    plan_split(M, N, K, slices, kper);
    return slices > 1 ? (size_t)slices * M * N * sizeof(float) : 0;
}

static int sgemm_launch(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                        float* C, int ldc, const float* bias, float beta, float* workspace, size_t workspace_bytes, void* stream,
                        int arith = RAT_ARITH_F32) {
    RAT_REQUIRE(M > 0 && N > 0 && K > 0, "bad dims");
    RAT_REQUIRE(A && B && C, "null pointer");
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, 0, 0, beta, 1, 0, nullptr};
    g.veca = (lda % 4 == 0) && aligned16(A);
    g.vecb = (ldb % 4 == 0) && aligned16(B);
    plan_split(M, N, K, g.slices, g.kper, trans_a ? 1024 : 512);       // (never more slices than rat_sgemm_workspace's plan: its target is 1024)
    if (g.slices > 1 && (workspace == nullptr || workspace_bytes < (size_t)g.slices * M * N * sizeof(float))) {
        g.slices = 1;                                       // no (or too small a) workspace: single pass over k
        g.kper = (K + GM_K - 1) / GM_K * GM_K;
    }
    g.partial = workspace;
    const int tiles = ((M + GM_TILE - 1) / GM_TILE) * ((N + GM_TILE - 1) / GM_TILE);
    // bf16x3: 16-byte fetches on both operands and a contraction long enough to matter; everything else runs exact fp32
    if (arith == RAT_ARITH_BF16X3 && g.veca && g.vecb && K >= 2 * GM_K && M >= 16 && N >= 16) {
        const size_t smem3 = (size_t)2 * GM3_OPERAND;
        const bool ka = g.ta != 0, kb = g.tb == 0;
        if (ka && kb) RAT_LAUNCH((sgemm3_kernel<true, true>), tiles * g.slices, GM_THREADS, smem3, stream, g);
        else if (ka) RAT_LAUNCH((sgemm3_kernel<true, false>), tiles * g.slices, GM_THREADS, smem3, stream, g);
        else if (kb) RAT_LAUNCH((sgemm3_kernel<false, true>), tiles * g.slices, GM_THREADS, smem3, stream, g);
        else RAT_LAUNCH((sgemm3_kernel<false, false>), tiles * g.slices, GM_THREADS, smem3, stream, g);
    } else {
        RAT_LAUNCH(sgemm_kernel, tiles * g.slices, GM_THREADS, (size_t)2 * GM_TILE * GM_LD * sizeof(float), stream, g);
    }
    if (rat_check_launch("rat_sgemm")) return -1;
    if (g.slices > 1) {
        const size_t total = (size_t)M * N;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        RAT_LAUNCH(sgemm_reduce_kernel, blocks, 256, 0, stream, g);
        return rat_check_launch("rat_sgemm (split-K reduction)");
    }
    return 0;
}

extern "C" int rat_sgemm(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                         float* C, int ldc, const float* bias, float beta, void* stream) {
    return sgemm_launch(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, beta, nullptr, 0, stream);
}

extern "C" int rat_sgemm_ws(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                            float* C, int ldc, const float* bias, float beta, float* workspace, size_t workspace_bytes,
                            void* stream) {
    return sgemm_launch(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, beta, workspace, workspace_bytes, stream);
}

extern "C" int rat_sgemm_arith(int trans_a, int trans_b, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                               float* C, int ldc, const float* bias, float beta, float* workspace, size_t workspace_bytes,
                               int arith, void* stream) {
    return sgemm_launch(trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, beta, workspace, workspace_bytes, stream, arith);
}
