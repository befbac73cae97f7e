// ffn.hip — K2b: the block MLP of CrossIntraEncoderBlock with its residual, forward and backward.
//
//   y = W2 gelu_erf(W1 x + b1) + b2 + x        (RAT_m2.py:163-174 FeedForward, RAT_m2.py:232; no LayerNorm in front)
//
// Token-wise, so the grid is treated as a flat [ntok][d] matrix.  A 512-thread work-group owns 64 tokens:
// x tile and the hidden tile live in LDS, both GEMMs run on v_mfma_f32_16x16x4_f32 with the weights
// streamed from L2 as B operands.  Backward recomputes the hidden activations, and keeps dW1 / dW2 in MFMA
// accumulators across the work-group's chunk loop (per-work-group slab + fixed-order reduction).
#include "rat_device.h"
#include "../../include/rat_hip.h"

#include <initializer_list>
#include <stdlib.h>

namespace {

constexpr int FFN_THREADS = 512;
constexpr int FFN_WAVES = FFN_THREADS / 64;
constexpr int FFN_ROWS = 64;
constexpr int FFN_MT = FFN_ROWS / 16;
constexpr int WSLOTS = 8;          // persistent tiles per wave for each of dW1, dW2  (H16/16 * D16/16 <= 64)

struct FfnArgs {
    const float* x;
    const float* dy;
    float* y;            // forward output / backward dx
    const float* res;    // forward: residual source (== x for the RAT_m2 block MLP; nullptr: none)
    int add_dy;          // backward: dx = dy W-chain (+ dy when the residual came from x itself)
    const float* w1;     // [H][D]
    const float* b1;
    const float* w2;     // [D][H]
    const float* b2;
    const float* w1t;    // backward fast path: [D][H] = w1^T, [H][D] = w2^T (16-byte B-fragment loads for the two dgrad GEMMs)
    const float* w2t;
    float* slabs;
    int64_t slab_stride;
    int64_t ntok, nchunks;
    int d, hidden;
    int vec_x, vec_w1, vec_w2;
    unsigned long long* prof;
    int dy_period;       // ffn_bwd_t4 only: 0 = dy is [ntok][d]; P > 0 = only tokens t with t % P == 0 have a gradient row, stored compactly at
                         // dy[(t / P)][d] — every other row of dy is zero by contract and is neither stored nor read (the last encoder block:
                         // the head reads the class token of each sample only, RAT_m2.py:138-140)
    RatDrop drop1, drop2;  // FeedForward's two nn.Dropout (RAT_m1.py:151-161, RAT_m0.py:150-160): behind GELU (index token * H + unit) and
                           // behind the second Linear, in front of the residual (index token * D + column); generic kernels only
};

struct FfnGeom {
    int D, H, D16, H16, ldx, ldh;
    __host__ __device__ FfnGeom(int d, int hidden) {
        D = d;
        H = hidden;
        D16 = (D + 15) / 16 * 16;
        H16 = (H + 15) / 16 * 16;
        ldx = D16 + 4;
        ldh = H16 + 4;
    }
    size_t fwd_smem() const { return (size_t)FFN_ROWS * (2 * ldx + ldh) * 4; }
    size_t bwd_smem() const { return (size_t)FFN_ROWS * (2 * ldx + 2 * ldh) * 4; }
    int64_t slab_floats() const { return 2 * (int64_t)H * D + H + D; }
};

__device__ __forceinline__ void ffn_load(float* tile, int ld, const float* src, int64_t tok0, int rows, int width, bool vec) {
    if (vec) {
        const int w4 = width >> 2;
        for (int e = threadIdx.x; e < FFN_ROWS * w4; e += FFN_THREADS) {
            const int r = e / w4, c4 = e - r * w4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows) v = *reinterpret_cast<const float4*>(src + (tok0 + r) * width + 4 * c4);
            *reinterpret_cast<float4*>(tile + (size_t)r * ld + 4 * c4) = v;
        }
    } else {
        for (int e = threadIdx.x; e < FFN_ROWS * width; e += FFN_THREADS) {
            const int r = e / width, c = e - r * width;
            tile[(size_t)r * ld + c] = r < rows ? src[(tok0 + r) * width + c] : 0.f;
        }
    }
}

__device__ __forceinline__ void ffn_store(float* dst, const float* tile, int ld, int64_t tok0, int rows, int width, bool vec) {
    if (vec) {
        const int w4 = width >> 2;
        for (int e = threadIdx.x; e < rows * w4; e += FFN_THREADS) {
            const int r = e / w4, c4 = e - r * w4;
            *reinterpret_cast<float4*>(dst + (tok0 + r) * width + 4 * c4) = *reinterpret_cast<const float4*>(tile + (size_t)r * ld + 4 * c4);
        }
    } else {
        for (int e = threadIdx.x; e < rows * width; e += FFN_THREADS) {
            const int r = e / width, c = e - r * width;
            dst[(tok0 + r) * width + c] = tile[(size_t)r * ld + c];
        }
    }
}

__device__ __forceinline__ void ffn_zero_cols(float* tile, int ld, int c0) {
    const int w = ld - c0;
    for (int e = threadIdx.x; e < FFN_ROWS * w; e += FFN_THREADS) tile[(size_t)(e / w) * ld + c0 + e % w] = 0.f;
}

// hs[rows][0:H] = xs W1^T + b1 ; optionally gs = gelu(hs)   (MODE 0: hs <- gelu(h) only; MODE 1: hs <- gelu'(h), gs <- gelu(h))
// MT: row tiles per (wave) task — FFN_MT = one task per column tile (the wide shapes); 1 = one task per 16 x 16 tile, for hidden widths of
// two or three column tiles, where FFN_MT leaves most waves without a task and two or three waves with all the GELU evaluations
template <int TD, int MODE, int MT, class BW>
__device__ __forceinline__ void ffn_hidden_on(const FfnArgs& a, const FfnGeom& g, const float* xs, float* hs, float* gs,
                                              int mt_valid, int rows, int64_t tok0, const BW& Bw) {
    constexpr bool FAST = TD > 0;
    const RatLdsRows A{xs, g.ldx};
    rat_gemm_phase<FAST, MT, FFN_WAVES, FFN_MT, (FAST ? TD / 16 : 0)>(A, Bw, mt_valid, g.H16 / 16, g.D16 / 16, [&](int mt, int nt, const f32x4& acc) {
        const int col = rat_acc_col(nt);
        if (FAST || col < g.H) {
            const float bias = a.b1[col];
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = rat_acc_row(mt, r) < rows ? acc[r] + bias : 0.f;    // padding rows stay exactly 0
#pragma unroll
            for (int r = 0; r < 4; r += 2) {                      // two activations per packed-math evaluation
                const size_t o0 = (size_t)rat_acc_row(mt, r) * g.ldh + col, o1 = (size_t)rat_acc_row(mt, r + 1) * g.ldh + col;
                // Dropout behind GELU (drop1; threshold 0 = none): g <- keep ? g / (1 - p) : 0, and the same factor on gelu'(h)
                const int64_t i0 = (tok0 + rat_acc_row(mt, r)) * g.H + col, i1 = (tok0 + rat_acc_row(mt, r + 1)) * g.H + col;
                if (MODE == 0) {
                    const rat_f2 gv = rat_gelu2(rat_f2_make(h[r], h[r + 1]));
                    hs[o0] = a.drop1.apply(gv.x, i0);
                    hs[o1] = a.drop1.apply(gv.y, i1);
                } else {                                         // backward: hs <- gelu'(h) (h itself is not needed again);
                    float g0, d0, g1, d1;                        // scalar form: the packed one measured 3 % slower here
                    rat_gelu_both(h[r], g0, d0);
                    rat_gelu_both(h[r + 1], g1, d1);
                    hs[o0] = a.drop1.apply(d0, i0);
                    hs[o1] = a.drop1.apply(d1, i1);
                    gs[o0] = a.drop1.apply(g0, i0);
                    gs[o1] = a.drop1.apply(g1, i1);
                }
            }
        }
    });
}

template <int TD, int MODE>
__device__ __forceinline__ void ffn_hidden(const FfnArgs& a, const FfnGeom& g, const float* xs, float* hs, float* gs,
                                           int mt_valid, int rows, int64_t tok0 = 0) {
    const RatGlobalWnkT<!(TD > 0)> Bw{a.w1, g.H, g.D, g.D, a.vec_w1 != 0};
    ffn_hidden_on<TD, MODE, FFN_MT>(a, g, xs, hs, gs, mt_valid, rows, tok0, Bw);
}

// ---- the shipped d = 10 geometries (MovieLens hidden 40, Tmall hidden 20), round 5.  A chunk of these kernels is a chain of four or five
// phases of a few MFMAs each; what they waited for was (a) their weight fragments, fetched from L2 per 16 x 16 tile, (b) run-time load loops
// (load -> wait -> LDS store per trip) and (c) two or three waves doing a phase's whole epilogue while the others idle.  SMALL = d 10:
//   * both weight matrices are copied to LDS once per work-group: w1s [H16][12] (row = hidden unit, the 10 weights of that unit; the fourth
//     k-quad of a fragment reads the next row's first floats — finite, and multiplied by the zero padding columns of the activation tile),
//     w2s [16][H16 + 4] (row = output feature, rows >= d zero);
//   * a full chunk's 64 rows x 10 floats are contiguous: 160 aligned 16-byte loads (x and dy in one sweep) instead of 640 scalar ones in loops;
//   * one GEMM task per 16 x 16 tile (MT = 1): every wave gets a task.
constexpr int FFN_SMALL_LD1 = 12;
constexpr int ffn_small_w_floats(int hidden) { return ((hidden + 15) / 16 * 16) * FFN_SMALL_LD1 + 16 + 16 * ((hidden + 15) / 16 * 16 + 4); }
__device__ __forceinline__ void ffn_small_stage_weights(const FfnArgs& a, int D, int H, int H16, float* w1s, float* w2s) {
    const int total = ffn_small_w_floats(H);
    for (int e = threadIdx.x; e < total; e += FFN_THREADS) w1s[e] = 0.f;          // (w2s follows w1s)
    __syncthreads();
    for (int e = threadIdx.x; e < H * D; e += FFN_THREADS) w1s[(e / D) * FFN_SMALL_LD1 + e % D] = a.w1[e];
    for (int e = threadIdx.x; e < D * H; e += FFN_THREADS) w2s[(e / H) * (H16 + 4) + e % H] = a.w2[e];
}
// the chunk's contiguous 64 x 10 floats as 160 16-byte pieces (threads < 160); piece e covers flat elements 4 e ... 4 e + 3
__device__ __forceinline__ float4 ffn_small_piece(const float* src, int64_t tok0) {
    return (int)threadIdx.x < FFN_ROWS * 10 / 4 ? *reinterpret_cast<const float4*>(src + tok0 * 10 + 4 * threadIdx.x) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void ffn_small_stash(float* tile, int ld, const float4& v) {
    if ((int)threadIdx.x < FFN_ROWS * 10 / 4) {
        const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 4 * threadIdx.x + j, r = f / 10;
            tile[(size_t)r * ld + (f - 10 * r)] = e4[j];
        }
    }
}
__device__ __forceinline__ bool ffn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// GD / GHID: embedding_dim / hidden width of a generic (TD = 0) instantiation as compile-time constants (0 = run-time): the shipped
// d = 10 configs (MovieLens hidden 40, Tmall hidden 20) — every LDS stride and loop bound folds, the SGPR spills of the run-time form go
template <int TD, int GD = 0, int GHID = 0>
__global__ void __launch_bounds__(FFN_THREADS) ffn_fwd_kernel(FfnArgs a) {
    constexpr bool FAST = TD > 0;
    RAT_DYN_SMEM(smem);
    const FfnGeom g(FAST ? TD : (GD > 0 ? GD : a.d), GHID > 0 ? GHID : a.hidden);
    float* xs = reinterpret_cast<float*>(smem);
    float* hs = xs + (size_t)FFN_ROWS * g.ldx;
    float* ys = hs + (size_t)FFN_ROWS * g.ldh;                  // [64][ldx] output staging (whole-row coalesced stores)
    constexpr bool SMALL = !FAST && GD == 10 && GHID > 0 && GHID <= 64;
    float* const w1s = ys + (size_t)FFN_ROWS * g.ldx;           // SMALL: the two weight matrices (ffn_small_stage_weights)
    float* const w2s = w1s + g.H16 * FFN_SMALL_LD1 + 16;
    if (SMALL) ffn_small_stage_weights(a, g.D, g.H, g.H16, w1s, w2s);
    ffn_zero_cols(xs, g.ldx, g.D);
    ffn_zero_cols(hs, g.ldh, g.H);
    __syncthreads();
    RAT_PROF_DECL
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x) {
        const int64_t tok0 = chunk * FFN_ROWS;
        const int rows = a.ntok - tok0 < FFN_ROWS ? (int)(a.ntok - tok0) : FFN_ROWS;
        const int mt_valid = (rows + 15) / 16;
        if (SMALL && rows == FFN_ROWS && ffn_aligned16(a.x)) ffn_small_stash(xs, g.ldx, ffn_small_piece(a.x, tok0));
        else ffn_load(xs, g.ldx, a.x, tok0, rows, g.D, FAST || a.vec_x != 0);
        __syncthreads();
        RAT_PROF_MARK(0);
        if (SMALL) ffn_hidden_on<TD, 0, 1>(a, g, xs, hs, nullptr, mt_valid, rows, tok0, RatLdsRows{w1s, FFN_SMALL_LD1});
        else ffn_hidden<TD, 0>(a, g, xs, hs, nullptr, mt_valid, rows, tok0);
        __syncthreads();
        float pf = 0.f;
        {   // touch one dword per 128-byte line of the NEXT chunk's x rows: they travel HBM -> L2 behind the second GEMM
            const int64_t nt0 = (chunk + gridDim.x) * FFN_ROWS;
            const int64_t e = nt0 * g.D + (int64_t)threadIdx.x * 32;
            if (e < a.ntok * g.D && threadIdx.x * 32 < FFN_ROWS * g.D) pf = a.x[e];
        }
        RAT_PROF_MARK(1);
        // y = gelu(h) W2^T + b2 + x
        {
            const RatLdsRows A{hs, g.ldh};
            auto y_epi = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
                if (FAST || col < g.D) {
                    const float bias = a.b2[col];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rat_acc_row(mt, r);
                        const size_t o = (size_t)row * g.ldx + col;
                        float rv = 0.f;
                        if (a.res == a.x) rv = xs[o];
                        else if (a.res != nullptr && row < rows) rv = a.res[(tok0 + row) * g.D + col];
                        ys[o] = a.drop2.apply(acc[r] + bias, (tok0 + row) * g.D + col) + rv;
                    }
                }
            };
            if (SMALL) rat_gemm_phase<false, 1, FFN_WAVES, FFN_MT, 0>(A, RatLdsRows{w2s, g.H16 + 4}, mt_valid, g.D16 / 16, g.H16 / 16, y_epi);
            else rat_gemm_phase<FAST, 2, FFN_WAVES, FFN_MT, 0>(A, RatGlobalWnkT<!FAST>{a.w2, g.D, g.H, g.H, a.vec_w2 != 0}, mt_valid, g.D16 / 16, g.H16 / 16, y_epi);
        }
        __syncthreads();
        ffn_store(a.y, ys, g.ldx, tok0, rows, g.D, FAST || a.vec_x != 0);
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));
#endif
        RAT_PROF_MARK(2);
    }
    RAT_PROF_FLUSH(a.prof, 24);
}

// generic shapes (any d, hidden up to the LDS tile): LDS-staged, weights streamed from L2 with guarded fragment loads.  The
// compiled fast shapes (d, 2d) = (64, 128), (16, 32) run ffn_bwd_t_kernel below.
template <int GD = 0, int GHID = 0>
__global__ void __launch_bounds__(FFN_THREADS) ffn_bwd_kernel(FfnArgs a) {
    constexpr bool FAST = false;
    constexpr int TD = 0;
    RAT_DYN_SMEM(smem);
    const FfnGeom g(GD > 0 ? GD : a.d, GHID > 0 ? GHID : a.hidden);
    const int D = g.D, H = g.H;
    float* xs = reinterpret_cast<float*>(smem);                 // [64][ldx] x
    float* dys = xs + (size_t)FFN_ROWS * g.ldx;                 // [64][ldx] dL/dy
    float* hs = dys + (size_t)FFN_ROWS * g.ldx;                 // [64][ldh] h = W1 x + b1
    float* gs = hs + (size_t)FFN_ROWS * g.ldh;                  // [64][ldh] gelu(h), later dh
    constexpr bool SMALL = GD == 10 && GHID > 0 && GHID <= 64;
    float* const w1s = gs + (size_t)FFN_ROWS * g.ldh;           // SMALL: the two weight matrices (ffn_small_stage_weights)
    float* const w2s = w1s + g.H16 * FFN_SMALL_LD1 + 16;
    if (SMALL) ffn_small_stage_weights(a, g.D, g.H, g.H16, w1s, w2s);

    f32x4 acc1[WSLOTS], acc2[WSLOTS];                           // dW1 tiles (H16/16 x D16/16), dW2 tiles (D16/16 x H16/16)
#pragma unroll
    for (int s = 0; s < WSLOTS; ++s) acc1[s] = acc2[s] = rat_zero4();
    float db1 = 0.f, db2 = 0.f;                                 // thread c owns column c (c < H resp. c < D)
    const int t1n = g.D16 / 16, t1 = (g.H16 / 16) * t1n;
    const int t2n = g.H16 / 16, t2 = (g.D16 / 16) * t2n;

    ffn_zero_cols(xs, g.ldx, D);
    ffn_zero_cols(dys, g.ldx, D);
    ffn_zero_cols(hs, g.ldh, H);
    ffn_zero_cols(gs, g.ldh, H);
    __syncthreads();
    RAT_PROF_DECL

    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x) {
        const int64_t tok0 = chunk * FFN_ROWS;
        const int rows = a.ntok - tok0 < FFN_ROWS ? (int)(a.ntok - tok0) : FFN_ROWS;
        const int mt_valid = (rows + 15) / 16;
        if (SMALL && rows == FFN_ROWS && ffn_aligned16(a.x) && ffn_aligned16(a.dy)) {
            const float4 vx = ffn_small_piece(a.x, tok0), vd = ffn_small_piece(a.dy, tok0);
            ffn_small_stash(xs, g.ldx, vx);
            ffn_small_stash(dys, g.ldx, vd);
        } else {
            ffn_load(xs, g.ldx, a.x, tok0, rows, D, FAST || a.vec_x != 0);
            ffn_load(dys, g.ldx, a.dy, tok0, rows, D, FAST || a.vec_x != 0);
        }
        if (a.drop2.threshold != 0) {                           // dy through the output Dropout: every product below takes the masked dy
            __syncthreads();
            for (int e = threadIdx.x; e < rows * D; e += FFN_THREADS) {
                const int r = e / D, c = e - r * D;
                dys[(size_t)r * g.ldx + c] = a.drop2.apply(dys[(size_t)r * g.ldx + c], (tok0 + r) * D + c);
            }
        }
        __syncthreads();
        RAT_PROF_MARK(0);
        if (SMALL) ffn_hidden_on<TD, 1, 1>(a, g, xs, hs, gs, mt_valid, rows, tok0, RatLdsRows{w1s, FFN_SMALL_LD1});
        else ffn_hidden<TD, 1>(a, g, xs, hs, gs, mt_valid, rows, tok0);
        __syncthreads();
        RAT_PROF_MARK(1);
        // dW2 += dy^T gelu(h) ; db2 += colsum(dy)
        {
            const RatLdsCols At{dys, g.ldx};
            const RatLdsCols Bt{gs, g.ldh};
            rat_wave_gemm_slots<WSLOTS, FFN_WAVES, 0>(acc2, At, Bt, t2, t2n, mt_valid);
            {   // db2 partials: thread = (column, row group), combined after the chunk loop
                const int nrg = FFN_THREADS / D, col = threadIdx.x % D, rg = threadIdx.x / D;
                if (rg < nrg)
                    for (int r = rg; r < rows; r += nrg) db2 += dys[(size_t)r * g.ldx + col];
            }
        }
        __syncthreads();
        float pf = 0.f;
        {   // next chunk's x and dy lines -> L2
            const int64_t nt0 = (chunk + gridDim.x) * FFN_ROWS;
            const int per = FFN_ROWS * D / 32;                   // 128-byte lines per operand tile
            const int t = threadIdx.x;
            const float* src = t < per ? a.x : a.dy;
            const int64_t e = nt0 * D + (int64_t)(t < per ? t : t - per) * 32;
            if (t < 2 * per && e < a.ntok * D) pf = src[e];
        }
        RAT_PROF_MARK(2);
        // dh = (dy W2) * gelu'(h)  -> gs
        {
            const RatLdsRows A{dys, g.ldx};
            auto dh_epi = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
                if (FAST || col < H)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rat_acc_row(mt, r);
                        const size_t o = (size_t)row * g.ldh + col;
                        gs[o] = row < rows ? acc[r] * hs[o] : 0.f;          // hs holds gelu'(h) since the recomputation
                    }
            };
            if (FAST) {
                const RatGlobalWnkT<false> Bw{a.w2t, H, D, D, true};
                rat_gemm_phase<true, FFN_MT, FFN_WAVES, FFN_MT, (FAST ? TD / 16 : 0)>(A, Bw, mt_valid, g.H16 / 16, g.D16 / 16, dh_epi);
            } else if (SMALL) {                                  // B[k][n] = W2[k][n] from the LDS copy (rows k >= d, columns n >= H are zero)
                rat_gemm_phase<false, 1, FFN_WAVES, FFN_MT, 0>(A, RatLdsCols{w2s, g.H16 + 4}, mt_valid, g.H16 / 16, g.D16 / 16, dh_epi);
            } else {
                const RatGlobalWknT<true> Bw{a.w2, D, H, H};
                rat_gemm_phase<false, FFN_MT, FFN_WAVES, FFN_MT, 0>(A, Bw, mt_valid, g.H16 / 16, g.D16 / 16, dh_epi);
            }
        }
        __syncthreads();
        RAT_PROF_MARK(3);
        // dx = dh W1 + dy ; dW1 += dh^T x ; db1 += colsum(dh)
        {
            const RatLdsRows A{gs, g.ldh};
            auto dx_epi = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
                if (FAST || col < D)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                                                   // dx tile, in place
                        const int row = rat_acc_row(mt, r);
                        const size_t o = (size_t)row * g.ldx + col;
                        float rsd = a.add_dy ? dys[o] : 0.f;
                        if (a.add_dy && a.drop2.threshold != 0) rsd = row < rows ? a.dy[(tok0 + row) * D + col] : 0.f;   // the residual takes the UNmasked dy
                        dys[o] = rsd + acc[r];
                    }
            };
            if (FAST) {
                const RatGlobalWnkT<false> Bw{a.w1t, D, H, H, true};
                rat_gemm_phase<true, 2, FFN_WAVES, FFN_MT, 0>(A, Bw, mt_valid, g.D16 / 16, g.H16 / 16, dx_epi);
            } else if (SMALL) {                                  // B[k][n] = W1[k][n] (columns n >= d of the tile: finite, dropped by the epilogue)
                rat_gemm_phase<false, 1, FFN_WAVES, FFN_MT, 0>(A, RatLdsCols{w1s, FFN_SMALL_LD1}, mt_valid, g.D16 / 16, g.H16 / 16, dx_epi);
            } else {
                const RatGlobalWknT<true> Bw{a.w1, H, D, D};
                rat_gemm_phase<false, 2, FFN_WAVES, FFN_MT, 0>(A, Bw, mt_valid, g.D16 / 16, g.H16 / 16, dx_epi);
            }
            const RatLdsCols At{gs, g.ldh};
            const RatLdsCols Bt{xs, g.ldx};
            rat_wave_gemm_slots<WSLOTS, FFN_WAVES, 0>(acc1, At, Bt, t1, t1n, mt_valid);
            {
                const int nrg = FFN_THREADS / H, col = threadIdx.x % H, rg = threadIdx.x / H;
                if (rg < nrg)
                    for (int r = rg; r < rows; r += nrg) db1 += gs[(size_t)r * g.ldh + col];
            }
        }
        __syncthreads();
        ffn_store(a.y, dys, g.ldx, tok0, rows, D, FAST || a.vec_x != 0);
        __syncthreads();
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));
#endif
        RAT_PROF_MARK(4);
    }
    RAT_PROF_FLUSH(a.prof, 36);

    // slab: [dW1 (H x D) | dW2 (D x H) | db1 (H) | db2 (D)]
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
    float* s_w1 = slab;
    float* s_w2 = s_w1 + (int64_t)H * D;
    float* s_b1 = s_w2 + (int64_t)D * H;
    float* s_b2 = s_b1 + H;
#pragma unroll
    for (int s = 0; s < WSLOTS; ++s) {
        const int id = rat_wave() + FFN_WAVES * s;
        if (id < t1) {
            const int col = rat_acc_col(id % t1n);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rat_acc_row(id / t1n, r);
                if (row < H && col < D) s_w1[(int64_t)row * D + col] = acc1[s][r];
            }
        }
        if (id < t2) {
            const int col = rat_acc_col(id % t2n);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rat_acc_row(id / t2n, r);
                if (row < D && col < H) s_w2[(int64_t)row * H + col] = acc2[s][r];
            }
        }
    }
    {   // combine the row-group partials of the bias gradients through LDS
        __syncthreads();
        float* red1 = hs;                                        // [nrg][H]
        float* red2 = gs;                                        // [nrg][D]
        const int nrg1 = FFN_THREADS / H, nrg2 = FFN_THREADS / D;
        if ((int)threadIdx.x / H < nrg1) red1[threadIdx.x] = db1;     // index = rg * H + col = threadIdx.x
        if ((int)threadIdx.x / D < nrg2) red2[threadIdx.x] = db2;
        __syncthreads();
        if (threadIdx.x < H) {
            float sacc = 0.f;
            for (int k = 0; k < nrg1; ++k) sacc += red1[k * H + threadIdx.x];
            s_b1[threadIdx.x] = sacc;
        }
        if (threadIdx.x < D) {
            float sacc = 0.f;
            for (int k = 0; k < nrg2; ++k) sacc += red2[k * D + threadIdx.x];
            s_b2[threadIdx.x] = sacc;
        }
    }
}


__device__ __forceinline__ float4 as_f4(const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ f32x4 as_v4(const float4& v) {
    f32x4 r = {v.x, v.y, v.z, v.w};
    return r;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int FT_THREADS = 1024;     // 16 waves = 4 per SIMD, ONE work-group per CU (uniform residency by construction)
constexpr int FT_WAVES = FT_THREADS / 64;

template <int D, int H>
struct FfnTGeom {
    static constexpr int LDW1 = D + 4, LDW2 = H + 4;
    static constexpr size_t fwd_smem = (size_t)(H * LDW1 + D * LDW2 + H + D) * 4;
};

// ======================================================================================================================
// Fast path (D, H multiples of 16; D <= 64): "token-on-lanes" formulation.  Every product is computed TRANSPOSED,
//     h^T = W1 x^T,   y^T = W2 g^T,   dh^T = W2^T dy^T,   dx^T = W1^T dhp^T,
// with the weights as the MFMA A operand and a 16-token tile as the B operand: lane (g, n) = (lane >> 4, lane & 15)
// holds x[token n][16 kb + 4 g + j] — one 16-byte global load per k-block.  The accumulator of such a product (rows =
// output features 4 g + r, column = token n) IS the B operand of the next product (rat_device.h k-permutation), and its four
// registers are four CONTIGUOUS features of one token: activations never pass through LDS between the GEMMs, results
// are stored straight from the accumulators with 16-byte stores, and a wave needs no barrier to run the whole chain.
// ======================================================================================================================
// two independent accumulator chains (M tiles `a0`, `a1` against the same B fragment), MFMAs issued alternately so that a
// wave alone sustains the 32-cycle issue rate (dependent-accumulator latency is 40 cycles)
__device__ __forceinline__ void mfma4x2(const float4& a0, const float4& a1, const float4& b, f32x4& c0, f32x4& c1) {
    c0 = RAT_MFMA16(a0.x, b.x, c0);
    c1 = RAT_MFMA16(a1.x, b.x, c1);
    c0 = RAT_MFMA16(a0.y, b.y, c0);
    c1 = RAT_MFMA16(a1.y, b.y, c1);
    c0 = RAT_MFMA16(a0.z, b.z, c0);
    c1 = RAT_MFMA16(a1.z, b.z, c1);
    c0 = RAT_MFMA16(a0.w, b.w, c0);
    c1 = RAT_MFMA16(a1.w, b.w, c1);
}
__device__ __forceinline__ float4 gelu4(const f32x4& h) {
    const rat_f2 g01 = rat_gelu2(rat_f2_make(h[0], h[1])), g23 = rat_gelu2(rat_f2_make(h[2], h[3]));
    return make_float4(g01.x, g01.y, g23.x, g23.y);
}

// forward: weights resident in LDS (A operands, 16-byte row reads); one 16-token tile per wave iteration, the next tile's
// x fragments prefetched into registers; hidden tiles are produced in pairs and the GELU of pair p-1 is scheduled into the
// MFMA shadow of pair p.
template <int D, int H, bool XRES>
__global__ void __launch_bounds__(FT_THREADS) ffn_fwd_t_kernel(FfnArgs a) {
    typedef FfnTGeom<D, H> G;
    constexpr int KD = D / 16, KH = H / 16;
    static_assert(KH % 2 == 0, "hidden tiles are processed in pairs");
    RAT_DYN_SMEM(smem);
    float* w1s = reinterpret_cast<float*>(smem);         // [H][LDW1]
    float* w2s = w1s + H * G::LDW1;                       // [D][LDW2]
    float* b1s = w2s + D * G::LDW2;                       // [H]
    float* b2s = b1s + H;                                 // [D]
    for (int e = threadIdx.x; e < H * D / 4; e += FT_THREADS) {
        const int r = e / (D / 4), c4 = e - r * (D / 4);
        st4(w1s + r * G::LDW1 + 4 * c4, ld4(a.w1 + (size_t)r * D + 4 * c4));
    }
    for (int e = threadIdx.x; e < D * H / 4; e += FT_THREADS) {
        const int r = e / (H / 4), c4 = e - r * (H / 4);
        st4(w2s + r * G::LDW2 + 4 * c4, ld4(a.w2 + (size_t)r * H + 4 * c4));
    }
    for (int e = threadIdx.x; e < H; e += FT_THREADS) b1s[e] = a.b1[e];
    for (int e = threadIdx.x; e < D; e += FT_THREADS) b2s[e] = a.b2[e];
    __syncthreads();

    const int l = rat_lane(), n = l & 15, g = l >> 4;
    const int64_t ntiles = (a.ntok + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * FT_WAVES;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t t = (int64_t)blockIdx.x * FT_WAVES + rat_wave();
    float4 xN[KD], xT[KD];
    {
        const int64_t tk = t * 16 + n;
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) xN[kb] = (t < ntiles && tk < a.ntok) ? ld4(a.x + tk * D + 16 * kb + 4 * g) : zero4;
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) xT[kb] = rat_consume4(xN[kb]);
    }
    for (; t < ntiles; t += stride) {
        const int64_t tok = t * 16 + n;
        {   // next tile's fragments: in flight behind this tile's 256 MFMAs; consumed (rat_consume4) before this tile's stores
            const int64_t tk = (t + stride) * 16 + n;
            const bool ok = t + stride < ntiles && tk < a.ntok;
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) xN[kb] = ok ? ld4(a.x + tk * D + 16 * kb + 4 * g) : zero4;
        }
        float4 gT[KH];
        f32x4 hp0 = rat_zero4(), hp1 = rat_zero4();
#pragma unroll
        for (int p = 0; p < KH / 2; ++p) {
            f32x4 c0 = as_v4(ld4(b1s + 32 * p + 4 * g)), c1 = as_v4(ld4(b1s + 32 * p + 16 + 4 * g));
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                const float4 a0 = ld4(w1s + (32 * p + n) * G::LDW1 + 16 * kb + 4 * g);
                const float4 a1 = ld4(w1s + (32 * p + 16 + n) * G::LDW1 + 16 * kb + 4 * g);
                mfma4x2(a0, a1, xT[kb], c0, c1);
            }
            if (p > 0) {
                gT[2 * p - 2] = gelu4(hp0);
                gT[2 * p - 1] = gelu4(hp1);
                RAT_SCHED_MFMA_VALU(8 * KD, (KD >= 4 ? 7 : 28));
            }
            hp0 = c0;
            hp1 = c1;
            RAT_SCHED_FENCE();
        }
        f32x4 yo[KD];
#pragma unroll
        for (int q = 0; q < (KD + 1) / 2; ++q) {
            const bool two = 2 * q + 1 < KD;
            const int m0 = 2 * q, m1 = two ? 2 * q + 1 : 2 * q;
            const float4 bs0 = ld4(b2s + 16 * m0 + 4 * g), bs1 = ld4(b2s + 16 * m1 + 4 * g);
            float4 r0 = xT[m0], r1 = xT[m1];                   // XRES: the residual is the tile already in registers
            if (!XRES) {
                const bool ok = a.res != nullptr && tok < a.ntok;
                r0 = ok ? ld4(a.res + tok * D + 16 * m0 + 4 * g) : zero4;
                r1 = ok ? ld4(a.res + tok * D + 16 * m1 + 4 * g) : zero4;
            }
            f32x4 c0 = {bs0.x + r0.x, bs0.y + r0.y, bs0.z + r0.z, bs0.w + r0.w};
            f32x4 c1 = {bs1.x + r1.x, bs1.y + r1.y, bs1.z + r1.z, bs1.w + r1.w};
            if (q == 0) {   // the last hidden pair's GELU rides in the shadow of the first K-blocks of this product
#pragma unroll
                for (int kb = 0; kb < KH - 2; ++kb) {
                    const float4 a0 = ld4(w2s + (16 * m0 + n) * G::LDW2 + 16 * kb + 4 * g);
                    const float4 a1 = ld4(w2s + (16 * m1 + n) * G::LDW2 + 16 * kb + 4 * g);
                    mfma4x2(a0, a1, gT[kb], c0, c1);
                }
                gT[KH - 2] = gelu4(hp0);
                gT[KH - 1] = gelu4(hp1);
                if (KH > 2) RAT_SCHED_MFMA_VALU(8 * (KH - 2), (KH >= 8 ? 5 : 28));
                RAT_SCHED_FENCE();
#pragma unroll
                for (int kb = KH - 2; kb < KH; ++kb) {
                    const float4 a0 = ld4(w2s + (16 * m0 + n) * G::LDW2 + 16 * kb + 4 * g);
                    const float4 a1 = ld4(w2s + (16 * m1 + n) * G::LDW2 + 16 * kb + 4 * g);
                    mfma4x2(a0, a1, gT[kb], c0, c1);
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < KH; ++kb) {
                    const float4 a0 = ld4(w2s + (16 * m0 + n) * G::LDW2 + 16 * kb + 4 * g);
                    const float4 a1 = ld4(w2s + (16 * m1 + n) * G::LDW2 + 16 * kb + 4 * g);
                    mfma4x2(a0, a1, gT[kb], c0, c1);
                }
            }
            yo[m0] = c0;
            if (two) yo[m1] = c1;
            RAT_SCHED_FENCE();
        }
        // hand-over point: every outstanding global operation (the prefetch issued at the top of this iteration, the stores
        // of the previous tile) is a whole tile old, so this wait is free; the stores below then start a fresh queue
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) xT[kb] = rat_consume4(xN[kb]);
        if (tok < a.ntok) {
#pragma unroll
            for (int m = 0; m < KD; ++m) st4(a.y + tok * D + 16 * m + 4 * g, as_f4(yo[m]));
        }
    }
}

// ---- bf16x3 variant of the token-on-lanes forward (D = 64, H = 128): both weight matrices live in LDS as three bf16 planes
// (rat_device.h "bf16x3"), split once per work-group; the token fragments are split in registers (x: once per tile, gelu(h): once
// per pair of hidden tiles) — each split feeds 8 resp. 4 row tiles.  W2 is stored with the hidden index PERMUTED so that the
// accumulators of hidden tiles (2 t, 2 t + 1) are, as they stand, the B fragment of K step t:
//     k slot j of lane group g  <->  hidden 32 t + 4 g + j (j < 4),  32 t + 16 + 4 g + (j - 4) (j >= 4).
constexpr int F3_D = 64, F3_H = 128;
typedef RatPlanes<128, 7, F3_H * 128> PlanesW1;          // [128 hidden][64 d]
typedef RatPlanes<256, 15, F3_D * 256> PlanesW2;         // [64 d][128 hidden, permuted]
constexpr size_t f3_fwd_smem() { return (size_t)3 * F3_H * 128 + (size_t)3 * F3_D * 256 + (size_t)(F3_H + F3_D) * 4; }

// DPAD: a narrower layer (embedding_dim 40 / 48 / 56, hidden = 2 d) inside the (64, 128) tiles — rows / columns beyond (d, hidden) are
// zeros in the staged planes, the biases and the token fragments, so they add nothing anywhere (gelu(0) = 0).
template <bool DPAD = false>
__device__ __forceinline__ void f3_stage_weights(const FfnArgs& a, const PlanesW1& w1p, const PlanesW2& w2p, int nthreads) {
    const int d = DPAD ? a.d : F3_D, hid = DPAD ? a.hidden : F3_H;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = threadIdx.x; e < F3_H * (F3_D / 8); e += nthreads) {
        const int r = e / (F3_D / 8), o = e - r * (F3_D / 8);
        rat_u4 h, m, l;
        const bool ok = !DPAD || (r < hid && 8 * o < d);
        rat_split8(ok ? ld4(a.w1 + (size_t)r * d + 8 * o) : z4, ok ? ld4(a.w1 + (size_t)r * d + 8 * o + 4) : z4, h, m, l);
        w1p.store(r, o, h, m, l);
    }
    for (int e = threadIdx.x; e < F3_D * (F3_H / 8); e += nthreads) {
        const int r = e / (F3_H / 8), o = e - r * (F3_H / 8), t = o >> 2, g = o & 3;
        rat_u4 h, m, l;
        const bool ok0 = !DPAD || (r < d && 32 * t + 4 * g < hid), ok1 = !DPAD || (r < d && 32 * t + 16 + 4 * g < hid);
        rat_split8(ok0 ? ld4(a.w2 + (size_t)r * hid + 32 * t + 4 * g) : z4, ok1 ? ld4(a.w2 + (size_t)r * hid + 32 * t + 16 + 4 * g) : z4, h, m, l);
        w2p.store(r, o, h, m, l);
    }
}

template <bool XRES, bool DPAD = false>
__global__ void __launch_bounds__(FT_THREADS) ffn_fwd_t3_kernel(FfnArgs a) {
    constexpr int D = F3_D, H = F3_H;
    const int dr = DPAD ? a.d : D, hid = DPAD ? a.hidden : H;                // the layer's real width / hidden width
    RAT_DYN_SMEM(smem);
    const PlanesW1 w1p{smem};
    const PlanesW2 w2p{smem + 3 * H * 128};
    float* b1s = reinterpret_cast<float*>(smem + 3 * H * 128 + 3 * D * 256);
    float* b2s = b1s + H;
    f3_stage_weights<DPAD>(a, w1p, w2p, FT_THREADS);
    for (int e = threadIdx.x; e < H; e += FT_THREADS) b1s[e] = (!DPAD || e < hid) ? a.b1[e] : 0.f;
    for (int e = threadIdx.x; e < D; e += FT_THREADS) b2s[e] = (!DPAD || e < dr) ? a.b2[e] : 0.f;
    __syncthreads();

    const int l = rat_lane(), n = l & 15, g = l >> 4;
    const int64_t ntiles = (a.ntok + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * FT_WAVES;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t t = (int64_t)blockIdx.x * FT_WAVES + rat_wave();
    float4 xN[4];                                             // x[token n][32 s + 8 g .. + 7] for s = 0, 1 (two float4 each)
    {
        const int64_t tk = t * 16 + n;
        const bool ok = t < ntiles && tk < a.ntok;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
            xN[q] = (ok && (!DPAD || c < dr)) ? ld4(a.x + tk * dr + c) : zero4;
        }
    }
    for (; t < ntiles; t += stride) {
        const int64_t tok = t * 16 + n;
        RatB3 xb[2];
        {
            float4 xT[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xT[q] = rat_consume4(xN[q]);
            xb[0] = rat_split8_frag(xT[0], xT[1]);
            xb[1] = rat_split8_frag(xT[2], xT[3]);
        }
        {   // next tile's fragments: in flight behind this tile's MFMAs
            const int64_t tk = (t + stride) * 16 + n;
            const bool ok = t + stride < ntiles && tk < a.ntok;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
                xN[q] = (ok && (!DPAD || c < dr)) ? ld4(a.x + tk * dr + c) : zero4;
            }
        }
        f32x4 yo[D / 16];
#pragma unroll
        for (int c = 0; c < D / 16; ++c) {
            float4 r0 = zero4;
            const float* rp = XRES ? a.x : a.res;
            if (rp != nullptr && tok < a.ntok && (!DPAD || 16 * c + 4 * g < dr)) r0 = ld4(rp + tok * dr + 16 * c + 4 * g);
            const float4 bs = ld4(b2s + 16 * c + 4 * g);
            yo[c] = f32x4{bs.x + r0.x, bs.y + r0.y, bs.z + r0.z, bs.w + r0.w};
        }
#pragma unroll
        for (int p = 0; p < H / 32; ++p) {
            if (DPAD && 32 * p >= hid) break;                                // (uniform) hidden tiles that do not exist
            f32x4 c0 = as_v4(ld4(b1s + 32 * p + 4 * g)), c1 = as_v4(ld4(b1s + 32 * p + 16 + 4 * g));
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const RatB3 a0 = w1p.row_frag(2 * p, s), a1 = w1p.row_frag(2 * p + 1, s);
                f32x4 cc[2] = {c0, c1};
                const RatB3 aa[2] = {a0, a1};
                rat_mfma3_block<2>(cc, aa, xb[s]);
                c0 = cc[0];
                c1 = cc[1];
            }
            const RatB3 gb = rat_split8_frag(gelu4(c0), gelu4(c1));
#pragma unroll
            for (int c = 0; c < D / 16; c += 2) {
                f32x4 cc[2] = {yo[c], yo[c + 1]};
                const RatB3 aa[2] = {w2p.row_frag(c, p), w2p.row_frag(c + 1, p)};
                rat_mfma3_block<2>(cc, aa, gb);
                yo[c] = cc[0];
                yo[c + 1] = cc[1];
            }
        }
        if (tok < a.ntok) {
#pragma unroll
            for (int m = 0; m < D / 16; ++m)
                if (!DPAD || 16 * m + 4 * g < dr) st4(a.y + tok * dr + 16 * m + 4 * g, as_f4(yo[m]));
        }
    }
}

// backward, same formulation.  512 threads (8 waves, one work-group per CU), 64 tokens per iteration: wave (tt, half) =
// (wave & 3, wave >> 2) owns token tile tt and one half of the hidden tiles.  Chain phase (no barrier): h^T and dh^T
// (shared token B fragments x^T / dy^T, weights W1 / W2^T rows as A operands straight from L2), gelu / gelu' on the
// accumulators, partial dx^T over the half's hidden tiles from the accumulator-resident dh'.  gelu(h) and dh' are written
// TOKEN-major into LDS only as operands of the weight-gradient GEMMs (dW1 += dh'^T x, dW2 += dy^T gelu(h)), whose
// accumulators persist across the work-group's chunk loop.  Two barriers per 64 tokens (the slab-era kernel had six).
constexpr int FB_THREADS = 512;
constexpr int FB_WAVES = FB_THREADS / 64;
constexpr int FB_TOK = 64;

template <int D, int H>
struct FfnBTGeom {
    static constexpr int LDX = D + 4, LDH = H + 4;
    static constexpr size_t smem = (size_t)FB_TOK * (3 * LDX + 2 * LDH) * 4;
};

__device__ __forceinline__ void mfma4x2b(const float4& a0, const float4& a1, const float4& b0, const float4& b1, f32x4& c0, f32x4& c1) {
    c0 = RAT_MFMA16(a0.x, b0.x, c0);
    c1 = RAT_MFMA16(a1.x, b1.x, c1);
    c0 = RAT_MFMA16(a0.y, b0.y, c0);
    c1 = RAT_MFMA16(a1.y, b1.y, c1);
    c0 = RAT_MFMA16(a0.z, b0.z, c0);
    c1 = RAT_MFMA16(a1.z, b1.z, c1);
    c0 = RAT_MFMA16(a0.w, b0.w, c0);
    c1 = RAT_MFMA16(a1.w, b1.w, c1);
}

template <int D, int H>
__global__ void __launch_bounds__(FB_THREADS) ffn_bwd_t_kernel(FfnArgs a) {
    typedef FfnBTGeom<D, H> G;
    constexpr int KD = D / 16, KH = H / 16, HT = KH / 2;          // HT hidden tiles per wave half
    constexpr int NT = KH * KD;                                    // 16x16 tiles of dW1 (and of dW2)
    constexpr int SL = (NT + FB_WAVES - 1) / FB_WAVES;             // persistent tiles per wave, each of dW1 / dW2
    static_assert(KH % 2 == 0 && SL <= 4, "geometry");
    RAT_DYN_SMEM(smem);
    float* xs = reinterpret_cast<float*>(smem);                    // [64][LDX] x         (B operand of dW1)
    float* dys = xs + FB_TOK * G::LDX;                             // [64][LDX] dy        (A^T operand of dW2)
    float* px = dys + FB_TOK * G::LDX;                             // [64][LDX] dx partial of the upper hidden half
    float* gs = px + FB_TOK * G::LDX;                              // [64][LDH] gelu(h)   (B operand of dW2)
    float* dhs = gs + FB_TOK * G::LDH;                             // [64][LDH] dh'       (A^T operand of dW1)

    const int l = rat_lane(), n = l & 15, g = l >> 4;
    const int w = rat_wave(), tt = w & 3, half = w >> 2;
    const int row = 16 * tt + n;                                   // this lane's token row inside the 64-token chunk
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4 acc1[SL], acc2[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) acc1[s] = acc2[s] = rat_zero4();
    f32x4 db1a[HT], db2a[KD];                                      // bias-gradient partials of this lane's token column
#pragma unroll
    for (int i = 0; i < HT; ++i) db1a[i] = rat_zero4();
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) db2a[kb] = rat_zero4();

    float4 xN[KD], dyN[KD], xT[KD], dyT[KD];
    int64_t chunk = blockIdx.x;
    {
        const int64_t tk = chunk * FB_TOK + row;
        const bool ok = chunk < a.nchunks && tk < a.ntok;
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) {
            xN[kb] = ok ? ld4(a.x + tk * D + 16 * kb + 4 * g) : zero4;
            dyN[kb] = ok ? ld4(a.dy + tk * D + 16 * kb + 4 * g) : zero4;
        }
    }
    RAT_PROF_DECL
    for (; chunk < a.nchunks; chunk += gridDim.x) {
        const int64_t tok = chunk * FB_TOK + row;
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) {
            xT[kb] = rat_consume4(xN[kb]);
            dyT[kb] = rat_consume4(dyN[kb]);
        }
        if (half == 0) {
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) st4(xs + row * G::LDX + 16 * kb + 4 * g, xT[kb]);
        } else {
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) st4(dys + row * G::LDX + 16 * kb + 4 * g, dyT[kb]);
        }
        RAT_PROF_MARK(0);
        // ---- chain: h^T, dh^T per hidden tile; gelu / gelu' of tile i-1 in the MFMA shadow of tile i
        float4 dhpT[HT];
        f32x4 hp0 = rat_zero4(), hp1 = rat_zero4();
        auto epilogue = [&](int i, const f32x4& hacc, const f32x4& dacc) {
            float4 gv, dp;
            float dg;
            rat_gelu_both(hacc[0], gv.x, dg);
            dp.x = dacc[0] * dg;
            rat_gelu_both(hacc[1], gv.y, dg);
            dp.y = dacc[1] * dg;
            rat_gelu_both(hacc[2], gv.z, dg);
            dp.z = dacc[2] * dg;
            rat_gelu_both(hacc[3], gv.w, dg);
            dp.w = dacc[3] * dg;
            const int col = 16 * (HT * half + i) + 4 * g;
            st4(gs + row * G::LDH + col, gv);
            st4(dhs + row * G::LDH + col, dp);
            dhpT[i] = dp;
            db1a[i][0] += dp.x;
            db1a[i][1] += dp.y;
            db1a[i][2] += dp.z;
            db1a[i][3] += dp.w;
        };
#pragma unroll
        for (int i = 0; i < HT; ++i) {
            const int m = HT * half + i;
            f32x4 c0 = as_v4(ld4(a.b1 + 16 * m + 4 * g)), c1 = rat_zero4();
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                const float4 a0 = ld4(a.w1 + (size_t)(16 * m + n) * D + 16 * kb + 4 * g);
                const float4 a1 = ld4(a.w2t + (size_t)(16 * m + n) * D + 16 * kb + 4 * g);
                mfma4x2b(a0, a1, xT[kb], dyT[kb], c0, c1);
            }
            if (i > 0) {
                epilogue(i - 1, hp0, hp1);
                RAT_SCHED_MFMA_VALU(8 * KD, (KD >= 4 ? 5 : 20));
            }
            hp0 = c0;
            hp1 = c1;
            RAT_SCHED_FENCE();
        }
        RAT_PROF_MARK(1);
        // ---- partial dx^T over this half's hidden tiles (k-blocks = the accumulator-resident dh' tiles)
        f32x4 dxa[KD];
#pragma unroll
        for (int q = 0; q < (KD + 1) / 2; ++q) {
            const bool two = 2 * q + 1 < KD;
            const int m0 = 2 * q, m1 = two ? 2 * q + 1 : 2 * q;
            f32x4 c0 = rat_zero4(), c1 = rat_zero4();
            if (q == 0) {
#pragma unroll
                for (int i = 0; i < HT - 1; ++i) {
                    const float4 a0 = ld4(a.w1t + (size_t)(16 * m0 + n) * H + 16 * (HT * half + i) + 4 * g);
                    const float4 a1 = ld4(a.w1t + (size_t)(16 * m1 + n) * H + 16 * (HT * half + i) + 4 * g);
                    mfma4x2(a0, a1, dhpT[i], c0, c1);
                }
                epilogue(HT - 1, hp0, hp1);
                if (HT > 1) RAT_SCHED_MFMA_VALU(8 * (HT - 1), (HT >= 4 ? 7 : 28));
                RAT_SCHED_FENCE();
                {
                    const float4 a0 = ld4(a.w1t + (size_t)(16 * m0 + n) * H + 16 * (HT * half + HT - 1) + 4 * g);
                    const float4 a1 = ld4(a.w1t + (size_t)(16 * m1 + n) * H + 16 * (HT * half + HT - 1) + 4 * g);
                    mfma4x2(a0, a1, dhpT[HT - 1], c0, c1);
                }
            } else {
#pragma unroll
                for (int i = 0; i < HT; ++i) {
                    const float4 a0 = ld4(a.w1t + (size_t)(16 * m0 + n) * H + 16 * (HT * half + i) + 4 * g);
                    const float4 a1 = ld4(a.w1t + (size_t)(16 * m1 + n) * H + 16 * (HT * half + i) + 4 * g);
                    mfma4x2(a0, a1, dhpT[i], c0, c1);
                }
            }
            dxa[m0] = c0;
            if (two) dxa[m1] = c1;
            RAT_SCHED_FENCE();
        }
        RAT_PROF_MARK(2);
        // the d tiles of dx are split between the halves: each parks its partial of the OTHER half's tiles in LDS
        constexpr int KD0 = (KD + 1) / 2;                          // half 0 finishes tiles [0, KD0), half 1 tiles [KD0, KD)
        if (half == 0) {
#pragma unroll
            for (int m = KD0; m < KD; ++m) st4(px + row * G::LDX + 16 * m + 4 * g, as_f4(dxa[m]));
        } else {
#pragma unroll
            for (int m = 0; m < KD0; ++m) st4(px + row * G::LDX + 16 * m + 4 * g, as_f4(dxa[m]));
        }
        __syncthreads();
        RAT_PROF_MARK(3);
        auto finish = [&](int m) {                                 // dx = dy + dh' W1 (both halves), straight from registers
            const float4 p = ld4(px + row * G::LDX + 16 * m + 4 * g);
            const float4 dr = a.add_dy ? dyT[m] : zero4;
            const float4 o = make_float4(dxa[m][0] + p.x + dr.x, dxa[m][1] + p.y + dr.y, dxa[m][2] + p.z + dr.z, dxa[m][3] + p.w + dr.w);
            if (tok < a.ntok) st4(a.y + tok * D + 16 * m + 4 * g, o);
            db2a[m][0] += dyT[m].x;
            db2a[m][1] += dyT[m].y;
            db2a[m][2] += dyT[m].z;
            db2a[m][3] += dyT[m].w;
        };
        if (half == 0) {
#pragma unroll
            for (int m = 0; m < KD0; ++m) finish(m);
        } else {
#pragma unroll
            for (int m = KD0; m < KD; ++m) finish(m);
        }
        {   // next chunk's token fragments: in flight behind the weight-gradient GEMMs
            const int64_t tk = (chunk + gridDim.x) * FB_TOK + row;
            const bool ok = chunk + gridDim.x < a.nchunks && tk < a.ntok;
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                xN[kb] = ok ? ld4(a.x + tk * D + 16 * kb + 4 * g) : zero4;
                dyN[kb] = ok ? ld4(a.dy + tk * D + 16 * kb + 4 * g) : zero4;
            }
        }
        RAT_PROF_MARK(4);
        // ---- dW1 += dh'^T x ; dW2 += dy^T gelu(h)    (contraction over the chunk's 64 tokens)
        rat_wave_gemm_ct<SL, FB_WAVES, NT, KD, FB_TOK / 16>(acc1, RatLdsCols{dhs, G::LDH}, RatLdsCols{xs, G::LDX});
        rat_wave_gemm_ct<SL, FB_WAVES, NT, KH, FB_TOK / 16>(acc2, RatLdsCols{dys, G::LDX}, RatLdsCols{gs, G::LDH});
        RAT_PROF_MARK(5);
        __syncthreads();
        RAT_PROF_MARK(6);
    }
    RAT_PROF_FLUSH(a.prof, 72);

    // slab: [dW1 (H x D) | dW2 (D x H) | db1 (H) | db2 (D)]
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
    float* s_w1 = slab;
    float* s_w2 = s_w1 + (int64_t)H * D;
    float* s_b1 = s_w2 + (int64_t)D * H;
    float* s_b2 = s_b1 + H;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int id = w + FB_WAVES * s;
        if (id < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s_w1[(int64_t)rat_acc_row(id / KD, r) * D + rat_acc_col(id % KD)] = acc1[s][r];
                s_w2[(int64_t)rat_acc_row(id / KH, r) * H + rat_acc_col(id % KH)] = acc2[s][r];
            }
        }
    }
    {   // bias gradients: per-token-column partials -> LDS [feature][64 token columns] -> fixed-order sums
        constexpr int LR = FB_TOK + 1;
        float* red1 = reinterpret_cast<float*>(smem);              // [H][LR]
        float* red2 = red1 + H * LR;                               // [D][LR]
#pragma unroll
        for (int i = 0; i < HT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red1[(16 * (HT * half + i) + 4 * g + r) * LR + row] = db1a[i][r];
#pragma unroll
        for (int kb = 0; kb < KD; ++kb)
            if ((kb < (KD + 1) / 2) == (half == 0))
#pragma unroll
                for (int r = 0; r < 4; ++r) red2[(16 * kb + 4 * g + r) * LR + row] = db2a[kb][r];
        __syncthreads();
        for (int c = threadIdx.x; c < H + D; c += FB_THREADS) {
            const float* src = c < H ? red1 + c * LR : red2 + (c - H) * LR;
            float sacc = 0.f;
            for (int k = 0; k < FB_TOK; ++k) sacc += src[k];
            if (c < H) s_b1[c] = sacc; else s_b2[c - H] = sacc;
        }
    }
}

// ---- bf16x3 variant of the token-on-lanes backward (D = 64, H = 128): all five GEMMs on v_mfma_f32_16x16x32_bf16.
//   * weights: A operands of the three chain products, as pre-split fragments from L2 (W1 rows, W2^T rows, W1^T rows with the hidden
//     index permuted like PlanesW2 above so that stacked dh' accumulator tiles are its B fragment);
//   * token fragments x^T / dy^T: split once per chunk in registers — the split pieces ARE the plane pieces of the token-major
//     LDS images that the weight-gradient GEMMs read with transposed block reads (no second split, no fp32 tile);
//   * gelu(h) and dh': split on the accumulators (4 values per lane and tile), stored as 8-byte half pieces.
// LDS: [x planes 24576][dy planes 24576][gelu(h) planes 49152][dh' planes 49152][dx exchange 16384] = 160 KiB.
typedef RatPlanes<128, 7, 64 * 128> PlanesT64;           // [64 tokens][64]
typedef RatPlanes<256, 7, 64 * 256, 1> PlanesT128;       // [64 tokens][128]
struct Ffn3W {
    RatWPlanes w1;       // A[hidden][k = d]            = w1[hidden][k]      N 128, K 64
    RatWPlanes w2t;      // A[hidden][k = d]            = w2[k][hidden]      N 128, K 64
    RatWPlanes w1t;      // A[d][k = hidden, permuted]  = w1[k][d]           N 64,  K 128
};
constexpr size_t F3_WP = (size_t)8 * 2 * 3 * 1024;     // bytes of each of the three fragment sets
constexpr size_t f3_bwd_smem() { return (size_t)2 * 3 * 64 * 128 + (size_t)2 * 3 * 64 * 256 + (size_t)64 * 64 * 4; }

struct HalfPieces {                                      // the three planes of 4 consecutive values (one accumulator quad)
    unsigned h0, h1, m0, m1, l0, l1;
};
__device__ __forceinline__ HalfPieces f3_split4(const float4& v) {
    HalfPieces p;
    rat_split2(v.x, v.y, p.h0, p.m0, p.l0);
    rat_split2(v.z, v.w, p.h1, p.m1, p.l1);
    return p;
}

template <bool DPAD = false>
__global__ void __launch_bounds__(FB_THREADS) ffn_bwd_t3_kernel(FfnArgs a, Ffn3W W) {
    constexpr int D = F3_D, H = F3_H, KD = D / 16, HT = 4, SL = 4;
    const int dr = DPAD ? a.d : D, hid = DPAD ? a.hidden : H;                // DPAD: see ffn_fwd_t3_kernel
    RAT_DYN_SMEM(smem);
    const PlanesT64 xsp{smem};
    const PlanesT64 dysp{smem + 3 * 64 * 128};
    const PlanesT128 gsp{smem + 2 * 3 * 64 * 128};
    const PlanesT128 dhsp{smem + 2 * 3 * 64 * 128 + 3 * 64 * 256};
    float* px = reinterpret_cast<float*>(smem + 2 * 3 * 64 * 128 + 2 * 3 * 64 * 256);     // [64][64], 16-byte pieces XOR-swizzled

    const int l = rat_lane(), n = l & 15, g = l >> 4;
    const int w = rat_wave(), tt = w & 3, half = w >> 2;
    const int row = 16 * tt + n;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto pxp = [&](int m) { return px + row * D + 4 * ((4 * m + g) ^ (row & 15)); };

    f32x4 acc1[SL], acc2[SL], db1a[HT];
#pragma unroll
    for (int i = 0; i < SL; ++i) acc1[i] = acc2[i] = db1a[i] = rat_zero4();
    float4 db2f[4] = {zero4, zero4, zero4, zero4};                 // column sums of dy in fragment order (half 1 only)

    float4 xN[4], dyN[4];                                          // [2 s + part]: x[token][32 s + 8 g + 4 part .. + 3]
    int64_t chunk = blockIdx.x;
    {
        const int64_t tk = chunk * FB_TOK + row;
        const bool ok = chunk < a.nchunks && tk < a.ntok;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
            const bool okc = ok && (!DPAD || c < dr);
            xN[q] = okc ? ld4(a.x + tk * dr + c) : zero4;
            dyN[q] = okc ? ld4(a.dy + tk * dr + c) : zero4;
        }
    }
    RAT_PROF_DECL
    for (; chunk < a.nchunks; chunk += gridDim.x) {
        const int64_t tok = chunk * FB_TOK + row;
        RatB3 xb[2], dyb[2];
        {
            float4 xT[4], dyT[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xT[q] = rat_consume4(xN[q]);
                dyT[q] = rat_consume4(dyN[q]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                rat_u4 h, m, lo;
                rat_split8(xT[2 * s], xT[2 * s + 1], h, m, lo);
                xb[s] = RatB3{rat_as_bf16x8(h), rat_as_bf16x8(m), rat_as_bf16x8(lo)};
                if (half == 0) xsp.store(row, 4 * s + g, h, m, lo);
                rat_split8(dyT[2 * s], dyT[2 * s + 1], h, m, lo);
                dyb[s] = RatB3{rat_as_bf16x8(h), rat_as_bf16x8(m), rat_as_bf16x8(lo)};
                if (half == 1) dysp.store(row, 4 * s + g, h, m, lo);
            }
            if (half == 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    db2f[q].x += dyT[q].x; db2f[q].y += dyT[q].y; db2f[q].z += dyT[q].z; db2f[q].w += dyT[q].w;
                }
        }
        RAT_PROF_MARK(0);
        // ---- chain: h^T = W1 x^T (+ b1), dh^T = W2^T dy^T per hidden tile; gelu / gelu' on the accumulators
        RatB3 dbs[HT / 2];                                       // stacked dh' quads of tiles (2 t, 2 t + 1): the dx phase's B fragments
        // the even tile's dh' pieces are needed again when the odd tile's exist (f3_stack): carried in registers they were kept in
        // SCRATCH by the compiler at 256 VGPRs (a 24-byte stack object: scratch_store x4 + x2 and two overlapping scratch_load x4 per
        // tile pair, 44 bytes per lane of scratch, +155 MB of HBM writes per launch — r2_traffic_pmc.json) — they are re-read from the
        // dh' planes instead, where this same thread has just stored them (3 ds_read_b64)
#pragma unroll
        for (int i = 0; i < HT; ++i) {
            const int m = HT * half + i;
            f32x4 c0 = (!DPAD || 16 * m + 4 * g < hid) ? as_v4(ld4(a.b1 + 16 * m + 4 * g)) : rat_zero4(), c1 = rat_zero4();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const RatB3 a0 = W.w1(m, s), a1 = W.w2t(m, s);
                c0 = RAT_MFMA_BF16(a0.l, xb[s].h, c0);
                c1 = RAT_MFMA_BF16(a1.l, dyb[s].h, c1);
                c0 = RAT_MFMA_BF16(a0.h, xb[s].l, c0);
                c1 = RAT_MFMA_BF16(a1.h, dyb[s].l, c1);
                c0 = RAT_MFMA_BF16(a0.m, xb[s].m, c0);
                c1 = RAT_MFMA_BF16(a1.m, dyb[s].m, c1);
                c0 = RAT_MFMA_BF16(a0.m, xb[s].h, c0);
                c1 = RAT_MFMA_BF16(a1.m, dyb[s].h, c1);
                c0 = RAT_MFMA_BF16(a0.h, xb[s].m, c0);
                c1 = RAT_MFMA_BF16(a1.h, dyb[s].m, c1);
                c0 = RAT_MFMA_BF16(a0.h, xb[s].h, c0);
                c1 = RAT_MFMA_BF16(a1.h, dyb[s].h, c1);
            }
            float4 gv, dp;
            float dg;
            rat_gelu_both(c0[0], gv.x, dg);
            dp.x = c1[0] * dg;
            rat_gelu_both(c0[1], gv.y, dg);
            dp.y = c1[1] * dg;
            rat_gelu_both(c0[2], gv.z, dg);
            dp.z = c1[2] * dg;
            rat_gelu_both(c0[3], gv.w, dg);
            dp.w = c1[3] * dg;
            const HalfPieces gp = f3_split4(gv);
            const HalfPieces dq = f3_split4(dp);
            gsp.store_half(row, 4 * m + g, gp.h0, gp.h1, gp.m0, gp.m1, gp.l0, gp.l1);
            dhsp.store_half(row, 4 * m + g, dq.h0, dq.h1, dq.m0, dq.m1, dq.l0, dq.l1);
            if (i % 2 == 1) {
                unsigned eh0, eh1, em0, em1, el0, el1;                    // (plain scalars: a struct here is kept in scratch too)
                dhsp.load_half(row, 4 * (m - 1) + g, eh0, eh1, em0, em1, el0, el1);
                rat_u4 sh, sm, sl;
                sh.x = eh0; sh.y = eh1; sh.z = dq.h0; sh.w = dq.h1;
                sm.x = em0; sm.y = em1; sm.z = dq.m0; sm.w = dq.m1;
                sl.x = el0; sl.y = el1; sl.z = dq.l0; sl.w = dq.l1;
                dbs[i / 2] = RatB3{rat_as_bf16x8(sh), rat_as_bf16x8(sm), rat_as_bf16x8(sl)};
            }
            db1a[i][0] += dp.x;
            db1a[i][1] += dp.y;
            db1a[i][2] += dp.z;
            db1a[i][3] += dp.w;
        }
        RAT_PROF_MARK(1);
        // ---- partial dx^T over this half's hidden tiles: the stacked dh' quads of tiles (2 t, 2 t + 1) are the B fragment
        f32x4 dxa[KD];
#pragma unroll
        for (int m = 0; m < KD; ++m) dxa[m] = rat_zero4();
#pragma unroll
        for (int t = 0; t < HT / 2; ++t) {
            const RatB3 db = dbs[t];
#pragma unroll
            for (int m = 0; m < KD; m += 2) {
                f32x4 cc[2] = {dxa[m], dxa[m + 1]};
                const RatB3 aa[2] = {W.w1t(m, 2 * half + t), W.w1t(m + 1, 2 * half + t)};
                rat_mfma3_block<2>(cc, aa, db);
                dxa[m] = cc[0];
                dxa[m + 1] = cc[1];
            }
        }
        // the d tiles of dx are split between the halves: each parks its partial of the OTHER half's tiles in LDS
        constexpr int KD0 = KD / 2;
        if (half == 0) {
#pragma unroll
            for (int m = KD0; m < KD; ++m) st4(pxp(m), as_f4(dxa[m]));
        } else {
#pragma unroll
            for (int m = 0; m < KD0; ++m) st4(pxp(m), as_f4(dxa[m]));
        }
        RAT_PROF_MARK(2);
        __syncthreads();
        RAT_PROF_MARK(3);
        // the residual term + dy: this work-group holds dy already — as the three bf16 planes of the dy tile in LDS, whose sum
        // h + m + l IS the fp32 value (the split is exact) — so the four columns a lane needs are rebuilt from three 8-byte LDS reads
        // instead of a second trip to L2 (the global re-read was the exposed latency of this phase: 13.6 % of the iteration,
        // profiles/round3/r3_phase_shares.txt)
        const bool live = tok < a.ntok;
        auto residual = [&](int m) {
            unsigned h0, h1, m0, m1, l0, l1;
            dysp.load_half(row, 4 * m + g, h0, h1, m0, m1, l0, l1);
            return make_float4(rat_join(h0, m0, l0, 0), rat_join(h0, m0, l0, 1), rat_join(h1, m1, l1, 0), rat_join(h1, m1, l1, 1));
        };
        auto finish = [&](int m, const float4& rsd) {
            const float4 p = ld4(pxp(m));
            const float4 o = make_float4(dxa[m][0] + p.x + rsd.x, dxa[m][1] + p.y + rsd.y, dxa[m][2] + p.z + rsd.z, dxa[m][3] + p.w + rsd.w);
            if (live && (!DPAD || 16 * m + 4 * g < dr)) st4(a.y + tok * dr + 16 * m + 4 * g, o);
        };
        static_assert(KD0 == 2, "two dx tiles per half");
        if (half == 0) {
            float4 r0 = zero4, r1 = zero4;
            if (a.add_dy) {                                         // uniform
                r0 = residual(0);
                r1 = residual(1);
            }
            finish(0, r0);
            finish(1, r1);
        } else {
            float4 r0 = zero4, r1 = zero4;
            if (a.add_dy) {
                r0 = residual(2);
                r1 = residual(3);
            }
            finish(2, r0);
            finish(3, r1);
        }
        {   // next chunk's token fragments: in flight behind the weight-gradient GEMMs
            const int64_t tk = (chunk + gridDim.x) * FB_TOK + row;
            const bool ok = chunk + gridDim.x < a.nchunks && tk < a.ntok;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
                const bool okc = ok && (!DPAD || c < dr);
                xN[q] = okc ? ld4(a.x + tk * dr + c) : zero4;
                dyN[q] = okc ? ld4(a.dy + tk * dr + c) : zero4;
            }
        }
        RAT_PROF_MARK(4);
        // ---- dW1 += dh'^T x (tiles: hidden (w >> 2) + 2 i  x  d (w & 3)) ; dW2 += dy^T gelu(h) (tiles: d (w & 3)  x  hidden (w >> 2) + 2 i)
        {
            const RatB3 xb0 = xsp.col_frag(w & 3, 0), xb1 = xsp.col_frag(w & 3, 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int mt = (w >> 2) + 2 * i;
                acc1[i] = rat_mfma3(dhsp.col_frag(mt, 0), xb0, acc1[i]);
                acc1[i] = rat_mfma3(dhsp.col_frag(mt, 1), xb1, acc1[i]);
            }
            const RatB3 ya0 = dysp.col_frag(w & 3, 0), ya1 = dysp.col_frag(w & 3, 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int nt = (w >> 2) + 2 * i;
                acc2[i] = rat_mfma3(ya0, gsp.col_frag(nt, 0), acc2[i]);
                acc2[i] = rat_mfma3(ya1, gsp.col_frag(nt, 1), acc2[i]);
            }
        }
        RAT_PROF_MARK(5);
        __syncthreads();
        RAT_PROF_MARK(6);
    }
    RAT_PROF_FLUSH(a.prof, 72);

    // slab: [dW1 (H x D) | dW2 (D x H) | db1 (H) | db2 (D)]
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
    float* s_w1 = slab;                                           // (the host's layout: [hidden][d], [d][hidden], [hidden], [d])
    float* s_w2 = s_w1 + (int64_t)hid * dr;
    float* s_b1 = s_w2 + (int64_t)dr * hid;
    float* s_b2 = s_b1 + hid;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
        const int t2 = (w >> 2) + 2 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!DPAD || (rat_acc_row(t2, r) < hid && rat_acc_col(w & 3) < dr))
                s_w1[(int64_t)rat_acc_row(t2, r) * dr + rat_acc_col(w & 3)] = acc1[i][r];
            if (!DPAD || (rat_acc_row(w & 3, r) < dr && rat_acc_col(t2) < hid))
                s_w2[(int64_t)rat_acc_row(w & 3, r) * hid + rat_acc_col(t2)] = acc2[i][r];
        }
    }
    {   // bias gradients: per-token-column partials -> LDS [feature][64 token columns] -> fixed-order sums
        constexpr int LR = FB_TOK + 1;
        float* red1 = reinterpret_cast<float*>(smem);              // [H][LR]
        float* red2 = red1 + H * LR;                               // [D][LR]
#pragma unroll
        for (int i = 0; i < HT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red1[(16 * (HT * half + i) + 4 * g + r) * LR + row] = db1a[i][r];
        if (half == 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
                red2[(c0 + 0) * LR + row] = db2f[q].x;
                red2[(c0 + 1) * LR + row] = db2f[q].y;
                red2[(c0 + 2) * LR + row] = db2f[q].z;
                red2[(c0 + 3) * LR + row] = db2f[q].w;
            }
        __syncthreads();
        for (int c = threadIdx.x; c < H + D; c += FB_THREADS) {
            const float* src = c < H ? red1 + c * LR : red2 + (c - H) * LR;
            float sacc = 0.f;
            for (int k = 0; k < FB_TOK; ++k) sacc += src[k];
            if (c < H) {
                if (!DPAD || c < hid) s_b1[c] = sacc;
            } else if (!DPAD || c - H < dr) {
                s_b2[c - H] = sacc;
            }
        }
    }
}

// ---- round 4: the same backward with the WEIGHTS stationary.  In ffn_bwd_t3_kernel a wave owns a token tile and streams the weight
// fragments of four hidden tiles from L2: every fragment is fetched by four waves, 576 KB per 64 tokens and work-group — at ~60 B per
// clock and CU that stream alone is longer than the kernel's MFMAs (PMC round 3: MFMA busy 29 %, waves waiting 46 %).  Here a wave owns
// a HIDDEN tile: its W1 / W2^T fragments (48 VGPRs) stay in registers for the whole launch and the token fragments come from the plane
// images in LDS, which the weight-gradient GEMMs need there anyway:
//   split   wave (tile tt, half): half 0 splits x of token tile tt into the x planes, half 1 dy into the dy planes        | barrier
//   chain   wave w = hidden tile: for the four token tiles  h^T = W1 x^T + b1, dh^T = W2^T dy^T  (B fragments = 16-byte row reads of
//           the planes), gelu / gelu' on the accumulators, gelu(h) and dh' split into their planes                          | barrier
//   dx      wave (d tile w & 3, token tiles 2 (w >> 2), + 1): dx^T = W1^T dh'^T over all 128 hidden (A: W1^T fragments from L2, 12 KB
//           per wave — a sixth of the old stream; B: the dh' planes in W1^T's permuted k order, 8-byte reads) + dy, stored from the
//           accumulators: no partial sums to exchange, the 16 KB exchange tile and its traffic are gone
//   dW      as before (transposed block reads of the four plane images)                                                     | barrier
template <bool DPAD = false>
__global__ void __launch_bounds__(FB_THREADS) ffn_bwd_t4_kernel(FfnArgs a, Ffn3W W) {
    constexpr int D = F3_D, H = F3_H, SL = 4;
    const int dr = DPAD ? a.d : D, hid = DPAD ? a.hidden : H;                // DPAD: see ffn_fwd_t3_kernel
    RAT_DYN_SMEM(smem);
    const PlanesT64 xsp{smem};
    const PlanesT64 dysp{smem + 3 * 64 * 128};
    const PlanesT128 gsp{smem + 2 * 3 * 64 * 128};
    const PlanesT128 dhsp{smem + 2 * 3 * 64 * 128 + 3 * 64 * 256};

    const int l = rat_lane(), n = l & 15, g = l >> 4;
    const int w = rat_wave(), tt = w & 3, half = w >> 2;
    const int row = 16 * tt + n;                                             // the token whose x (half 0) / dy (half 1) this thread splits
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* const src = half == 0 ? a.x : a.dy;

    f32x4 acc1[SL], acc2[SL];
#pragma unroll
    for (int i = 0; i < SL; ++i) acc1[i] = acc2[i] = rat_zero4();
    f32x4 db1a = rat_zero4();                                                // hidden 16 w + 4 g + r, this lane's token columns
    float4 db2f[4] = {zero4, zero4, zero4, zero4};                           // column sums of dy in fragment order (half 1 only)

    // the stationary operands of the chain: hidden tile w of W1 and W2^T (two K steps each), and its bias quad
    const RatB3 w1a = W.w1(w, 0), w1b = W.w1(w, 1), w2a = W.w2t(w, 0), w2b = W.w2t(w, 1);
    const f32x4 b1v = (!DPAD || 16 * w + 4 * g < hid) ? as_v4(ld4(a.b1 + 16 * w + 4 * g)) : rat_zero4();

    float4 tN[4];                                                            // [2 s + part]: src[token][32 s + 8 g + 4 part .. + 3]
    int64_t chunk = blockIdx.x;
    const unsigned period = half == 1 ? (unsigned)a.dy_period : 0u;          // (wave-uniform)
    {
        int64_t tk = chunk * FB_TOK + row;
        bool ok = chunk < a.nchunks && tk < a.ntok;
        if (period != 0) {                                                   // sparse dy: the row exists for tokens t = k P only, at row k
            const unsigned k = (unsigned)tk / period;
            ok = ok && k * period == (unsigned)tk;
            tk = k;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
            tN[q] = (ok && (!DPAD || c < dr)) ? ld4(src + tk * dr + c) : zero4;
        }
    }
    RAT_PROF_DECL
    for (; chunk < a.nchunks; chunk += gridDim.x) {
        {
            float4 tT[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) tT[q] = rat_consume4(tN[q]);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                rat_u4 h, m, lo;
                rat_split8(tT[2 * s], tT[2 * s + 1], h, m, lo);
                if (half == 0) xsp.store(row, 4 * s + g, h, m, lo);
                else dysp.store(row, 4 * s + g, h, m, lo);
            }
            if (half == 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    db2f[q].x += tT[q].x; db2f[q].y += tT[q].y; db2f[q].z += tT[q].z; db2f[q].w += tT[q].w;
                }
        }
        RAT_PROF_MARK(0);
        __syncthreads();
        RAT_PROF_MARK(1);
        // ---- chain: hidden tile w against the four token tiles
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const RatB3 x0 = xsp.row_frag(t, 0), x1 = xsp.row_frag(t, 1), y0 = dysp.row_frag(t, 0), y1 = dysp.row_frag(t, 1);
            f32x4 c0 = b1v, c1 = rat_zero4();
            c0 = RAT_MFMA_BF16(w1a.l, x0.h, c0);
            c1 = RAT_MFMA_BF16(w2a.l, y0.h, c1);
            c0 = RAT_MFMA_BF16(w1a.h, x0.l, c0);
            c1 = RAT_MFMA_BF16(w2a.h, y0.l, c1);
            c0 = RAT_MFMA_BF16(w1a.m, x0.m, c0);
            c1 = RAT_MFMA_BF16(w2a.m, y0.m, c1);
            c0 = RAT_MFMA_BF16(w1a.m, x0.h, c0);
            c1 = RAT_MFMA_BF16(w2a.m, y0.h, c1);
            c0 = RAT_MFMA_BF16(w1a.h, x0.m, c0);
            c1 = RAT_MFMA_BF16(w2a.h, y0.m, c1);
            c0 = RAT_MFMA_BF16(w1a.h, x0.h, c0);
            c1 = RAT_MFMA_BF16(w2a.h, y0.h, c1);
            c0 = RAT_MFMA_BF16(w1b.l, x1.h, c0);
            c1 = RAT_MFMA_BF16(w2b.l, y1.h, c1);
            c0 = RAT_MFMA_BF16(w1b.h, x1.l, c0);
            c1 = RAT_MFMA_BF16(w2b.h, y1.l, c1);
            c0 = RAT_MFMA_BF16(w1b.m, x1.m, c0);
            c1 = RAT_MFMA_BF16(w2b.m, y1.m, c1);
            c0 = RAT_MFMA_BF16(w1b.m, x1.h, c0);
            c1 = RAT_MFMA_BF16(w2b.m, y1.h, c1);
            c0 = RAT_MFMA_BF16(w1b.h, x1.m, c0);
            c1 = RAT_MFMA_BF16(w2b.h, y1.m, c1);
            c0 = RAT_MFMA_BF16(w1b.h, x1.h, c0);
            c1 = RAT_MFMA_BF16(w2b.h, y1.h, c1);
            float4 gv, dp;
            float dg;
            rat_gelu_both(c0[0], gv.x, dg);
            dp.x = c1[0] * dg;
            rat_gelu_both(c0[1], gv.y, dg);
            dp.y = c1[1] * dg;
            rat_gelu_both(c0[2], gv.z, dg);
            dp.z = c1[2] * dg;
            rat_gelu_both(c0[3], gv.w, dg);
            dp.w = c1[3] * dg;
            const HalfPieces gp = f3_split4(gv);
            const HalfPieces dq = f3_split4(dp);
            gsp.store_half(16 * t + n, 4 * w + g, gp.h0, gp.h1, gp.m0, gp.m1, gp.l0, gp.l1);
            dhsp.store_half(16 * t + n, 4 * w + g, dq.h0, dq.h1, dq.m0, dq.m1, dq.l0, dq.l1);
            db1a[0] += dp.x;
            db1a[1] += dp.y;
            db1a[2] += dp.z;
            db1a[3] += dp.w;
        }
        // the dx phase's A operand (W1^T, d tile w & 3, all four K steps: 12 KB per wave from L2) is requested HERE, in front of the
        // barrier: the round trip runs while the wave waits for the slower chains (stamps, round 4: the barrier took 11 % of the
        // iteration and the dx phase, with its fragments requested one step ahead inside the loop, 29 %)
        RAT_SCHED_FENCE();                                           // (not earlier: inside the chain they would be 48 more live registers)
        const RatB3 wt0 = W.w1t(w & 3, 0), wt1 = W.w1t(w & 3, 1), wt2 = W.w1t(w & 3, 2), wt3 = W.w1t(w & 3, 3);
        RAT_SCHED_FENCE();
        RAT_PROF_MARK(2);
        __syncthreads();
        RAT_PROF_MARK(3);
        {   // next chunk's token fragments: in flight behind the dx and weight-gradient GEMMs
            int64_t tk = (chunk + gridDim.x) * FB_TOK + row;
            bool ok = chunk + gridDim.x < a.nchunks && tk < a.ntok;
            if (period != 0) {
                const unsigned k = (unsigned)tk / period;
                ok = ok && k * period == (unsigned)tk;
                tk = k;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
            tN[q] = (ok && (!DPAD || c < dr)) ? ld4(src + tk * dr + c) : zero4;
            }
        }
        // ---- dx^T (d tile md, token tiles t0, t0 + 1) = W1^T dh'^T over the four K steps of the hidden dimension; W1^T's planes carry
        // the hidden index in the stacked-accumulator order (k slot j of lane group g <-> hidden 32 s + 4 g + j | 32 s + 16 + 4 g + j - 4):
        // the B fragment takes those two quads of a token's dh' row
        {
            const int md = w & 3, t0 = 2 * (w >> 2);
            f32x4 dxa[2] = {rat_zero4(), rat_zero4()};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const RatB3& wa = s == 0 ? wt0 : (s == 1 ? wt1 : (s == 2 ? wt2 : wt3));
                RatB3 db[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned ah0, ah1, am0, am1, al0, al1, bh0, bh1, bm0, bm1, bl0, bl1;      // (plain scalars: a struct here lands in scratch)
                    dhsp.load_half(16 * (t0 + i) + n, 8 * s + g, ah0, ah1, am0, am1, al0, al1);
                    dhsp.load_half(16 * (t0 + i) + n, 8 * s + 4 + g, bh0, bh1, bm0, bm1, bl0, bl1);
                    rat_u4 sh, sm, sl;
                    sh.x = ah0; sh.y = ah1; sh.z = bh0; sh.w = bh1;
                    sm.x = am0; sm.y = am1; sm.z = bm0; sm.w = bm1;
                    sl.x = al0; sl.y = al1; sl.z = bl0; sl.w = bl1;
                    db[i] = RatB3{rat_as_bf16x8(sh), rat_as_bf16x8(sm), rat_as_bf16x8(sl)};
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) dxa[i] = RAT_MFMA_BF16(wa.l, db[i].h, dxa[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) dxa[i] = RAT_MFMA_BF16(wa.h, db[i].l, dxa[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) dxa[i] = RAT_MFMA_BF16(wa.m, db[i].m, dxa[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) dxa[i] = RAT_MFMA_BF16(wa.m, db[i].h, dxa[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) dxa[i] = RAT_MFMA_BF16(wa.h, db[i].m, dxa[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) dxa[i] = RAT_MFMA_BF16(wa.h, db[i].h, dxa[i]);
                if (s == 1) RAT_SCHED_FENCE();                               // (all four steps' LDS reads hoisted to the top spill 9 VGPRs)
            }
            // + dy: rebuilt from the three planes of the dy tile (h + m + l IS the fp32 value), then out
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = 16 * (t0 + i) + n;
                const int64_t tok = chunk * FB_TOK + r;
                float4 rsd = zero4;
                if (a.add_dy) {                                               // uniform
                    unsigned h0, h1, m0, m1, l0, l1;
                    dysp.load_half(r, 4 * md + g, h0, h1, m0, m1, l0, l1);
                    rsd = make_float4(rat_join(h0, m0, l0, 0), rat_join(h0, m0, l0, 1), rat_join(h1, m1, l1, 0), rat_join(h1, m1, l1, 1));
                }
                const float4 o = make_float4(dxa[i][0] + rsd.x, dxa[i][1] + rsd.y, dxa[i][2] + rsd.z, dxa[i][3] + rsd.w);
                if (tok < a.ntok && (!DPAD || 16 * md + 4 * g < dr)) st4(a.y + tok * dr + 16 * md + 4 * g, o);
            }
        }
        RAT_PROF_MARK(4);
        // ---- dW1 += dh'^T x (tiles: hidden (w >> 2) + 2 i  x  d (w & 3)) ; dW2 += dy^T gelu(h) (tiles: d (w & 3)  x  hidden (w >> 2) + 2 i)
        {
            const RatB3 xb0 = xsp.col_frag(w & 3, 0), xb1 = xsp.col_frag(w & 3, 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int mt = (w >> 2) + 2 * i;
                acc1[i] = rat_mfma3(dhsp.col_frag(mt, 0), xb0, acc1[i]);
                acc1[i] = rat_mfma3(dhsp.col_frag(mt, 1), xb1, acc1[i]);
            }
            const RatB3 ya0 = dysp.col_frag(w & 3, 0), ya1 = dysp.col_frag(w & 3, 1);
#pragma unroll
            for (int i = 0; i < SL; ++i) {
                const int nt = (w >> 2) + 2 * i;
                acc2[i] = rat_mfma3(ya0, gsp.col_frag(nt, 0), acc2[i]);
                acc2[i] = rat_mfma3(ya1, gsp.col_frag(nt, 1), acc2[i]);
            }
        }
        RAT_PROF_MARK(5);
        __syncthreads();
        RAT_PROF_MARK(6);
    }
    RAT_PROF_FLUSH(a.prof, 72);

    // slab: [dW1 (H x D) | dW2 (D x H) | db1 (H) | db2 (D)]
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
    float* s_w1 = slab;                                           // (the host's layout: [hidden][d], [d][hidden], [hidden], [d])
    float* s_w2 = s_w1 + (int64_t)hid * dr;
    float* s_b1 = s_w2 + (int64_t)dr * hid;
    float* s_b2 = s_b1 + hid;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
        const int t2 = (w >> 2) + 2 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!DPAD || (rat_acc_row(t2, r) < hid && rat_acc_col(w & 3) < dr))
                s_w1[(int64_t)rat_acc_row(t2, r) * dr + rat_acc_col(w & 3)] = acc1[i][r];
            if (!DPAD || (rat_acc_row(w & 3, r) < dr && rat_acc_col(t2) < hid))
                s_w2[(int64_t)rat_acc_row(w & 3, r) * hid + rat_acc_col(t2)] = acc2[i][r];
        }
    }
    {   // bias gradients: db1 — 16 token-column partials per hidden unit; db2 — 64 token-row partials per column; fixed-order sums
        constexpr int L1 = 17, L2 = FB_TOK + 1;
        float* red1 = reinterpret_cast<float*>(smem);              // [H][L1]
        float* red2 = red1 + H * L1;                               // [D][L2]
#pragma unroll
        for (int r = 0; r < 4; ++r) red1[(16 * w + 4 * g + r) * L1 + n] = db1a[r];
        if (half == 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * (q >> 1) + 8 * g + 4 * (q & 1);
                red2[(c0 + 0) * L2 + row] = db2f[q].x;
                red2[(c0 + 1) * L2 + row] = db2f[q].y;
                red2[(c0 + 2) * L2 + row] = db2f[q].z;
                red2[(c0 + 3) * L2 + row] = db2f[q].w;
            }
        __syncthreads();
        for (int c = threadIdx.x; c < H + D; c += FB_THREADS) {
            float sacc = 0.f;
            if (c < H) {
                for (int k = 0; k < 16; ++k) sacc += red1[c * L1 + k];
                if (!DPAD || c < hid) s_b1[c] = sacc;
            } else {
                for (int k = 0; k < FB_TOK; ++k) sacc += red2[(c - H) * L2 + k];
                if (!DPAD || c - H < dr) s_b2[c - H] = sacc;
            }
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the bf16x3 kernels also serve (d, 2 d) for d = 40 / 48 / 56 inside their (64, 128) tiles (DPAD; the shipped KKBox config is (40, 80))
bool f3_dpad_dims(int d, int hidden) { return (d == 40 || d == 48 || d == 56) && hidden == 2 * d; }
bool f3_dims(int d, int hidden) { return (d == F3_D && hidden == F3_H) || f3_dpad_dims(d, hidden); }
bool f3_dpad(const FfnArgs& a, std::initializer_list<const void*> ptrs) {
    if (!f3_dpad_dims(a.d, a.hidden)) return false;
    for (const void* p : ptrs)
        if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return false;
    return (reinterpret_cast<uintptr_t>(a.w1) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w2) & 15) == 0;
}

int ffn_fast_dim(const FfnArgs& a, std::initializer_list<const void*> ptrs) {
    if (a.hidden != 2 * a.d) return 0;              // the compiled fast shapes are (d, 2 d) = (64, 128), (16, 32)
    for (const void* p : ptrs)
        if (!aligned16(p)) return 0;
    if (!aligned16(a.w1) || !aligned16(a.w2)) return 0;
    return (a.d == 64 || a.d == 16) ? a.d : 0;
}

int ffn_check(int64_t ntok, int d, int hidden, bool backward) {
    RAT_REQUIRE(ntok > 0 && d > 0 && hidden > 0, "bad dims");
    RAT_REQUIRE(d <= FFN_THREADS && hidden <= FFN_THREADS, "d / hidden above 512 not supported");
    const FfnGeom g(d, hidden);
    RAT_REQUIRE((backward ? g.bwd_smem() : g.fwd_smem()) <= 160 * 1024, "d*scale_dim too large for the LDS tile");
    if (backward)
        RAT_REQUIRE((g.H16 / 16) * (g.D16 / 16) <= WSLOTS * FFN_WAVES, "hidden x d exceeds the in-register dW accumulator budget");
    return 0;
}

}  // namespace

extern "C" int rat_ffn_fwd(const float* x, float* y, const float* w1, const float* b1, const float* w2, const float* b2,
                           int64_t ntok, int d, int hidden, void* stream) {
    return rat_ffn_fwd_res(x, x, y, w1, b1, w2, b2, ntok, d, hidden, RAT_ARITH_F32, stream);
}

static RatDrop ffn_drop(float p, const uint64_t* seed_dev) {
    RatDrop dr{};
    if (p > 0.f) dr = RatDrop{0, (uint32_t)((double)p * 4294967296.0), 1.0f / (1.0f - p), seed_dev};
    return dr;
}

// FeedForward WITH its two Dropout layers (RAT_m1.py:151-161, RAT_m0.py:150-160: Linear, GELU, Dropout, Linear, Dropout), training mode:
// y = Dropout2(W2 Dropout1(gelu(W1 x + b1)) + b2) + res.  The masks are the counter-based ones of rat_dropout on (seed word, element
// index), seeds read from device memory (rat_dropout_seeds).  Exact fp32 on the generic LDS-staged kernels (no shipped config trains
// with this dropout; the fast kernels carry no mask).
extern "C" int rat_ffn_fwd_drop(const float* x, const float* res, float* y, const float* w1, const float* b1, const float* w2,
                                const float* b2, int64_t ntok, int d, int hidden, float p, const uint64_t* seed1_dev,
                                const uint64_t* seed2_dev, void* stream) {
    if (ffn_check(ntok, d, hidden, false)) return -1;
    RAT_REQUIRE(x && y && w1 && b1 && w2 && b2 && p >= 0.f && p < 1.f && (p == 0.f || (seed1_dev && seed2_dev)), "bad args");
    FfnArgs a{};
    a.x = x;
    a.res = res;
    a.y = y;
    a.w1 = w1;
    a.b1 = b1;
    a.w2 = w2;
    a.b2 = b2;
    a.ntok = ntok;
    a.nchunks = (ntok + FFN_ROWS - 1) / FFN_ROWS;
    a.d = d;
    a.hidden = hidden;
    a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(y);
    a.vec_w1 = (d % 4 == 0) && aligned16(w1);
    a.vec_w2 = (hidden % 4 == 0) && aligned16(w2);
    a.prof = rat_prof_buffer();
    a.drop1 = ffn_drop(p, seed1_dev);
    a.drop2 = ffn_drop(p, seed2_dev);
    const FfnGeom g(d, hidden);
    const int64_t blocks = a.nchunks < rat_max_blocks() ? a.nchunks : rat_max_blocks();
    RAT_LAUNCH((ffn_fwd_kernel<0>), (unsigned)blocks, FFN_THREADS, g.fwd_smem(), stream, a);
    return rat_check_launch("rat_ffn_fwd_drop");
}

extern "C" int rat_ffn_bwd_drop(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2,
                                const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                                size_t workspace_bytes, int64_t ntok, int d, int hidden, int add_dy, float p,
                                const uint64_t* seed1_dev, const uint64_t* seed2_dev, void* stream) {
    if (ffn_check(ntok, d, hidden, true)) return -1;
    RAT_REQUIRE(x && dy && dx && w1 && b1 && w2 && b2 && dw1 && db1 && dw2 && db2 && workspace, "null pointer");
    RAT_REQUIRE(workspace_bytes >= rat_ffn_bwd_workspace(d, hidden), "workspace too small");
    RAT_REQUIRE(p >= 0.f && p < 1.f && (p == 0.f || (seed1_dev && seed2_dev)), "bad dropout arguments");
    FfnArgs a{};
    a.x = x;
    a.dy = dy;
    a.y = dx;
    a.add_dy = add_dy;
    a.w1 = w1;
    a.b1 = b1;
    a.w2 = w2;
    a.b2 = b2;
    a.ntok = ntok;
    a.nchunks = (ntok + FFN_ROWS - 1) / FFN_ROWS;
    a.d = d;
    a.hidden = hidden;
    a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(dx);
    a.vec_w1 = (d % 4 == 0) && aligned16(w1);
    a.vec_w2 = (hidden % 4 == 0) && aligned16(w2);
    a.prof = rat_prof_buffer();
    a.drop1 = ffn_drop(p, seed1_dev);
    a.drop2 = ffn_drop(p, seed2_dev);
    const FfnGeom g(d, hidden);
    a.slabs = workspace;
    a.slab_stride = g.slab_floats();
    const int blocks = (int)(a.nchunks < rat_max_blocks() ? a.nchunks : rat_max_blocks());
    RAT_LAUNCH((ffn_bwd_kernel<>), blocks, FFN_THREADS, g.bwd_smem(), stream, a);
    if (rat_check_launch("rat_ffn_bwd_drop")) return -1;
    float* outs[4] = {dw1, dw2, db1, db2};
    const int64_t sizes[4] = {(int64_t)hidden * d, (int64_t)d * hidden, hidden, d};
    const int64_t offs[4] = {0, (int64_t)hidden * d, 2 * (int64_t)hidden * d, 2 * (int64_t)hidden * d + hidden};
    return rat_launch_reduce_slabs(workspace, blocks, a.slab_stride, outs, offs, sizes, 4, stream);
}

extern "C" int rat_ffn_fwd_res(const float* x, const float* res, float* y, const float* w1, const float* b1, const float* w2,
                               const float* b2, int64_t ntok, int d, int hidden, int arith, void* stream) {
    if (ffn_check(ntok, d, hidden, false)) return -1;
    RAT_REQUIRE(x && y && w1 && b1 && w2 && b2, "null pointer");
    FfnArgs a{};
    a.x = x;
    a.res = res;
    a.y = y;
    a.w1 = w1;
    a.b1 = b1;
    a.w2 = w2;
    a.b2 = b2;
    a.ntok = ntok;
    a.nchunks = (ntok + FFN_ROWS - 1) / FFN_ROWS;
    a.d = d;
    a.hidden = hidden;
    a.vec_x = (d % 4 == 0) && aligned16(x);
    a.vec_w1 = (d % 4 == 0) && aligned16(w1);
    a.vec_w2 = (hidden % 4 == 0) && aligned16(w2);
    a.prof = rat_prof_buffer();
    const FfnGeom g(d, hidden);
    const size_t smem = g.fwd_smem();
    int per_cu = (int)((160 * 1024) / smem);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    const int64_t blocks = a.nchunks < rat_max_blocks() * per_cu ? a.nchunks : rat_max_blocks() * per_cu;
    const int fast = ffn_fast_dim(a, {x, y, b1, b2, res});
    const int64_t wtiles = ((ntok + 15) / 16 + FT_WAVES - 1) / FT_WAVES;
    const unsigned tgrid = (unsigned)(wtiles < rat_max_blocks() ? wtiles : rat_max_blocks());
    const bool xres = res == x;
    if (arith == RAT_ARITH_BF16X3 && f3_dpad(a, {x, y, b1, b2, res})) {
        if (xres) RAT_LAUNCH((ffn_fwd_t3_kernel<true, true>), tgrid, FT_THREADS, f3_fwd_smem(), stream, a);
        else RAT_LAUNCH((ffn_fwd_t3_kernel<false, true>), tgrid, FT_THREADS, f3_fwd_smem(), stream, a);
    } else if (fast == 64 && hidden == 128 && arith == RAT_ARITH_BF16X3) {
        if (xres) RAT_LAUNCH((ffn_fwd_t3_kernel<true>), tgrid, FT_THREADS, f3_fwd_smem(), stream, a);
        else RAT_LAUNCH((ffn_fwd_t3_kernel<false>), tgrid, FT_THREADS, f3_fwd_smem(), stream, a);
    } else if (fast == 64 && hidden == 128) {
        if (xres) RAT_LAUNCH((ffn_fwd_t_kernel<64, 128, true>), tgrid, FT_THREADS, (FfnTGeom<64, 128>::fwd_smem), stream, a);
        else RAT_LAUNCH((ffn_fwd_t_kernel<64, 128, false>), tgrid, FT_THREADS, (FfnTGeom<64, 128>::fwd_smem), stream, a);
    } else if (fast == 16 && hidden == 32) {
        if (xres) RAT_LAUNCH((ffn_fwd_t_kernel<16, 32, true>), tgrid, FT_THREADS, (FfnTGeom<16, 32>::fwd_smem), stream, a);
        else RAT_LAUNCH((ffn_fwd_t_kernel<16, 32, false>), tgrid, FT_THREADS, (FfnTGeom<16, 32>::fwd_smem), stream, a);
    } else if (d == 10 && hidden == 40) {                       // shipped MovieLens geometry (+ the LDS copies of the weights)
        RAT_LAUNCH((ffn_fwd_kernel<0, 10, 40>), (unsigned)blocks, FFN_THREADS, smem + (size_t)ffn_small_w_floats(40) * 4, stream, a);
    } else if (d == 10 && hidden == 20) {                       // shipped Tmall geometry
        RAT_LAUNCH((ffn_fwd_kernel<0, 10, 20>), (unsigned)blocks, FFN_THREADS, smem + (size_t)ffn_small_w_floats(20) * 4, stream, a);
    } else if (d == 16 && hidden == 64) {                       // BASELINE configs[0] (d = 16, scale_dim 4)
        RAT_LAUNCH((ffn_fwd_kernel<0, 16, 64>), (unsigned)blocks, FFN_THREADS, smem, stream, a);
    } else {
        RAT_LAUNCH((ffn_fwd_kernel<0>), (unsigned)blocks, FFN_THREADS, smem, stream, a);
    }
    return rat_check_launch("rat_ffn_fwd");
}

extern "C" size_t rat_ffn_bwd_workspace(int d, int hidden) {
    const FfnGeom g(d, hidden);
    const size_t fp32 = ((size_t)256 * (size_t)g.slab_floats() + 2 * (size_t)d * hidden) * sizeof(float);   // slabs + w1^T + w2^T
    return fp32 + (f3_dims(d, hidden) ? 3 * F3_WP + 16 : 0);                                                 // + bf16x3 weight fragments
}

// `planes` of rat_ffn_bwd_res: [W1 | W2^T | W1^T (hidden index permuted)] fragment planes
extern "C" size_t rat_ffn_planes_bytes(int d, int hidden) { return f3_dims(d, hidden) ? 3 * F3_WP : 0; }
extern "C" int rat_ffn_split_jobs(const float* w1, const float* w2, int d, int hidden, void* planes, RatSplitJob* jobs_out) {
    RAT_REQUIRE(w1 && w2 && jobs_out, "null pointer");
    if (!f3_dims(d, hidden) || planes == nullptr) return 0;
    RAT_REQUIRE(aligned16(planes), "planes must be 16-byte aligned");
    char* ws = static_cast<char*>(planes);
    // narrower layers: the planes cover (128, 64); `reserved` = n_valid | k_valid << 16 says how much of them is the matrix
    const int pad = d != F3_D;
    jobs_out[0] = RatSplitJob{w1, ws, F3_H, d, d, 0, 0, pad ? hidden : 0};                                  // A[hidden][d]   = w1[hidden][d]
    jobs_out[1] = RatSplitJob{w2, ws + F3_WP, F3_H, d, hidden, 1, 0, pad ? hidden : 0};                     // A[hidden][d]   = w2[d][hidden]
    jobs_out[2] = RatSplitJob{w1, ws + 2 * F3_WP, F3_D, F3_H, d, 1, 1, pad ? (d | (hidden << 16)) : 0};     // A[d][hidden*] = w1[hidden][d]
    return 3;
}

extern "C" int rat_ffn_bwd(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2,
                           const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                           size_t workspace_bytes, int64_t ntok, int d, int hidden, void* stream) {
    return rat_ffn_bwd_res(x, dy, dx, w1, b1, w2, b2, dw1, db1, dw2, db2, workspace, workspace_bytes, nullptr, ntok, d, hidden, 1, RAT_ARITH_F32,
                           stream);
}

static int ffn_bwd_res_launch(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2,
                              const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                              size_t workspace_bytes, const void* planes, int64_t ntok, int d, int hidden, int add_dy, int arith,
                              int64_t dy_period, void* stream);

extern "C" int rat_ffn_bwd_res(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                               size_t workspace_bytes, const void* planes, int64_t ntok, int d, int hidden, int add_dy, int arith,
                               void* stream) {
    return ffn_bwd_res_launch(x, dy, dx, w1, b1, w2, b2, dw1, db1, dw2, db2, workspace, workspace_bytes, planes, ntok, d, hidden, add_dy,
                              arith, 0, stream);
}

// 1 if rat_ffn_bwd_res_rows accepts this layer (the bf16x3 weight-stationary kernel: d = 64 or 40 / 48 / 56, hidden = 2 d)
extern "C" int rat_ffn_bwd_rows_supported(int d, int hidden, int arith) {
    const bool t3 = rat_knob(RAT_KNOB_FFN_BWD_T3) == 1;
    if (t3 || arith != RAT_ARITH_BF16X3) return 0;
    return (d == F3_D && hidden == F3_H) || ((d == 40 || d == 48 || d == 56) && hidden == 2 * d);
}

// rat_ffn_bwd_res for a gradient that is zero except on the rows t = k * dy_period: dy_rows [ceil(ntok / dy_period)][d] holds those rows
// compactly; the zero rows are neither stored nor read (no [ntok][d] zero fill in front of the last block's backward).
extern "C" int rat_ffn_bwd_res_rows(const float* x, const float* dy_rows, int64_t dy_period, float* dx, const float* w1, const float* b1,
                                    const float* w2, const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                                    size_t workspace_bytes, const void* planes, int64_t ntok, int d, int hidden, int add_dy, int arith,
                                    void* stream) {
    RAT_REQUIRE(dy_period > 0 && dy_period < ((int64_t)1 << 31) && ntok < ((int64_t)1 << 31), "bad dy_period");
    RAT_REQUIRE(rat_ffn_bwd_rows_supported(d, hidden, arith), "rat_ffn_bwd_res_rows: geometry without the bf16x3 weight-stationary kernel");
    return ffn_bwd_res_launch(x, dy_rows, dx, w1, b1, w2, b2, dw1, db1, dw2, db2, workspace, workspace_bytes, planes, ntok, d, hidden,
                              add_dy, arith, dy_period, stream);
}

static int ffn_bwd_res_launch(const float* x, const float* dy, float* dx, const float* w1, const float* b1, const float* w2,
                              const float* b2, float* dw1, float* db1, float* dw2, float* db2, float* workspace,
                              size_t workspace_bytes, const void* planes, int64_t ntok, int d, int hidden, int add_dy, int arith,
                              int64_t dy_period, void* stream) {
    if (ffn_check(ntok, d, hidden, true)) return -1;
    RAT_REQUIRE(x && dy && dx && w1 && b1 && w2 && b2 && dw1 && db1 && dw2 && db2 && workspace, "null pointer");
    RAT_REQUIRE(workspace_bytes >= rat_ffn_bwd_workspace(d, hidden), "workspace too small");
    FfnArgs a{};
    a.x = x;
    a.dy = dy;
    a.y = dx;
    a.add_dy = add_dy;
    a.w1 = w1;
    a.b1 = b1;
    a.w2 = w2;
    a.b2 = b2;
    a.ntok = ntok;
    a.nchunks = (ntok + FFN_ROWS - 1) / FFN_ROWS;
    a.d = d;
    a.hidden = hidden;
    a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(dy);
    a.vec_w1 = (d % 4 == 0) && aligned16(w1);
    a.vec_w2 = (hidden % 4 == 0) && aligned16(w2);
    a.prof = rat_prof_buffer();
    const FfnGeom g(d, hidden);
    a.slabs = workspace;
    a.slab_stride = g.slab_floats();
    const int blocks = (int)(a.nchunks < rat_max_blocks() ? a.nchunks : rat_max_blocks());
    const int fast = ffn_fast_dim(a, {x, dy, dx, b1});
    const bool dpad = arith == RAT_ARITH_BF16X3 && f3_dpad(a, {x, dy, dx, b1});
    const bool b3 = dpad || (fast == F3_D && hidden == F3_H && arith == RAT_ARITH_BF16X3);
    a.dy_period = (int)dy_period;
    RAT_REQUIRE(dy_period == 0 || b3, "sparse dy rows: the pointers do not allow the bf16x3 kernel");
    if (b3) {
        const char* ws;
        if (planes != nullptr && aligned16(planes)) {            // split once per step by the caller (rat_split_weights_batch)
            ws = static_cast<const char*>(planes);
        } else {
            uintptr_t wsb = reinterpret_cast<uintptr_t>(workspace + (size_t)256 * a.slab_stride + 2 * (size_t)d * hidden);
            char* wsw = reinterpret_cast<char*>((wsb + 15) & ~(uintptr_t)15);
            if (rat_launch_split_weights(w1, F3_H, d, d, 0, wsw, stream, 0, dpad ? hidden : 0) ||                           // A[hidden][d]   = w1[hidden][d]
                rat_launch_split_weights(w2, F3_H, d, hidden, 1, wsw + F3_WP, stream, 0, dpad ? hidden : 0) ||              // A[hidden][d]   = w2[d][hidden]
                rat_launch_split_weights(w1, F3_D, F3_H, d, 1, wsw + 2 * F3_WP, stream, 1, dpad ? (d | (hidden << 16)) : 0))  // A[d][hidden*] = w1[hidden][d]
                return -1;
            ws = wsw;
        }
        Ffn3W W{};
        W.w1 = RatWPlanes{reinterpret_cast<const rat_u4*>(ws), 2};
        W.w2t = RatWPlanes{reinterpret_cast<const rat_u4*>(ws + F3_WP), 2};
        W.w1t = RatWPlanes{reinterpret_cast<const rat_u4*>(ws + 2 * F3_WP), 4};
        const bool t3 = rat_knob(RAT_KNOB_FFN_BWD_T3) == 1;
        if (t3) {                                                 // the round-2/3 kernel (token-stationary), kept for A/Bs: RAT_FFN_BWD=t3
            if (dpad) RAT_LAUNCH((ffn_bwd_t3_kernel<true>), blocks, FB_THREADS, f3_bwd_smem(), stream, a, W);
            else RAT_LAUNCH((ffn_bwd_t3_kernel<false>), blocks, FB_THREADS, f3_bwd_smem(), stream, a, W);
        } else if (dpad) RAT_LAUNCH((ffn_bwd_t4_kernel<true>), blocks, FB_THREADS, f3_bwd_smem(), stream, a, W);
        else RAT_LAUNCH((ffn_bwd_t4_kernel<false>), blocks, FB_THREADS, f3_bwd_smem(), stream, a, W);
    } else
    if (fast) {
        float* w1t = workspace + (size_t)256 * a.slab_stride;
        float* w2t = w1t + (size_t)d * hidden;
        if (rat_launch_transpose(w1, w1t, hidden, d, stream) || rat_launch_transpose(w2, w2t, d, hidden, stream)) return -1;
        a.w1t = w1t;
        a.w2t = w2t;
    }
    if (b3) {
        // launched above
    } else if (fast == 64 && hidden == 128) {
        RAT_LAUNCH((ffn_bwd_t_kernel<64, 128>), blocks, FB_THREADS, (FfnBTGeom<64, 128>::smem), stream, a);
    } else if (fast == 16 && hidden == 32) {
        RAT_LAUNCH((ffn_bwd_t_kernel<16, 32>), blocks, FB_THREADS, (FfnBTGeom<16, 32>::smem), stream, a);
    } else if (d == 10 && hidden == 40) {
        RAT_LAUNCH((ffn_bwd_kernel<10, 40>), blocks, FFN_THREADS, g.bwd_smem() + (size_t)ffn_small_w_floats(40) * 4, stream, a);
    } else if (d == 10 && hidden == 20) {
        RAT_LAUNCH((ffn_bwd_kernel<10, 20>), blocks, FFN_THREADS, g.bwd_smem() + (size_t)ffn_small_w_floats(20) * 4, stream, a);
    } else if (d == 16 && hidden == 64) {
        RAT_LAUNCH((ffn_bwd_kernel<16, 64>), blocks, FFN_THREADS, g.bwd_smem(), stream, a);
    } else {
        RAT_LAUNCH((ffn_bwd_kernel<>), blocks, FFN_THREADS, g.bwd_smem(), stream, a);
    }
    if (rat_check_launch("rat_ffn_bwd")) return -1;
    float* outs[4] = {dw1, dw2, db1, db2};
    const int64_t sizes[4] = {(int64_t)hidden * d, (int64_t)d * hidden, hidden, d};
    const int64_t offs[4] = {0, (int64_t)hidden * d, 2 * (int64_t)hidden * d, 2 * (int64_t)hidden * d + hidden};
    return rat_launch_reduce_slabs(workspace, blocks, a.slab_stride, outs, offs, sizes, 4, stream);
}
