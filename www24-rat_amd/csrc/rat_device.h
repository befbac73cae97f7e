// rat_device.h — device-side building blocks shared by the RAT_m2 kernels (gfx950 / CDNA4).
//
// Everything here is written for 64-lane wavefronts and the v_mfma_f32_16x16x4_f32 matrix instruction
// (exact fp32, bit-identical to a k-ordered fmaf chain — which is what keeps the kernels inside the
// reference's fp32 tolerance).  Lane maps (cdna_hip_programming.md §3):
//     A: lane l holds A[i = l & 15][k = l >> 4]      B: lane l holds B[k = l >> 4][j = l & 15]
//     C/D: col = l & 15, row = (l >> 4) * 4 + reg
// The contraction index of one 16-wide "k-block" is permuted so that a lane's four values are CONTIGUOUS
// in memory: MFMA step j of lane group g = l >> 4 consumes k = 16*kb + 4*g + j.  A and B use the same
// permutation, so the sum is unchanged, and row-major operands are fetched with one 16-byte load.
#pragma once

#include <stdint.h>
#include <stddef.h>

#ifdef RAT_EMU
#include "hip_emu.h"
#define RAT_MFMA16(a, b, c) emu_mfma_f32_16x16x4f32((a), (b), (c))
#define RAT_MFMA4(a, b, c) emu_mfma_f32_4x4x1f32((a), (b), (c))
#else
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((vector_size(16)));
#define RAT_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4x1 outer products per wave (block = lane / 4):
//   D[reg r][lane l] += A(lane 4*(l/4) + r) * B(lane l) — the shape of one (sequence, head) attention block
#define RAT_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#define RAT_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define RAT_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#include <string>

// ---- optional in-kernel phase stamps (diagnostic build only: -DRAT_PROF -> librat_hip_prof.so, never the product).
// Thread 0 of every work-group accumulates s_memtime deltas per phase and adds them to a debug buffer that no other
// code reads (cdna_hip_programming.md §7 "In-kernel stamps").
#if defined(RAT_PROF) && !defined(RAT_EMU)
#define RAT_PROF_DECL unsigned long long prof_t0 = clock64(); unsigned long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RAT_PROF_MARK(i) do { const unsigned long long prof_t = clock64(); prof_acc[i] += prof_t - prof_t0; prof_t0 = prof_t; } while (0)
#define RAT_PROF_FLUSH(ptr, base) do { if (threadIdx.x == 0 && (ptr) != nullptr) for (int pi = 0; pi < 12; ++pi) atomicAdd((ptr) + (base) + pi, prof_acc[pi]); } while (0)
// per-wave timeline of ONE chunk iteration of work-group 0 (trace slots live behind the 256 phase sums)
#define RAT_TRACE(ptr, on, slot) do { if ((ptr) != nullptr && (on) && blockIdx.x == 0 && (threadIdx.x & 63) == 0) (ptr)[256 + (threadIdx.x >> 6) * 32 + (slot)] = clock64(); } while (0)
#else
#define RAT_PROF_DECL
#define RAT_PROF_MARK(i) do { } while (0)
#define RAT_PROF_FLUSH(ptr, base) do { } while (0)
#define RAT_TRACE(ptr, on, slot) do { } while (0)
#endif
// lanes of ONE wave exchanging data through a wave-private LDS tile: order the wave's own LDS writes before its reads (no
// work-group barrier).  Emulation: the wave's lanes are OS threads — a real wave barrier.
#ifdef RAT_EMU
#define RAT_WAVE_FENCE() emu::wave_sync()
#else
#define RAT_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
unsigned long long* rat_prof_buffer();                // device pointer set by rat_debug_set_prof (nullptr by default)

// ------------------------------------------------------------------------------------------- host side
const char* rat_set_error(const std::string& msg);   // stores thread-local, returns c_str
int rat_fail(const std::string& msg);                 // sets error, returns -1
int rat_check_launch(const char* what);               // hipGetLastError -> 0 / -1
enum { RAT_KNOB_MAX_BLOCKS, RAT_KNOB_ATTN_BWD_PH, RAT_KNOB_ATTN_FWD_CORE_MFMA, RAT_KNOB_ATTN_BWD_CORE_MFMA, RAT_KNOB_FFN_BWD_T3,
       RAT_KNOB_SGEMM_SPLIT_TARGET, RAT_KNOB_COUNT };
int rat_knob(int which);                              // diagnostic knobs (rat_common.hip): read from the environment ONCE, at load
int rat_max_blocks();                                 // 256 (one work-group per CU) unless the RAT_MAX_BLOCKS test knob lowers it

#define RAT_REQUIRE(cond, msg)                  \
    do {                                        \
        if (!(cond)) return rat_fail(std::string(__func__) + ": " + (msg)); \
    } while (0)

static inline int rat_round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------- device side
#define RAT_WAVE 64
// scheduling fence: keeps hipcc from hoisting the next tile's operand loads (and their registers) across this point
// RAT_SCHED_MFMA_VALU(n, v): inside the current scheduling region, order the instruction stream as n x {1 MFMA, v VALU}
// (cdna_hip_programming.md T19): the VALU work of an independent tile (GELU, softmax) issues in the shadow of the MFMAs.
#ifdef RAT_EMU
#define RAT_SCHED_FENCE() do { } while (0)
#define RAT_SCHED_MFMA_VALU(n, v) do { } while (0)
#else
#define RAT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define RAT_SCHED_MFMA_VALU(n, v)                                   \
    do {                                                            \
        _Pragma("unroll") for (int sgb_i = 0; sgb_i < (n); ++sgb_i) { \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
            __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);    \
        }                                                           \
    } while (0)
#endif

__device__ __forceinline__ int rat_lane() { return threadIdx.x & 63; }
// wave index, PROVABLY wave-uniform for the compiler: anything derived from threadIdx is divergent to hipcc (even
// threadIdx.x >> 6), which turns every per-wave task loop / tile predicate into EXEC-masked branches around each MFMA;
// readfirstlane makes it an SGPR value and those branches scalar (cdna_hip_programming.md T5/T20 notes).
#ifdef RAT_EMU
__device__ __forceinline__ int rat_wave() { return threadIdx.x >> 6; }
#else
__device__ __forceinline__ int rat_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#endif

// Hand a prefetched (global-load) register set over to its consumer with REAL moves.  hipcc's s_waitcnt insertion is
// conservative across a loop back-edge: a value loaded in iteration i and first used in iteration i+1 is waited for with
// vmcnt(0) at that use, i.e. AFTER the next prefetch has been issued — which serialises the prefetch it was meant to hide.
// Copying at a point where every outstanding vector-memory operation is old makes the (free) wait happen there, and the
// copy's destination carries no pending-load state.
__device__ __forceinline__ float4 rat_consume4(const float4& src) {
#ifdef RAT_EMU
    return src;
#else
    float4 d;
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(d.x), "=&v"(d.y), "=&v"(d.z), "=&v"(d.w)
                 : "v"(src.x), "v"(src.y), "v"(src.z), "v"(src.w));
    return d;
#endif
}

// counter-based dropout generator shared by rat_dropout (optim.hip) and the attention output-projection dropout (attn.hip):
// element i of a tensor is kept iff rat_hash32(seed, i) >= p * 2^32 — a pure function of (seed, i), so backward re-derives the mask
__device__ __forceinline__ uint32_t rat_hash32(uint64_t seed, uint64_t i) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);          // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
struct RatDrop {                     // threshold == 0: no dropout
    uint64_t seed;
    uint32_t threshold;
    float scale;                     // 1 / (1 - p)
    const uint64_t* seed_dev;        // ABI v6: when set, the seed is read from device memory (rat_dropout_seeds refreshes it once per
                                     // training step, so a captured step replays with a new mask every time); `seed` is ignored
    __device__ __forceinline__ uint64_t the_seed() const { return seed_dev != nullptr ? *seed_dev : seed; }
    __device__ __forceinline__ float apply(float v, int64_t idx) const {
        return threshold == 0 ? v : (rat_hash32(the_seed(), (uint64_t)idx) >= threshold ? v * scale : 0.f);
    }
};

// 16-byte accesses with the non-temporal hint, for data that is written (read) once and not touched again for a long time — saved
// activations, optimizer moments: they should not evict what the next kernel is about to re-read from L2 / Infinity Cache
__device__ __forceinline__ void rat_st4_stream(float* p, const float4& v) {
#ifdef RAT_EMU
    *reinterpret_cast<float4*>(p) = v;
#else
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#endif
}
__device__ __forceinline__ float4 rat_ld4_stream(const float* p) {
#ifdef RAT_EMU
    return *reinterpret_cast<const float4*>(p);
#else
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
#endif
}

__device__ __forceinline__ f32x4 rat_zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}

// sum over groups of `width` consecutive lanes (width a power of two <= 64); every lane gets the result
template <int WIDTH>
__device__ __forceinline__ float rat_group_sum(float v) {
#pragma unroll
    for (int m = WIDTH / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// softmax runs in the log2 domain: one v_exp_f32 / v_log_f32 per call (about 1 ulp), no range-reduction sequence
#ifdef RAT_EMU
__device__ __forceinline__ float rat_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float rat_log2(float x) { return log2f(x); }
#else
__device__ __forceinline__ float rat_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rat_log2(float x) { return __builtin_amdgcn_logf(x); }
#endif
#define RAT_LOG2E 1.44269504088896340736f

// erf(x), fp32, max error 1.12 ulp / 6.7e-8 absolute over [-6, 6] (both ranges are evaluated and selected: no divergence;
// ~22 VALU instructions against ~55 for ocml's erff).  Polynomials: N. Juffa's single-precision minimax fits
// (|x| <= 0.9277: x + x*P(x^2);  above: 1 - exp(Q(|x|))), exp through the hardware exp2.
__device__ __forceinline__ float rat_erf(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - rat_exp2(r * RAT_LOG2E), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}
// Two-at-a-time variants: the polynomial parts run on v_pk_fma_f32 / v_pk_mul_f32 (plain fp32 VALU instructions cost 4
// cycles per wave on gfx950 — the packed forms do twice the work in the same slot); per-component arithmetic is identical
// to rat_erf (IEEE fma per lane), so scalar and packed paths agree bit for bit.
#ifdef RAT_EMU
struct rat_f2 { float x, y; };
__device__ __forceinline__ rat_f2 rat_f2_make(float a, float b) { return rat_f2{a, b}; }
__device__ __forceinline__ rat_f2 rat_pk_fma(rat_f2 a, rat_f2 b, rat_f2 c) { return rat_f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ rat_f2 rat_pk_mul(rat_f2 a, rat_f2 b) { return rat_f2{a.x * b.x, a.y * b.y}; }
#else
typedef float rat_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rat_f2 rat_f2_make(float a, float b) { rat_f2 r = {a, b}; return r; }
__device__ __forceinline__ rat_f2 rat_pk_fma(rat_f2 a, rat_f2 b, rat_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ rat_f2 rat_pk_mul(rat_f2 a, rat_f2 b) { return a * b; }
#endif
__device__ __forceinline__ rat_f2 rat_f2_splat(float a) { return rat_f2_make(a, a); }
__device__ __forceinline__ rat_f2 rat_erf2(rat_f2 a) {
    const rat_f2 t = rat_f2_make(fabsf(a.x), fabsf(a.y)), s = rat_pk_mul(a, a);
    rat_f2 r = rat_pk_fma(rat_f2_splat(-1.72853470e-5f), t, rat_f2_splat(3.83197126e-4f));
    const rat_f2 u = rat_pk_fma(rat_f2_splat(-3.88396438e-3f), t, rat_f2_splat(2.42546219e-2f));
    r = rat_pk_fma(r, s, u);
    r = rat_pk_fma(r, t, rat_f2_splat(-1.06777877e-1f));
    r = rat_pk_fma(r, t, rat_f2_splat(-6.34846687e-1f));
    r = rat_pk_fma(r, t, rat_f2_splat(-1.28717512e-1f));
    r = rat_pk_fma(r, t, rat_f2_make(-t.x, -t.y));
    r = rat_pk_mul(r, rat_f2_splat(RAT_LOG2E));
    const float bx = copysignf(1.0f - rat_exp2(r.x), a.x), by = copysignf(1.0f - rat_exp2(r.y), a.y);
    rat_f2 q = rat_f2_splat(-5.96761703e-4f);
    q = rat_pk_fma(q, s, rat_f2_splat(4.99119423e-3f));
    q = rat_pk_fma(q, s, rat_f2_splat(-2.67681349e-2f));
    q = rat_pk_fma(q, s, rat_f2_splat(1.12819925e-1f));
    q = rat_pk_fma(q, s, rat_f2_splat(-3.76125336e-1f));
    q = rat_pk_fma(q, s, rat_f2_splat(1.28379166e-1f));
    const rat_f2 sm = rat_pk_fma(q, a, a);
    return rat_f2_make(t.x > 0.927734375f ? bx : sm.x, t.y > 0.927734375f ? by : sm.y);
}
// gelu of two values
__device__ __forceinline__ rat_f2 rat_gelu2(rat_f2 x) {
    const rat_f2 e = rat_erf2(rat_pk_mul(x, rat_f2_splat(0.70710678118654752440f)));
    return rat_pk_mul(rat_pk_mul(rat_f2_splat(0.5f), x), rat_f2_make(1.0f + e.x, 1.0f + e.y));
}
// gelu and gelu' of two values from one erf evaluation each
__device__ __forceinline__ void rat_gelu_both2(rat_f2 x, rat_f2& g, rat_f2& dg) {
    const rat_f2 e = rat_erf2(rat_pk_mul(x, rat_f2_splat(0.70710678118654752440f)));
    const rat_f2 cdf = rat_pk_mul(rat_f2_splat(0.5f), rat_f2_make(1.0f + e.x, 1.0f + e.y));
    const rat_f2 ex = rat_pk_mul(rat_pk_mul(x, x), rat_f2_splat(-0.5f * RAT_LOG2E));
    const rat_f2 pdf = rat_pk_mul(rat_f2_splat(0.39894228040143267794f), rat_f2_make(rat_exp2(ex.x), rat_exp2(ex.y)));
    g = rat_pk_mul(x, cdf);
    dg = rat_pk_fma(x, pdf, cdf);
}

__device__ __forceinline__ float rat_gelu(float x) {          // nn.GELU() exact erf form
    return 0.5f * x * (1.0f + rat_erf(x * 0.70710678118654752440f));
}
// gelu(x) and d/dx gelu(x) from ONE erf evaluation (the backward FFN needs both for every hidden activation)
__device__ __forceinline__ void rat_gelu_both(float x, float& g, float& dg) {
    const float cdf = 0.5f * (1.0f + rat_erf(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * rat_exp2(x * x * (-0.5f * RAT_LOG2E));
    g = x * cdf;
    dg = fmaf(x, pdf, cdf);
}
__device__ __forceinline__ float rat_gelu_grad(float x) {     // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    const float cdf = 0.5f * (1.0f + rat_erf(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * rat_exp2(x * x * (-0.5f * RAT_LOG2E));
    return fmaf(x, pdf, cdf);
}

// ---- MFMA operand fetchers.  Each returns the lane's 4 values of k-block `kb` for 16-row/col tile `tile`.
// A[m][k] or B[n][k]-as-weights, row-major in LDS with the contraction index contiguous (16-byte reads).
struct RatLdsRows {
    const float* base;
    int ld;                                   // floats, multiple of 4; base 16-byte aligned
    __device__ __forceinline__ float4 operator()(int tile, int kb) const {
        const int l = rat_lane();
        return *reinterpret_cast<const float4*>(base + (size_t)(tile * 16 + (l & 15)) * ld + kb * 16 + 4 * (l >> 4));
    }
};
// operand stored with the contraction index as the ROW of an LDS tile: X[k][m] (token-contraction GEMMs)
struct RatLdsCols {
    const float* base;
    int ld;
    __device__ __forceinline__ float4 operator()(int tile, int kb) const {
        const int l = rat_lane();
        const float* p = base + (size_t)(kb * 16 + 4 * (l >> 4)) * ld + tile * 16 + (l & 15);
        return make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
    }
};
// nn.Linear weight W[N][K] in global memory used as B[k][n] = W[n][k]  (y = x W^T).
// GUARD=false: N, K multiples of 16, ld % 4 == 0, 16-byte aligned base -> one unguarded 16-byte load.
template <bool GUARD>
struct RatGlobalWnkT {
    const float* w;
    int N, K, ld;
    bool vec;                                 // (GUARD only) K % 4 == 0 && ld % 4 == 0 && 16-byte aligned base
    __device__ __forceinline__ float4 operator()(int tile, int kb) const {
        const int l = rat_lane();
        const int n = tile * 16 + (l & 15);
        const int k = kb * 16 + 4 * (l >> 4);
        const float* p = w + (size_t)n * ld + k;
        if (!GUARD) return *reinterpret_cast<const float4*>(p);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) {
            if (vec && k + 3 < K) {
                r = *reinterpret_cast<const float4*>(p);
            } else {
                if (k + 0 < K) r.x = p[0];
                if (k + 1 < K) r.y = p[1];
                if (k + 2 < K) r.z = p[2];
                if (k + 3 < K) r.w = p[3];
            }
        }
        return r;
    }
};
typedef RatGlobalWnkT<true> RatGlobalWnk;
// weight W[K][N] in global memory used as B[k][n] = W[k][n]  (dx = dy W)
template <bool GUARD>
struct RatGlobalWknT {
    const float* w;
    int K, N, ld;
    __device__ __forceinline__ float4 operator()(int tile, int kb) const {
        const int l = rat_lane();
        const int n = tile * 16 + (l & 15);
        const int k = kb * 16 + 4 * (l >> 4);
        const float* p = w + (size_t)k * ld + n;
        if (!GUARD) return make_float4(p[0], p[(size_t)ld], p[(size_t)2 * ld], p[(size_t)3 * ld]);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) {
            if (k + 0 < K) r.x = p[0];
            if (k + 1 < K) r.y = p[(size_t)ld];
            if (k + 2 < K) r.z = p[(size_t)2 * ld];
            if (k + 3 < K) r.w = p[(size_t)3 * ld];
        }
        return r;
    }
};
typedef RatGlobalWknT<true> RatGlobalWkn;

// acc[i] += A(tile mt0+i) * B(tile nt) for i < MT, every tile valid.  The B operand comes from global memory / L2
// (latency of several hundred cycles), so its fragments are requested well ahead of the MFMAs that consume them:
//   KBS > 0 (k extent known at compile time): all KBS fragments are in flight before the first MFMA;
//   KBS == 0: a two-deep rotating prefetch.  A fragments (LDS) are fetched one k-block ahead.
template <int MT>
__device__ __forceinline__ void rat_mfma_block(f32x4 (&acc)[MT], const float4 (&a)[MT], const float4& b) {
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA16(a[i].x, b.x, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA16(a[i].y, b.y, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA16(a[i].z, b.z, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA16(a[i].w, b.w, acc[i]);
}

template <int MT, int KBS, class AF, class BF>
__device__ __forceinline__ void rat_wave_gemm_col(f32x4 (&acc)[MT], const AF& af, const BF& bf, int mt0, int nt,
                                                  int kblocks, int kb_begin = 0) {
    if (KBS > 0) {
        float4 b[KBS > 0 ? KBS : 1];
#pragma unroll
        for (int kb = 0; kb < KBS; ++kb) b[kb] = bf(nt, kb);
#pragma unroll
        for (int kb = 0; kb < KBS; ++kb) {
            float4 a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = af(mt0 + i, kb);
            rat_mfma_block<MT>(acc, a, b[kb]);
        }
    } else {
        const int last = kblocks - 1;                              // k-blocks [kb_begin, kblocks)
        float4 b0 = bf(nt, kb_begin);
        float4 b1 = bf(nt, last < kb_begin + 1 ? last : kb_begin + 1);
        float4 an[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) an[i] = af(mt0 + i, kb_begin);
        for (int kb = kb_begin; kb < kblocks; ++kb) {
            const float4 b2 = bf(nt, kb + 2 < last ? kb + 2 : last);
            float4 a[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = an[i];
            const int kn = kb + 1 < last ? kb + 1 : last;
#pragma unroll
            for (int i = 0; i < MT; ++i) an[i] = af(mt0 + i, kn);
            rat_mfma_block<MT>(acc, a, b0);
            b0 = b1;
            b1 = b2;
        }
    }
}

// acc[i][j] += A(tile mt0+i) * B(tile nt0+j) over kblocks 16-wide k-blocks; tiles beyond mt_valid/nb_valid
// are skipped (wave-uniform).  All 64 lanes of the wave must call this together.
template <int MT, int NB, class AF, class BF>
__device__ __forceinline__ void rat_wave_gemm(f32x4 (&acc)[MT][NB], const AF& af, const BF& bf, int mt0, int nt0,
                                              int mt_valid, int nb_valid, int kblocks) {
    for (int kb = 0; kb < kblocks; ++kb) {
        float4 a[MT], b[NB];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = (i < mt_valid) ? af(mt0 + i, kb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < NB; ++j) b[j] = (j < nb_valid) ? bf(nt0 + j, kb) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (i < mt_valid && j < nb_valid) acc[i][j] = RAT_MFMA16(a[i].x, b[j].x, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (i < mt_valid && j < nb_valid) acc[i][j] = RAT_MFMA16(a[i].y, b[j].y, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (i < mt_valid && j < nb_valid) acc[i][j] = RAT_MFMA16(a[i].z, b[j].z, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (i < mt_valid && j < nb_valid) acc[i][j] = RAT_MFMA16(a[i].w, b[j].w, acc[i][j]);
    }
}

// C[MTILES*16 rows][n_tiles*16] = A (LDS rows) x B (weights in L2) as 16x16 tiles; tasks = (M block of MT tiles) x (N tile),
// dealt round-robin to the NWAVES waves; epi(mt, nt, acc) consumes each finished tile.
// FAST: all M tiles are always computed (padding rows are zero), no validity predicates, B one k-block ahead.
template <bool FAST, int MT, int NWAVES, int MTILES, int KBS, class AF, class BF, class Epi>
__device__ __forceinline__ void rat_gemm_phase(const AF& A, const BF& Bw, int mt_valid, int n_tiles, int kblocks, const Epi& epi) {
    constexpr int MBLOCKS = MTILES / MT;
    const int ntasks = MBLOCKS * n_tiles;
    for (int task = rat_wave(); task < ntasks; task += NWAVES) {
        const int mt0 = (task / n_tiles) * MT, nt = task % n_tiles;
        if (!FAST && mt0 >= mt_valid) continue;
        if (FAST) {
            f32x4 acc[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = rat_zero4();
            rat_wave_gemm_col<MT, KBS>(acc, A, Bw, mt0, nt, kblocks);
#pragma unroll
            for (int i = 0; i < MT; ++i) epi(mt0 + i, nt, acc[i]);
        } else {
            f32x4 acc[MT][1];
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][0] = rat_zero4();
            const int mtv = mt_valid - mt0 < MT ? mt_valid - mt0 : MT;
            rat_wave_gemm<MT, 1>(acc, A, Bw, mt0, nt, mtv, 1, kblocks);
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (i < mtv) epi(mt0 + i, nt, acc[i][0]);
        }
    }
}

// single-tile variant for the persistent weight-gradient accumulators
template <class AF, class BF>
__device__ __forceinline__ f32x4 rat_wave_gemm1(f32x4 acc, const AF& af, const BF& bf, int mt, int nt, int kblocks) {
    for (int kb = 0; kb < kblocks; ++kb) {
        const float4 a = af(mt, kb);
        const float4 b = bf(nt, kb);
        acc = RAT_MFMA16(a.x, b.x, acc);
        acc = RAT_MFMA16(a.y, b.y, acc);
        acc = RAT_MFMA16(a.z, b.z, acc);
        acc = RAT_MFMA16(a.w, b.w, acc);
    }
    return acc;
}

// Persistent weight-gradient tiles of one wave: tile ids wave + NWAVES*s (s < SLOTS) of an (m_tiles x ntn) grid,
// acc[s] += A^T-tile(mt_s) * B-tile(nt_s) over `kblocks` token blocks.  When ntn divides NWAVES every slot of a wave
// shares the same nt, so the B fragment is fetched once per k-block and the (independent) slots' MFMAs issue back to back.
// KBC > 0: the k extent is a compile-time constant (fast paths: all 64 rows, padding rows are exact zeros) — the k loop
// unrolls and the operand fetches of later k-blocks are scheduled over the MFMAs of earlier ones.
template <int SLOTS, int NWAVES, int KBC = 0, class AF, class BF>
__device__ __forceinline__ void rat_wave_gemm_slots(f32x4 (&acc)[SLOTS], const AF& af, const BF& bf, int ntiles, int ntn,
                                                    int kblocks_rt) {
    const int kblocks = KBC > 0 ? KBC : kblocks_rt;
    const int w = rat_wave();
    if (NWAVES % ntn == 0) {
        const int nt = w % ntn;
        constexpr int HALF = SLOTS > 4 ? 4 : SLOTS;
#pragma unroll
        for (int s0 = 0; s0 < SLOTS; s0 += HALF) {
            if (w + NWAVES * s0 >= ntiles) break;
            // live slots of this group (wave-uniform): slots past the last tile issue nothing
            int live = (ntiles - w - NWAVES * s0 + NWAVES - 1) / NWAVES;
            if (live > HALF) live = HALF;
#pragma unroll
            for (int kb = 0; kb < kblocks; ++kb) {
                const float4 b = bf(nt, kb);
                float4 a[HALF];
#pragma unroll
                for (int s = 0; s < HALF; ++s)
                    if (s < live) a[s] = af((w + NWAVES * (s0 + s)) / ntn, kb);
#pragma unroll
                for (int s = 0; s < HALF; ++s)
                    if (s < live) acc[s0 + s] = RAT_MFMA16(a[s].x, b.x, acc[s0 + s]);
#pragma unroll
                for (int s = 0; s < HALF; ++s)
                    if (s < live) acc[s0 + s] = RAT_MFMA16(a[s].y, b.y, acc[s0 + s]);
#pragma unroll
                for (int s = 0; s < HALF; ++s)
                    if (s < live) acc[s0 + s] = RAT_MFMA16(a[s].z, b.z, acc[s0 + s]);
#pragma unroll
                for (int s = 0; s < HALF; ++s)
                    if (s < live) acc[s0 + s] = RAT_MFMA16(a[s].w, b.w, acc[s0 + s]);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int id = w + NWAVES * s;
            if (id < ntiles) acc[s] = rat_wave_gemm1(acc[s], af, bf, id / ntn, id % ntn, kblocks);
        }
    }
}

// Compile-time geometry variant: NT tiles (NTN per tile row) spread over NWAVES waves, KB k-blocks.  Slots that every wave
// owns (NT / NWAVES of them) run branch-free with their MFMAs interleaved; a ragged last slot takes ONE wave-uniform branch.
// KB > 0: k extent known, loop unrolled; KB == 0: `kblocks` at run time, loop kept rolled.
template <int SL, int NWAVES, int NT, int NTN, int KB, class AF, class BF>
__device__ __forceinline__ void rat_wave_gemm_ct(f32x4 (&acc)[SL], const AF& af, const BF& bf, int kblocks = KB) {
    static_assert(NWAVES % NTN == 0, "every wave keeps one B tile column");
    constexpr int FULL = NT / NWAVES;
    static_assert(FULL <= SL && (NT + NWAVES - 1) / NWAVES <= SL, "slots");
    const int w = rat_wave();
    const int nt = w % NTN;
    if (FULL > 0) {
#pragma unroll(KB > 0 ? KB : 1)
        for (int kb = 0; kb < (KB > 0 ? KB : kblocks); ++kb) {
            const float4 b = bf(nt, kb);
            float4 a[FULL > 0 ? FULL : 1];
#pragma unroll
            for (int s = 0; s < FULL; ++s) a[s] = af((w + NWAVES * s) / NTN, kb);
#pragma unroll
            for (int s = 0; s < FULL; ++s) acc[s] = RAT_MFMA16(a[s].x, b.x, acc[s]);
#pragma unroll
            for (int s = 0; s < FULL; ++s) acc[s] = RAT_MFMA16(a[s].y, b.y, acc[s]);
#pragma unroll
            for (int s = 0; s < FULL; ++s) acc[s] = RAT_MFMA16(a[s].z, b.z, acc[s]);
#pragma unroll
            for (int s = 0; s < FULL; ++s) acc[s] = RAT_MFMA16(a[s].w, b.w, acc[s]);
        }
    }
    if (FULL * NWAVES < NT) {
        if (w + NWAVES * FULL < NT) {
#pragma unroll(KB > 0 ? KB : 1)
            for (int kb = 0; kb < (KB > 0 ? KB : kblocks); ++kb) {
                const float4 b = bf(nt, kb);
                const float4 a = af((w + NWAVES * FULL) / NTN, kb);
                acc[FULL] = RAT_MFMA16(a.x, b.x, acc[FULL]);
                acc[FULL] = RAT_MFMA16(a.y, b.y, acc[FULL]);
                acc[FULL] = RAT_MFMA16(a.z, b.z, acc[FULL]);
                acc[FULL] = RAT_MFMA16(a.w, b.w, acc[FULL]);
            }
        }
    }
}

// ============================================================================================================================
// bf16x3: fp32 GEMMs on the bf16 matrix instruction (v_mfma_f32_16x16x32_bf16, 16x the fp32 MFMA rate per FLOP).
//
//   x = h + m + l EXACTLY, with h, m, l bf16 numbers (h = x rounded to nearest bf16, m = x - h rounded to nearest, l = x - h - m;
//   every subtraction is exact, rat_split2).  A product a*b is evaluated
//   as the six cross terms of weight >= 2^-16:  al*bh + ah*bl + am*bm + am*bh + ah*bm + ah*bh  (each bf16 x bf16 product is exact
//   in fp32; the MFMA accumulates in fp32).  Dropped: am*bl + al*bm + al*bl <= 2^-25 |a b| — a quarter of ONE fp32 rounding of the
//   product, so the result has fp32-class accuracy (measured: tools/probes/bf16x3_probe.hip) while costing 6 x 16 cycles per
//   K = 32 step of a 16x16 tile instead of 8 x 32 cycles on v_mfma_f32_16x16x4_f32.
// Lane maps (cdna_hip_programming.md §3): A: lane l holds A[row l & 15][k = 8 (l >> 4) + j], j = 0..7 (one 16-byte fragment);
// B: lane l holds B[k = 8 (l >> 4) + j][col l & 15]; C/D as the fp32 form.  As everywhere in this file the k order inside a
// step may be permuted as long as A and B agree.
#ifdef RAT_EMU
#define RAT_MFMA_BF16(a, b, c) emu_mfma_f32_16x16x32_bf16((a), (b), (c))
#define RAT_LDS_TR16(p) emu_lds_tr16((const unsigned short*)(p))
struct rat_u4 { unsigned x, y, z, w; };
#else
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define RAT_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// ds_read_b64_tr_b16: lane 4q + p of every 16-lane group supplies the address of (row q, columns 4p..4p+3) of a 4 x 16 block of
// 16-bit elements; lane i of the group receives column i, row q in element q (cdna_hip_programming.md T10; probe (e))
#define RAT_LDS_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p))
typedef uint4 rat_u4;
#endif

struct RatB3 {                       // one operand fragment in its three planes
    bf16x8 h, m, l;
};

__device__ __forceinline__ unsigned rat_fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float rat_bitsf(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ bf16x8 rat_as_bf16x8(const rat_u4& v) { return __builtin_bit_cast(bf16x8, v); }

// two floats -> packed bf16 (low half = x0, high half = x1), ROUND TO NEAREST EVEN: v_cvt_pk_bf16_f32 on the device (one instruction
// per pair), the integer formula on the host emulation (bit-identical for finite values)
__device__ __forceinline__ unsigned rat_bf16_pair(float x0, float x1) {
#ifdef RAT_EMU
    const unsigned u0 = rat_fbits(x0), u1 = rat_fbits(x1);
    return ((u0 + 0x7fffu + ((u0 >> 16) & 1u)) >> 16) | ((u1 + 0x7fffu + ((u1 >> 16) & 1u)) & 0xffff0000u);
#else
    typedef __bf16 rat_bf16x2 __attribute__((ext_vector_type(2)));
    const rat_bf16x2 v = {(__bf16)x0, (__bf16)x1};
    return __builtin_bit_cast(unsigned, v);
#endif
}
// two floats -> their packed (low half = x0, high half = x1) bf16 chunks.  Round-to-nearest split (round 4; rounds 2-3 truncated):
// h = rne(x), m = rne(x - h), l = x - h - m.  Every subtraction is exact and l needs at most 8 significant bits, so x = h + m + l
// still holds EXACTLY; against the truncation split the remainders are half as large (|m| <= 2^-9 |x|, |l| <= 2^-17 |x|: the dropped
// cross terms of rat_mfma3 shrink from 2^-23 to 2^-25 |a b|) and, unlike truncated remainders, carry no sign bias — a truncated m / l
// always has the sign of x, so the dropped terms always had the sign of a b.  11 VALU instructions per pair instead of 14.
__device__ __forceinline__ void rat_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = rat_bf16_pair(x0, x1);
    const float r0 = x0 - rat_bitsf(h << 16), r1 = x1 - rat_bitsf(h & 0xffff0000u);
    m = rat_bf16_pair(r0, r1);
    const float s0 = r0 - rat_bitsf(m << 16), s1 = r1 - rat_bitsf(m & 0xffff0000u);
    l = rat_bf16_pair(s0, s1);                           // exact: s0, s1 have no more than 8 significant bits
}
// eight floats (two float4: elements 0..3, 4..7) -> three 16-byte pieces
__device__ __forceinline__ void rat_split8(const float4& a, const float4& b, rat_u4& h, rat_u4& m, rat_u4& l) {
    rat_split2(a.x, a.y, h.x, m.x, l.x);
    rat_split2(a.z, a.w, h.y, m.y, l.y);
    rat_split2(b.x, b.y, h.z, m.z, l.z);
    rat_split2(b.z, b.w, h.w, m.w, l.w);
}
__device__ __forceinline__ RatB3 rat_split8_frag(const float4& a, const float4& b) {
    rat_u4 h, m, l;
    rat_split8(a, b, h, m, l);
    return RatB3{rat_as_bf16x8(h), rat_as_bf16x8(m), rat_as_bf16x8(l)};
}
// value of element e (0..7) of a 16-byte piece triple: h + m + l is exact, in this order
__device__ __forceinline__ float rat_join(unsigned h, unsigned m, unsigned l, int hi) {
    const unsigned sh = hi ? 0u : 16u, mk = hi ? 0xffff0000u : 0xffffffffu;
    return (rat_bitsf((h << sh) & mk) + rat_bitsf((m << sh) & mk)) + rat_bitsf((l << sh) & mk);
}

// c += a * b, six cross products, small terms first
__device__ __forceinline__ f32x4 rat_mfma3(const RatB3& a, const RatB3& b, f32x4 c) {
    c = RAT_MFMA_BF16(a.l, b.h, c);
    c = RAT_MFMA_BF16(a.h, b.l, c);
    c = RAT_MFMA_BF16(a.m, b.m, c);
    c = RAT_MFMA_BF16(a.m, b.h, c);
    c = RAT_MFMA_BF16(a.h, b.m, c);
    c = RAT_MFMA_BF16(a.h, b.h, c);
    return c;
}
// MT row tiles against one B fragment: the MT independent accumulator chains interleave product by product
template <int MT>
__device__ __forceinline__ void rat_mfma3_block(f32x4 (&acc)[MT], const RatB3 (&a)[MT], const RatB3& b) {
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA_BF16(a[i].l, b.h, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA_BF16(a[i].h, b.l, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA_BF16(a[i].m, b.m, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA_BF16(a[i].m, b.h, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA_BF16(a[i].h, b.m, acc[i]);
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = RAT_MFMA_BF16(a[i].h, b.h, acc[i]);
}

// ---- an activation tile [64 rows][W columns] held in LDS as three bf16 planes.  Unit = a 16-byte PIECE (8 consecutive columns
// of one row); rows are RS bytes apart; SWZ (0, 7 or 15): the piece index is XORed with (row & SWZ), which makes the 16-byte row
// reads of an MFMA fragment AND the transposed 4-row block reads below bank-conflict free for 128-byte rows (SWZ 7, W = 64) and
// the row reads for 256-byte rows (SWZ 15, W = 128).
// SHIFT: the XOR acts on piece-index bits SHIFT.. (SHIFT 1 keeps the two pieces of a 32-byte column pair adjacent: the transposed
// reads of 256-byte rows then spread 8 consecutive rows over all 64 banks).
template <int RS, int SWZ, int PLANE_BYTES, int SHIFT = 0>
struct RatPlanes {
    char* base;
    __device__ __forceinline__ int off(int r, int o) const { return r * RS + 16 * (o ^ ((r & SWZ) << SHIFT)); }
    // half a piece: the 4 columns [4 q4, 4 q4 + 4) of row r (8 bytes per plane) — what an accumulator register quad holds
    __device__ __forceinline__ void store_half(int r, int q4, unsigned h0, unsigned h1, unsigned m0, unsigned m1, unsigned l0,
                                               unsigned l1) const {
        const int a = off(r, q4 >> 1) + 8 * (q4 & 1);
        *reinterpret_cast<uint2*>(base + a) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(base + PLANE_BYTES + a) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(base + 2 * PLANE_BYTES + a) = make_uint2(l0, l1);
    }
    // ... and the same 8 bytes per plane read back (by the thread that wrote them: no barrier needed, LDS operations of a wave
    // stay in order)
    __device__ __forceinline__ void load_half(int r, int q4, unsigned& h0, unsigned& h1, unsigned& m0, unsigned& m1, unsigned& l0,
                                              unsigned& l1) const {
        const int a = off(r, q4 >> 1) + 8 * (q4 & 1);
        const uint2 h = *reinterpret_cast<const uint2*>(base + a);
        const uint2 m = *reinterpret_cast<const uint2*>(base + PLANE_BYTES + a);
        const uint2 l = *reinterpret_cast<const uint2*>(base + 2 * PLANE_BYTES + a);
        h0 = h.x; h1 = h.y; m0 = m.x; m1 = m.y; l0 = l.x; l1 = l.y;
    }
    __device__ __forceinline__ void store(int r, int o, const rat_u4& h, const rat_u4& m, const rat_u4& l) const {
        const int a = off(r, o);
        *reinterpret_cast<rat_u4*>(base + a) = h;
        *reinterpret_cast<rat_u4*>(base + PLANE_BYTES + a) = m;
        *reinterpret_cast<rat_u4*>(base + 2 * PLANE_BYTES + a) = l;
    }
    // the tile as a ROW operand (A[row][k] of C = A B): fragment of row tile `mt`, K-step `s` (columns 32 s .. 32 s + 31)
    __device__ __forceinline__ RatB3 row_frag(int mt, int s) const {
        const int l = rat_lane(), a = off(16 * mt + (l & 15), 4 * s + (l >> 4));
        return RatB3{rat_as_bf16x8(*reinterpret_cast<const rat_u4*>(base + a)),
                     rat_as_bf16x8(*reinterpret_cast<const rat_u4*>(base + PLANE_BYTES + a)),
                     rat_as_bf16x8(*reinterpret_cast<const rat_u4*>(base + 2 * PLANE_BYTES + a))};
    }
    // the tile as a COLUMN operand (contraction over the 64 rows): fragment for column tile `ct` (columns 16 ct .. +15), K-step s
    // = rows 32 s .. 32 s + 31.  k slot j of lane group g <-> row 32 s + 4 g + j (j < 4), 32 s + 16 + 4 g + (j - 4) (j >= 4):
    // two transposed 4 x 16 block reads per plane.  Every col_frag of every tile uses this same row order.
    __device__ __forceinline__ bf16x8 col_plane(const char* pl, int ct, int s) const {
        const int l = rat_lane(), g = l >> 4, q = (l >> 2) & 3, p = l & 3;
        const int o = 2 * ct + (p >> 1), hb = 8 * (p & 1);
        const s16x4 lo = RAT_LDS_TR16(pl + off(32 * s + 4 * g + q, o) + hb);
        const s16x4 hi = RAT_LDS_TR16(pl + off(32 * s + 16 + 4 * g + q, o) + hb);
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[e] = lo[e];
            r[4 + e] = hi[e];
        }
        return r;
    }
    __device__ __forceinline__ RatB3 col_frag(int ct, int s) const {
        return RatB3{col_plane(base, ct, s), col_plane(base + PLANE_BYTES, ct, s), col_plane(base + 2 * PLANE_BYTES, ct, s)};
    }
};
// the row order of col_frag, for operands that are fetched some other way (fp32 column reads + split)
__device__ __forceinline__ int rat_col_slot_row(int s, int g, int j) { return 32 * s + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)); }

// ---- weights as B operands: pre-split by rat_launch_split_weights into FRAGMENT-MAJOR planes in global memory —
// [n tile][K step][plane][lane] x 16 bytes — so that a wave's fragment is one coalesced 1 KB load per plane (L2-resident)
struct RatWPlanes {
    const rat_u4* base;
    int steps;
    __device__ __forceinline__ RatB3 operator()(int nt, int s) const {
        const rat_u4* p = base + ((size_t)(nt * steps + s) * 3) * 64 + rat_lane();
        return RatB3{rat_as_bf16x8(p[0]), rat_as_bf16x8(p[64]), rat_as_bf16x8(p[128])};
    }
};
// B[k][n] = transpose ? w[k * ld + n] : w[n * ld + k] for n < N, k < K (zero beyond); out: rat_wplanes_bytes(N, K) bytes
inline size_t rat_wplanes_bytes(int N, int K) { return (size_t)((N + 15) / 16) * ((K + 31) / 32) * 3 * 64 * 16; }
int rat_launch_split_weights(const float* w, int N, int K, int ld, int transpose, void* out, void* stream, int perm = 0, int valid = 0);

// row/col of accumulator register r of a 16x16 tile
__device__ __forceinline__ int rat_acc_row(int tile_m, int r) { return tile_m * 16 + (rat_lane() >> 4) * 4 + r; }
__device__ __forceinline__ int rat_acc_col(int tile_n) { return tile_n * 16 + (rat_lane() & 15); }

// sum the persistent-gradient slabs of all work-groups: out[p] = sum_wg slab[wg][p], fixed order
int rat_launch_transpose(const float* src, float* dst, int R, int C, void* stream);     // dst[C][R] = src[R][C]^T
int rat_launch_reduce_slabs(const float* slabs, int nslabs, int64_t stride, float* const* outs_host,
                            const int64_t* offsets, const int64_t* sizes, int nouts, void* stream);
