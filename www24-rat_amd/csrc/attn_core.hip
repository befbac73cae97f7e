// attn_core.hip — K2d: the softmax(Q K^T * scale) V core for LONG sequences, forward and backward, on projected Q|K|V rows.
//
// RAT_m0 (RAT_m0.py:123-127) attends jointly over all T*S tokens of a sample (231 at the north-star shape): a whole
// sequence no longer fits the 64-row LDS tile of the fused kernel in attn.hip, so that variant runs the projections as
// plain MFMA GEMMs (rat_sgemm) and LayerNorm as K2c, and only this core is new.  One work-group per (sequence, head) —
// several pairs per work-group when the sequences are short:
// the head's K and V rows (backward: also Q, dO, lse, delta) are staged in LDS once, one lane owns one query row (pass 2
// of backward: one key row) and walks all keys with the online softmax of attn.hip — same log2-domain arithmetic, same
// saved log-sum-exp convention (lse = m + log2(l) of the scores scaled by scale * log2(e)).
//
// qkv is [ntok][3*I] (I = heads*dh; Q | K | V, head-major inside each third, exactly nn.Linear(d, 3I)'s output), o and
// do are [ntok][I], lse is [ntok][heads]; sequence q owns tokens [q*L, (q+1)*L).
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int CORE_DH_MAX = 32;

struct CoreArgs {
    const float* qkv;
    const float* o;
    const float* dout;
    const float* lse_in;
    float* o_out;
    float* lse_out;
    float* dqkv;
    int64_t nseq, q_div, hi_stride, lo_stride, pos_stride;      // RatSeqMap: token of (sequence q, position p)
    int L, heads, dh;
    float scale;
};

// same addressing as attn.hip: sequences may be strided through the token grid (the cross-sample phase of RAT_m2)
__device__ __forceinline__ int64_t core_token(const CoreArgs& a, int64_t q, int p) {
    return (q / a.q_div) * a.hi_stride + (q % a.q_div) * a.lo_stride + (int64_t)p * a.pos_stride;
}

template <int DH>
struct Vec {
    static constexpr int N = DH > 0 ? DH : CORE_DH_MAX;
    float v[N];
    __device__ __forceinline__ void load(const float* p, int dh) {
        if (DH > 0 && DH % 2 == 0) {                    // rows start on 8-byte boundaries for an even dim_head: 8-byte accesses
#pragma unroll
            for (int k = 0; k < N; k += 2) {
                const float2 t = *reinterpret_cast<const float2*>(p + k);
                v[k] = t.x;
                v[k + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = (DH > 0 || k < dh) ? p[k] : 0.f;
        }
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = 0.f;
    }
    __device__ __forceinline__ float dot(const Vec& o) const {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) s = fmaf(v[k], o.v[k], s);
        return s;
    }
    __device__ __forceinline__ void axpy(float a, const Vec& x) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = fmaf(a, x.v[k], v[k]);
    }
    __device__ __forceinline__ void scale_axpy(float c, float a, const Vec& x) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = fmaf(a, x.v[k], v[k] * c);
    }
    __device__ __forceinline__ void store(float* p, int dh, float mul) const {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if (DH > 0 || k < dh) p[k] = v[k] * mul;
    }
};

// LDS rows are unpadded [L][dh]: in the key / query walks every lane of a (sequence, head) pair reads the SAME row (a
// broadcast), and the staging writes are contiguous.  A work-group serves G = max(1, blockDim / L) (sequence, head) pairs per
// pass — pair = task / L, row = task % L — so that short sequences (the 9- and 31-token phases of a wide-head RAT_m2) fill
// the lanes instead of leaving most of each wave idle; consecutive pairs are consecutive heads of one sequence, i.e.
// neighbouring slices of the same Q|K|V rows.
template <int DH, bool MULTI>
__global__ void __launch_bounds__(1024) core_fwd_kernel(CoreArgs a) {
    RAT_DYN_SMEM(smem);
    const int dh = DH > 0 ? DH : a.dh, L = a.L, I = a.heads * dh;
    const int G = MULTI ? (int)blockDim.x / L : 1;        // MULTI = false: one pair per work-group, the pair arithmetic folds away
    float* ks = reinterpret_cast<float*>(smem);          // [G][L][dh]
    float* vs = ks + (size_t)G * L * dh;                 // [G][L][dh]
    const float sl2 = a.scale * RAT_LOG2E;
    const int64_t ntask = a.nseq * a.heads, ngroups = (ntask + G - 1) / G;
    for (int64_t group = blockIdx.x; group < ngroups; group += gridDim.x) {
        for (int e = threadIdx.x; e < G * L * dh; e += blockDim.x) {
            const int p = MULTI ? e / (L * dh) : 0, rem = e - p * (L * dh), j = rem / dh, c = rem - j * dh;
            const int64_t task = group * G + p;
            if (task < ntask) {
                const int64_t sq = task / a.heads;
                const int h = (int)(task - sq * a.heads);
                const float* row = a.qkv + core_token(a, sq, j) * (3 * I) + h * dh;
                ks[e] = row[I + c];
                vs[e] = row[2 * I + c];
            }
        }
        __syncthreads();
        for (int w = threadIdx.x; w < G * L; w += blockDim.x) {
            const int p = MULTI ? w / L : 0, i = w - p * L;
            const int64_t task = group * G + p;
            if (task >= ntask) continue;
            const int64_t sq = task / a.heads;
            const int h = (int)(task - sq * a.heads);
            const float* kp = ks + (size_t)p * L * dh;
            const float* vp = vs + (size_t)p * L * dh;
            Vec<DH> q, o, kv;
            const int64_t tok = core_token(a, sq, i);
            q.load(a.qkv + tok * (3 * I) + h * dh, dh);
            o.zero();
            float m = -3.0e38f, l = 0.f;
            for (int j = 0; j < L; ++j) {
                kv.load(kp + (size_t)j * dh, dh);
                const float s = q.dot(kv) * sl2;
                const float mn = fmaxf(m, s);
                const float corr = rat_exp2(m - mn), pr = rat_exp2(s - mn);
                l = l * corr + pr;
                kv.load(vp + (size_t)j * dh, dh);
                o.scale_axpy(corr, pr, kv);
                m = mn;
            }
            o.store(a.o_out + tok * I + h * dh, dh, 1.0f / l);
            if (a.lse_out != nullptr) a.lse_out[tok * a.heads + h] = m + rat_log2(l);
        }
        __syncthreads();
    }
}

template <int DH, bool MULTI>
__global__ void __launch_bounds__(1024) core_bwd_kernel(CoreArgs a) {
    RAT_DYN_SMEM(smem);
    const int dh = DH > 0 ? DH : a.dh, L = a.L, I = a.heads * dh;
    const int G = MULTI ? (int)blockDim.x / L : 1;        // MULTI = false: one pair per work-group, the pair arithmetic folds away
    const size_t plane = (size_t)G * L * dh;
    float* qs = reinterpret_cast<float*>(smem);          // [G][L][dh]
    float* ks = qs + plane;
    float* vs = ks + plane;
    float* gs = vs + plane;                              // dO
    float* ls = gs + plane;                              // [G][L] lse
    float* ds = ls + (size_t)G * L;                      // [G][L] delta = dO . O
    const float sl2 = a.scale * RAT_LOG2E;
    const int64_t ntask = a.nseq * a.heads, ngroups = (ntask + G - 1) / G;
    for (int64_t group = blockIdx.x; group < ngroups; group += gridDim.x) {
        for (int e = threadIdx.x; e < G * L * dh; e += blockDim.x) {
            const int p = MULTI ? e / (L * dh) : 0, rem = e - p * (L * dh), j = rem / dh, c = rem - j * dh;
            const int64_t task = group * G + p;
            if (task < ntask) {
                const int64_t sq = task / a.heads;
                const int h = (int)(task - sq * a.heads);
                const int64_t tok = core_token(a, sq, j);
                const float* row = a.qkv + tok * (3 * I) + h * dh;
                qs[e] = row[c];
                ks[e] = row[I + c];
                vs[e] = row[2 * I + c];
                gs[e] = a.dout[tok * I + h * dh + c];
            }
        }
        for (int w = threadIdx.x; w < G * L; w += blockDim.x) {
            const int p = MULTI ? w / L : 0, i = w - p * L;
            const int64_t task = group * G + p;
            if (task >= ntask) continue;
            const int64_t sq = task / a.heads;
            const int h = (int)(task - sq * a.heads);
            const int64_t tok = core_token(a, sq, i);
            ls[w] = a.lse_in[tok * a.heads + h];
            float dsum = 0.f;
            for (int c = 0; c < dh; ++c) dsum = fmaf(a.dout[tok * I + h * dh + c], a.o[tok * I + h * dh + c], dsum);
            ds[w] = dsum;
        }
        __syncthreads();
        // pass 1: one lane per query row -> dQ
        for (int w = threadIdx.x; w < G * L; w += blockDim.x) {
            const int p = MULTI ? w / L : 0, i = w - p * L;
            const int64_t task = group * G + p;
            if (task >= ntask) continue;
            const int64_t sq = task / a.heads;
            const int h = (int)(task - sq * a.heads);
            const size_t po = (size_t)p * L * dh;
            Vec<DH> q, go, dq, kv;
            q.load(qs + po + (size_t)i * dh, dh);
            go.load(gs + po + (size_t)i * dh, dh);
            dq.zero();
            const float lse = ls[w], delta = ds[w];
            for (int j = 0; j < L; ++j) {
                kv.load(vs + po + (size_t)j * dh, dh);
                const float dp = go.dot(kv);
                kv.load(ks + po + (size_t)j * dh, dh);
                const float pr = rat_exp2(q.dot(kv) * sl2 - lse);
                dq.axpy(pr * (dp - delta), kv);
            }
            dq.store(a.dqkv + core_token(a, sq, i) * (3 * I) + h * dh, dh, a.scale);
        }
        // pass 2: one lane per key row -> dK, dV
        for (int w = threadIdx.x; w < G * L; w += blockDim.x) {
            const int p = MULTI ? w / L : 0, j = w - p * L;
            const int64_t task = group * G + p;
            if (task >= ntask) continue;
            const int64_t sq = task / a.heads;
            const int h = (int)(task - sq * a.heads);
            const size_t po = (size_t)p * L * dh;
            Vec<DH> kk, vv, dk, dv, t, qv;
            kk.load(ks + po + (size_t)j * dh, dh);
            vv.load(vs + po + (size_t)j * dh, dh);
            dk.zero();
            dv.zero();
            for (int i = 0; i < L; ++i) {
                t.load(gs + po + (size_t)i * dh, dh);
                const float dp = t.dot(vv);
                qv.load(qs + po + (size_t)i * dh, dh);
                const float pr = rat_exp2(qv.dot(kk) * sl2 - ls[p * L + i]);
                dv.axpy(pr, t);
                dk.axpy(pr * (dp - ds[p * L + i]), qv);
            }
            float* drow = a.dqkv + core_token(a, sq, j) * (3 * I) + h * dh;
            dk.store(drow + I, dh, a.scale);
            dv.store(drow + 2 * I, dh, 1.0f);
        }
        __syncthreads();
    }
}

int core_threads(int L) {                                  // short sequences share a 256-thread work-group (G pairs per pass)
    if (L <= 256) return 256;
    int t = (L + 63) / 64 * 64;
    return t > 1024 ? 1024 : t;
}
int core_pairs(int L) { const int t = core_threads(L); return t >= L ? t / L : 1; }

int core_check(int64_t nseq, int L, int heads, int dh, size_t smem) {
    RAT_REQUIRE(nseq > 0 && L > 0 && heads > 0 && dh > 0, "bad dims");
    RAT_REQUIRE(dh <= CORE_DH_MAX, "dim_head > 32 is not supported by the long-sequence attention core");
    RAT_REQUIRE(smem <= 160 * 1024, "sequence too long for the LDS staging of one head's K, V (and Q, dO) rows");
    return 0;
}

// ---- the forward core on the matrix pipe, exact fp32, for LONG sequences at dim_head 10 (round 5) ----------------------------------
// The construction of attn.hip's b3_fwd_core_mfma as a flash loop: a 256-thread work-group owns one (sequence, head) pair, its K and V
// rows staged in LDS ([L16][12] each); a wave owns 16-query tiles and walks the keys 32 at a time:
//   S^T = K Q^T on v_mfma_f32_16x16x4_f32 (three k steps for dim_head 10): lane (g, m) holds S^T[key 16 jt + 4 g + r][query m], so a
//   query's scores of the key block sit in the four lanes of its column — block maximum in registers + two cross-row swaps, then
//   p = exp2(s - m_new), the running sum and the O^T accumulator rescaled by exp2(m_old - m_new) (the online softmax of the VALU
//   kernel, per key block instead of per key), and O^T += V^T P^T with the SAME registers as the B operand (k-step (jt, r) contracts
//   over the keys {16 jt + 4 g + r}).  Seven MFMAs per (query tile, key tile); nothing passes through LDS after the staging.
// RAT_m0's joint sequences (231 tokens at the north-star shape): 15 x 15 tiles, 96 % full.
#ifndef RAT_CM_KB
#define RAT_CM_KB 3
#endif
#ifndef RAT_CM_THREADS
#define RAT_CM_THREADS 512      // round 6, same-box A/B at L = 231 (profiles/round6/r6_attn_core_fwd_ab.txt): 8 waves x 3 key tiles per trip 1.58-1.61 ms,
#endif                          // 4 waves x 4 tiles (round 5) 1.78-1.79 ms — like the backward, the kernel lives on waves in flight
constexpr int CM_THREADS = RAT_CM_THREADS, CM_WAVES = CM_THREADS / 64, CM_LD = 12, CM_DH = 10, CM_KB = RAT_CM_KB;      // CM_KB: 16-key tiles per trip of the key loop
__device__ __forceinline__ float cm_rows_max(float v) {
#ifdef RAT_EMU
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
#else
    auto r = __builtin_amdgcn_permlane16_swap(rat_fbits(v), rat_fbits(v), false, false);
    v = fmaxf(rat_bitsf(r[0]), rat_bitsf(r[1]));
    r = __builtin_amdgcn_permlane32_swap(rat_fbits(v), rat_fbits(v), false, false);
    return fmaxf(rat_bitsf(r[0]), rat_bitsf(r[1]));
#endif
}
__device__ __forceinline__ float cm_rows_sum(float v) {
#ifdef RAT_EMU
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
#else
    auto r = __builtin_amdgcn_permlane16_swap(rat_fbits(v), rat_fbits(v), false, false);
    v = rat_bitsf(r[0]) + rat_bitsf(r[1]);
    r = __builtin_amdgcn_permlane32_swap(rat_fbits(v), rat_fbits(v), false, false);
    return rat_bitsf(r[0]) + rat_bitsf(r[1]);
#endif
}
__global__ void __launch_bounds__(CM_THREADS) core_fwd_mfma_kernel(CoreArgs a) {
    RAT_DYN_SMEM(smem);
    const int L = a.L, I = a.heads * CM_DH, L16 = (L + 15) / 16 * 16, NT = L16 / 16;
    constexpr int SPARE = 16 * (CM_KB - 1);              // zero rows past the sequence: the last trip's tiles beyond it
    float* ks = reinterpret_cast<float*>(smem);          // [L16 + SPARE][12]: rows >= L and columns 10, 11 are zero
    float* vs = ks + (size_t)(L16 + SPARE) * CM_LD;
    const int w = rat_wave(), l = rat_lane(), g = l >> 4, m = l & 15;
    const float sl2 = a.scale * RAT_LOG2E;
    const int64_t ntask = a.nseq * a.heads;
    for (int e = threadIdx.x; e < 2 * (L16 + SPARE) * CM_LD; e += CM_THREADS) ks[e] = 0.f;
    __syncthreads();
    for (int64_t task = blockIdx.x; task < ntask; task += gridDim.x) {
        const int64_t sq = task / a.heads;
        const int h = (int)(task - sq * a.heads);
        for (int e = threadIdx.x; e < L * (CM_DH / 2); e += CM_THREADS) {       // K, V rows of the pair -> LDS (8-byte pieces)
            const int j = e / (CM_DH / 2), c2 = e - j * (CM_DH / 2);
            const float* row = a.qkv + core_token(a, sq, j) * (3 * I) + h * CM_DH + 2 * c2;
            *reinterpret_cast<float2*>(ks + j * CM_LD + 2 * c2) = *reinterpret_cast<const float2*>(row + I);
            *reinterpret_cast<float2*>(vs + j * CM_LD + 2 * c2) = *reinterpret_cast<const float2*>(row + 2 * I);
        }
        __syncthreads();
        for (int qt = w; qt < NT; qt += CM_WAVES) {
            const int qi = 16 * qt + m;
            const bool qok = qi < L;
            const int64_t tok = core_token(a, sq, qok ? qi : 0);
            float bq[3];
#pragma unroll
            for (int ks_ = 0; ks_ < 3; ++ks_) {
                const int c = 4 * ks_ + g;
                bq[ks_] = (qok && c < CM_DH) ? a.qkv[tok * (3 * I) + h * CM_DH + c] : 0.f;
            }
            float mx = -INFINITY, lsum = 0.f;
            f32x4 ot = rat_zero4(), ot2 = rat_zero4();
            for (int kt0 = 0; kt0 < NT; kt0 += CM_KB) {                         // (a second tile past the sequence reads the spare zero rows)
                float ak[3][CM_KB], av[4][CM_KB];
#pragma unroll
                for (int jt = 0; jt < CM_KB; ++jt) {
#pragma unroll
                    for (int ks_ = 0; ks_ < 3; ++ks_) ak[ks_][jt] = ks[(16 * (kt0 + jt) + m) * CM_LD + 4 * ks_ + g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // (lanes m >= 10 hold a column that is no head column — zeros in 10, 11, column 0 again from 12 on: as the A operand they
                        //  only produce output ROWS c >= 10 of O^T, which are never stored: no select needed)
                        av[r][jt] = vs[(16 * (kt0 + jt) + 4 * g + r) * CM_LD + (m < CM_LD ? m : 0)];
                    }
                }
                f32x4 st[CM_KB];
#pragma unroll
                for (int jt = 0; jt < CM_KB; ++jt) st[jt] = rat_zero4();
#pragma unroll
                for (int ks_ = 0; ks_ < 3; ++ks_)
#pragma unroll
                    for (int jt = 0; jt < CM_KB; ++jt) st[jt] = RAT_MFMA16(ak[ks_][jt], bq[ks_], st[jt]);
                float bm = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < CM_KB; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[jt][r] = 16 * (kt0 + jt) + 4 * g + r < L ? st[jt][r] * sl2 : -INFINITY;
                        bm = fmaxf(bm, st[jt][r]);
                    }
                const float mn = fmaxf(mx, cm_rows_max(bm));                    // (finite: the block's first key exists)
                const float corr = rat_exp2(mx - mn);
                lsum *= corr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ot[r] *= corr;
                    ot2[r] *= corr;
                }
#pragma unroll
                for (int jt = 0; jt < CM_KB; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[jt][r] = rat_exp2(st[jt][r] - mn);
                        lsum += st[jt][r];
                    }
#pragma unroll
                for (int jt = 0; jt < CM_KB; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                               // two accumulators: two independent MFMA chains
                        if ((jt & 1) == 0) ot = RAT_MFMA16(av[r][jt], st[jt][r], ot);
                        else ot2 = RAT_MFMA16(av[r][jt], st[jt][r], ot2);
                    }
                mx = mn;
            }
            const float lt = cm_rows_sum(lsum);
            if (qok) {
                const float inv = 1.0f / lt;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < CM_DH) a.o_out[tok * I + h * CM_DH + 4 * g + r] = (ot[r] + ot2[r]) * inv;
                if (g == 0 && a.lse_out != nullptr) a.lse_out[tok * a.heads + h] = mx + rat_log2(lt);
            }
        }
        __syncthreads();                                                        // (the next pair's rows overwrite the tiles)
    }
}
// dispatch: dim_head 10, sequences of 48 ... 1024 tokens (tiles at least three quarters full; the tiles fit LDS), 8-byte aligned rows
bool core_fwd_mfma_ok(const CoreArgs& a) {
    return a.dh == CM_DH && a.L >= 48 && a.L <= 1024 && (reinterpret_cast<uintptr_t>(a.qkv) & 7) == 0 && rat_knob(RAT_KNOB_ATTN_FWD_CORE_MFMA) != 3;
}

// ---- the backward core for LONG sequences at dim_head 10 on the matrix pipe, exact fp32 (round 6) ------------------------------------------
// The backward is two independent passes over the (query, key) pairs of a (sequence, head): dQ_i = sum_j dS_ij K_j and
// (dK_j, dV_j) = sum_i (dS_ij Q_i, P_ij dO_i), both from P_ij = exp2(s_ij - lse_i) and dP_ij = dO_i . V_j.  core_bwd_kernel runs them on
// the VALU at ~38 instructions per 64 pairs; here both run on v_mfma_f32_16x16x4_f32, each in the orientation whose ACCUMULATORS are
// directly the operand of its next product, so nothing is transposed, nothing passes through LDS after the staging, and no wave
// needs another wave's data (every output element is written once: bit-reproducible).  A 512-thread work-group owns one pair, its Q, K, V,
// dO rows staged in LDS ([L16 + 16][12], zero padded), lse / delta beside them:
//   pass A (a wave owns 16-query tiles; the forward kernel's construction): S^T = K Q^T, dP^T = V dO^T (three k steps each, Q / dO of the
//       tile are the B operands, in registers across the key loop); lane (g, m) holds keys 4 g + r of query m, so lse_i and delta_i are
//       per-lane scalars; dS^T = p (dP^T - delta) on the accumulators, which then ARE the B operand of dQ^T += K^T dS^T (k step (jt, r)
//       contracts over the keys {16 jt + 4 g + r}): ten MFMAs per (query tile, key tile);
//   pass B (a wave owns CB_KT consecutive 16-key tiles, their K / V operands and dK / dV accumulators in registers across the query
//       loop): S = Q K^T, dP = dO V^T with the QUERY rows as the A operand — lane (g, m) holds queries 4 g + r of key m — and P, dS on
//       the accumulators are the A operand (rows = keys, k = those queries) of dV += P^T dO and dK += dS^T Q: fourteen MFMAs per pair of tiles.
// 24 MFMAs of 32 cycles per 256 (query, key) pairs against ~150 VALU instructions; the elementwise work (exp2, masks) is ~70 instructions.
// Measured at RAT_m0's shape (4096 sequences of 231 tokens, 8 heads): profiles/round6/r6_attn_core_bwd_ab.txt.  The first form of the
// round — dQ on the matrix pipe BESIDE the VALU (dK, dV) pass — was no faster than the VALU kernel (tools/experiments/attn_core_bwd_hybrid.hip.txt).
// Work-group shape by same-box A/B at L = 231 (profiles/round6/r6_attn_core_bwd_ab.txt): 8 waves, 3 key tiles per trip of pass A, groups of
// 2 key tiles in pass B: 4.00-4.05 ms; 4 / 5 waves per work-group 4.7-6.4 ms (the kernel lives on waves in flight), 16 waves 4.8 ms (one
// work-group per CU: staging no longer overlaps), register caps for 5 / 6 waves per SIMD: no change.
constexpr int CB_THREADS = 512, CB_WAVES = CB_THREADS / 64, CB_LD = 12, CB_DH = 10, CB_KB = 3, CB_KT = 2;
__global__ void __launch_bounds__(CB_THREADS) core_bwd_mfma_kernel(CoreArgs a) {
    RAT_DYN_SMEM(smem);
    const int L = a.L, I = a.heads * CB_DH, L16 = (L + 15) / 16 * 16, NT = L16 / 16;
    constexpr int SPARE = 16 * ((CB_KB > CB_KT ? CB_KB : CB_KT) - 1);   // zero rows past the sequence: a trip's / a group's tiles beyond it
    const int LR = L16 + SPARE;
    float* qs = reinterpret_cast<float*>(smem);             // [LR][12] each: rows >= L and columns 10, 11 are zero
    float* ks = qs + (size_t)LR * CB_LD;
    float* vs = ks + (size_t)LR * CB_LD;
    float* gs = vs + (size_t)LR * CB_LD;                    // dO
    float* ls = gs + (size_t)LR * CB_LD;                    // [L16] lse (+inf beyond the sequence: p = 0 there)
    float* ds = ls + L16;                                   // [L16] delta = dO . O
    const int w = rat_wave(), l = rat_lane(), g = l >> 4, m = l & 15;
    const int mc = m < CB_LD ? m : 0;                       // this lane's column of a [.][dim_head] operand (masked by cm below)
    const bool cm = m < CB_DH;
    const float sl2 = a.scale * RAT_LOG2E;
    const int64_t ntask = a.nseq * a.heads;
    for (int e = threadIdx.x; e < 4 * LR * CB_LD; e += CB_THREADS) qs[e] = 0.f;
    __syncthreads();
    for (int64_t task = blockIdx.x; task < ntask; task += gridDim.x) {
        const int64_t sq = task / a.heads;
        const int h = (int)(task - sq * a.heads);
        for (int e = threadIdx.x; e < L * (CB_DH / 2); e += CB_THREADS) {       // the pair's rows -> LDS (8-byte pieces)
            const int j = e / (CB_DH / 2), c2 = e - j * (CB_DH / 2);
            const int64_t tok = core_token(a, sq, j);
            const float* row = a.qkv + tok * (3 * I) + h * CB_DH + 2 * c2;
            *reinterpret_cast<float2*>(qs + j * CB_LD + 2 * c2) = *reinterpret_cast<const float2*>(row);
            *reinterpret_cast<float2*>(ks + j * CB_LD + 2 * c2) = *reinterpret_cast<const float2*>(row + I);
            *reinterpret_cast<float2*>(vs + j * CB_LD + 2 * c2) = *reinterpret_cast<const float2*>(row + 2 * I);
            *reinterpret_cast<float2*>(gs + j * CB_LD + 2 * c2) = *reinterpret_cast<const float2*>(a.dout + tok * I + h * CB_DH + 2 * c2);
        }
        for (int i = threadIdx.x; i < L16; i += CB_THREADS) {
            float lse = INFINITY, dsum = 0.f;
            if (i < L) {
                const int64_t tok = core_token(a, sq, i);
                lse = a.lse_in[tok * a.heads + h];
                const float* go = a.dout + tok * I + h * CB_DH;
                const float* oo = a.o + tok * I + h * CB_DH;
#pragma unroll
                for (int c = 0; c < CB_DH; ++c) dsum = fmaf(go[c], oo[c], dsum);
            }
            ls[i] = lse;
            ds[i] = dsum;
        }
        __syncthreads();
        // ---- pass A: dQ, one 16-query tile at a time
        for (int qt = w; qt < NT; qt += CB_WAVES) {
            const int qi = 16 * qt + m;
            float bq[3], bg[3];
#pragma unroll
            for (int k_ = 0; k_ < 3; ++k_) {
                bq[k_] = qs[qi * CB_LD + 4 * k_ + g];                             // (columns 10, 11 and rows >= L are zero in LDS)
                bg[k_] = gs[qi * CB_LD + 4 * k_ + g];
            }
            const float lse_i = ls[qi], delta_i = ds[qi];
            f32x4 dq = rat_zero4(), dq2 = rat_zero4();
            for (int kt0 = 0; kt0 < NT; kt0 += CB_KB) {                          // (a second tile past the sequence reads the spare zero rows)
                // (requesting the operands of trip t + 1 before the MFMAs of trip t by hand: 145 VGPRs, and 4.33 ms against 4.02 at 128)
                float ak[3][CB_KB], av[3][CB_KB], akc[4][CB_KB];
#pragma unroll
                for (int jt = 0; jt < CB_KB; ++jt) {
#pragma unroll
                    for (int k_ = 0; k_ < 3; ++k_) {
                        ak[k_][jt] = ks[(16 * (kt0 + jt) + m) * CB_LD + 4 * k_ + g];
                        av[k_][jt] = vs[(16 * (kt0 + jt) + m) * CB_LD + 4 * k_ + g];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) akc[r][jt] = ks[(16 * (kt0 + jt) + 4 * g + r) * CB_LD + mc];
                }
                f32x4 st[CB_KB], dp[CB_KB];
#pragma unroll
                for (int jt = 0; jt < CB_KB; ++jt) st[jt] = dp[jt] = rat_zero4();
#pragma unroll
                for (int k_ = 0; k_ < 3; ++k_)
#pragma unroll
                    for (int jt = 0; jt < CB_KB; ++jt) {
                        st[jt] = RAT_MFMA16(ak[k_][jt], bq[k_], st[jt]);        // S^T[key 16 jt + 4 g + r][query m]
                        dp[jt] = RAT_MFMA16(av[k_][jt], bg[k_], dp[jt]);        // dP^T, same layout
                    }
                if (16 * (kt0 + CB_KB) <= L) {                                   // (wave-uniform: every key of the trip exists — no masks)
#pragma unroll
                    for (int jt = 0; jt < CB_KB; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) st[jt][r] = rat_exp2(st[jt][r] * sl2 - lse_i) * (dp[jt][r] - delta_i);   // dS^T
                } else {
#pragma unroll
                    for (int jt = 0; jt < CB_KB; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pr = 16 * (kt0 + jt) + 4 * g + r < L ? rat_exp2(st[jt][r] * sl2 - lse_i) : 0.f;
                            st[jt][r] = pr * (dp[jt][r] - delta_i);
                        }
                }
#pragma unroll
                for (int jt = 0; jt < CB_KB; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                                // dQ^T[c][query] += K[key][c] dS^T[key][query]; two independent chains
                        // (akc of a lane m >= 10 is no head column: it only reaches rows c >= 10 of dQ^T, which are never stored)
                        if ((jt & 1) == 0) dq = RAT_MFMA16(akc[r][jt], st[jt][r], dq);
                        else dq2 = RAT_MFMA16(akc[r][jt], st[jt][r], dq2);
                    }
            }
            if (qi < L) {                                                        // dq[r] = dQ[query 16 qt + m][c = 4 g + r]
                float* out = a.dqkv + core_token(a, sq, qi) * (3 * I) + h * CB_DH + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < CB_DH) out[r] = (dq[r] + dq2[r]) * a.scale;
            }
        }
        // ---- pass B: dK, dV, CB_KT key tiles at a time (their operands and accumulators stay in registers across the query loop)
        for (int kg = w; kg * CB_KT < NT; kg += CB_WAVES) {
            float bk[3][CB_KT], bv[3][CB_KT];
            f32x4 adk[CB_KT], adv[CB_KT];
#pragma unroll
            for (int t = 0; t < CB_KT; ++t) {
                const int kj = 16 * (kg * CB_KT + t) + m;                        // (a tile past the sequence: the spare zero rows)
#pragma unroll
                for (int k_ = 0; k_ < 3; ++k_) {
                    bk[k_][t] = ks[kj * CB_LD + 4 * k_ + g];
                    bv[k_][t] = vs[kj * CB_LD + 4 * k_ + g];
                }
                adk[t] = adv[t] = rat_zero4();
            }
            for (int it = 0; it < NT; ++it) {
                const int i0 = 16 * it;
                float aq[3], ag[3], cq[4], cg[4];
#pragma unroll
                for (int k_ = 0; k_ < 3; ++k_) {
                    aq[k_] = qs[(i0 + m) * CB_LD + 4 * k_ + g];                   // A of S:  Q[query i0 + m][c = 4 k + g]
                    ag[k_] = gs[(i0 + m) * CB_LD + 4 * k_ + g];                   // A of dP: dO, same layout
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cq[r] = qs[(i0 + 4 * g + r) * CB_LD + mc];                    // B of dK: Q[query i0 + 4 g + r][c = m]
                    cg[r] = gs[(i0 + 4 * g + r) * CB_LD + mc];                    // B of dV: dO, same layout
                }
                const float4 lse4 = *reinterpret_cast<const float4*>(ls + i0 + 4 * g);     // queries i0 + 4 g + r (+inf past the sequence)
                const float4 dl4 = *reinterpret_cast<const float4*>(ds + i0 + 4 * g);
                const float lse_r[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dl_r[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
                f32x4 sa[CB_KT], da[CB_KT];
#pragma unroll
                for (int t = 0; t < CB_KT; ++t) sa[t] = da[t] = rat_zero4();
#pragma unroll
                for (int k_ = 0; k_ < 3; ++k_)
#pragma unroll
                    for (int t = 0; t < CB_KT; ++t) {
                        sa[t] = RAT_MFMA16(aq[k_], bk[k_][t], sa[t]);           // S[query i0 + 4 g + r][key m of tile t]
                        da[t] = RAT_MFMA16(ag[k_], bv[k_][t], da[t]);           // dP, same layout
                    }
                // (keys past the sequence have zero K / V rows: their columns hold p = exp2(-lse_i), which only reaches dK / dV rows
                //  that are never stored; queries past it have lse = +inf: p = 0)
#pragma unroll
                for (int t = 0; t < CB_KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pr = rat_exp2(sa[t][r] * sl2 - lse_r[r]);
                        sa[t][r] = pr;
                        da[t][r] = pr * (da[t][r] - dl_r[r]);                    // dS
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (cq / cg of a lane m >= 10 are no head column: as the B operand they only reach COLUMNS c >= 10 of dK / dV: never stored)
#pragma unroll
                    for (int t = 0; t < CB_KT; ++t) {
                        adv[t] = RAT_MFMA16(sa[t][r], cg[r], adv[t]);           // dV[key 4 g' + r'][c m] += P[query][key] dO[query][c]
                        adk[t] = RAT_MFMA16(da[t][r], cq[r], adk[t]);           // dK += dS[query][key] Q[query][c]
                    }
                }
            }
            if (cm)
#pragma unroll
                for (int t = 0; t < CB_KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * (kg * CB_KT + t) + 4 * g + r;
                        if (j < L) {
                            float* drow = a.dqkv + core_token(a, sq, j) * (3 * I) + h * CB_DH + m;
                            drow[I] = adk[t][r] * a.scale;
                            drow[2 * I] = adv[t][r];
                        }
                    }
        }
        __syncthreads();                                                        // (the next pair's rows overwrite the tiles)
    }
}
size_t core_bwd_mfma_smem(int L) {
    const int L16 = (L + 15) / 16 * 16, spare = 16 * ((CB_KB > CB_KT ? CB_KB : CB_KT) - 1);
    return ((size_t)4 * (L16 + spare) * CB_LD + 2 * (size_t)L16) * sizeof(float);
}
// dispatch: dim_head 10, sequences of 48+ tokens whose four row tiles fit the LDS (<= ~800 tokens), 8-byte aligned rows; the
// attn_bwd_core_mfma knob's value 0 (RAT_ATTN_BWD_CORE=valu) keeps the two VALU passes
bool core_bwd_mfma_ok(const CoreArgs& a) {
    const int I = a.heads * a.dh;
    return a.dh == CB_DH && a.L >= 48 && core_bwd_mfma_smem(a.L) <= 160 * 1024 && I % 2 == 0 &&
           ((reinterpret_cast<uintptr_t>(a.qkv) | reinterpret_cast<uintptr_t>(a.dout)) & 7) == 0 && rat_knob(RAT_KNOB_ATTN_BWD_CORE_MFMA) != 0;
}

template <template <int, bool> class Launch, bool MULTI>
int core_dispatch_dh(int dh, const CoreArgs& a, unsigned grid, int threads, size_t smem, void* stream) {
    switch (dh) {
        case 4: return Launch<4, MULTI>::go(a, grid, threads, smem, stream);
        case 8: return Launch<8, MULTI>::go(a, grid, threads, smem, stream);
        case 10: return Launch<10, MULTI>::go(a, grid, threads, smem, stream);
        case 16: return Launch<16, MULTI>::go(a, grid, threads, smem, stream);
        case 20: return Launch<20, MULTI>::go(a, grid, threads, smem, stream);
        default: return Launch<0, MULTI>::go(a, grid, threads, smem, stream);
    }
}
template <template <int, bool> class Launch>
int core_dispatch(int dh, const CoreArgs& a, unsigned grid, int threads, size_t smem, void* stream) {
    return threads / a.L > 1 ? core_dispatch_dh<Launch, true>(dh, a, grid, threads, smem, stream)
                             : core_dispatch_dh<Launch, false>(dh, a, grid, threads, smem, stream);
}
template <int DH, bool MULTI>
struct LaunchFwd {
    static int go(const CoreArgs& a, unsigned grid, int threads, size_t smem, void* stream) {
        RAT_LAUNCH((core_fwd_kernel<DH, MULTI>), grid, threads, smem, stream, a);
        return rat_check_launch("rat_attn_core_fwd");
    }
};
template <int DH, bool MULTI>
struct LaunchBwd {
    static int go(const CoreArgs& a, unsigned grid, int threads, size_t smem, void* stream) {
        RAT_LAUNCH((core_bwd_kernel<DH, MULTI>), grid, threads, smem, stream, a);
        return rat_check_launch("rat_attn_core_bwd");
    }
};

unsigned core_grid(int64_t tasks) { return (unsigned)(tasks < 65536 ? tasks : 65536); }

}  // namespace

namespace {
RatSeqMap contiguous_map(int64_t nseq, int L) {
    RatSeqMap m{};
    m.nseq = nseq;
    m.L = L;
    m.q_div = nseq;
    m.hi_stride = 0;
    m.lo_stride = L;
    m.pos_stride = 1;
    return m;
}
void fill_map(CoreArgs& a, const RatSeqMap* m) {
    a.nseq = m->nseq;
    a.L = m->L;
    a.q_div = m->q_div;
    a.hi_stride = m->hi_stride;
    a.lo_stride = m->lo_stride;
    a.pos_stride = m->pos_stride;
}
}  // namespace

extern "C" int rat_attn_core_fwd(const float* qkv, float* o, float* lse, int64_t nseq, int L, int heads, int dim_head,
                                 float softmax_scale, void* stream) {
    RAT_REQUIRE(nseq > 0 && L > 0, "bad dims");
    const RatSeqMap m = contiguous_map(nseq, L);
    return rat_attn_core_fwd_map(qkv, o, lse, &m, heads, dim_head, softmax_scale, stream);
}

extern "C" int rat_attn_core_fwd_map(const float* qkv, float* o, float* lse, const RatSeqMap* map_host, int heads, int dim_head,
                                     float softmax_scale, void* stream) {
    RAT_REQUIRE(map_host != nullptr && map_host->q_div >= 1, "bad seq map");
    const int64_t nseq = map_host->nseq;
    const int L = map_host->L;
    const size_t smem = (size_t)2 * core_pairs(L) * L * dim_head * sizeof(float);
    if (core_check(nseq, L, heads, dim_head, smem)) return -1;
    RAT_REQUIRE(qkv && o, "null pointer");
    CoreArgs a{};
    a.qkv = qkv;
    a.o_out = o;
    a.lse_out = lse;
    fill_map(a, map_host);
    a.heads = heads;
    a.dh = dim_head;
    a.scale = softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)dim_head);
    if (core_fwd_mfma_ok(a)) {
        const int L16 = (L + 15) / 16 * 16;
        const int64_t tasks = nseq * heads, cap = (int64_t)rat_max_blocks() * 8;
        RAT_LAUNCH(core_fwd_mfma_kernel, (unsigned)(tasks < cap ? tasks : cap), CM_THREADS, (size_t)2 * (L16 + 16 * (CM_KB - 1)) * CM_LD * sizeof(float), stream, a);
        return rat_check_launch("rat_attn_core_fwd (matrix pipe)");
    }
    return core_dispatch<LaunchFwd>(dim_head, a, core_grid((nseq * heads + core_pairs(L) - 1) / core_pairs(L)), core_threads(L), smem, stream);
}

extern "C" int rat_attn_core_bwd(const float* qkv, const float* o, const float* lse, const float* dout, float* dqkv,
                                 int64_t nseq, int L, int heads, int dim_head, float softmax_scale, void* stream) {
    RAT_REQUIRE(nseq > 0 && L > 0, "bad dims");
    const RatSeqMap m = contiguous_map(nseq, L);
    return rat_attn_core_bwd_map(qkv, o, lse, dout, dqkv, &m, heads, dim_head, softmax_scale, stream);
}

extern "C" int rat_attn_core_bwd_map(const float* qkv, const float* o, const float* lse, const float* dout, float* dqkv,
                                     const RatSeqMap* map_host, int heads, int dim_head, float softmax_scale, void* stream) {
    RAT_REQUIRE(map_host != nullptr && map_host->q_div >= 1, "bad seq map");
    const int64_t nseq = map_host->nseq;
    const int L = map_host->L;
    const size_t smem = (size_t)core_pairs(L) * ((size_t)4 * L * dim_head + 2 * (size_t)L) * sizeof(float);
    if (core_check(nseq, L, heads, dim_head, smem)) return -1;
    RAT_REQUIRE(qkv && o && lse && dout && dqkv, "null pointer");
    CoreArgs a{};
    a.qkv = qkv;
    a.o = o;
    a.lse_in = lse;
    a.dout = dout;
    a.dqkv = dqkv;
    fill_map(a, map_host);
    a.heads = heads;
    a.dh = dim_head;
    a.scale = softmax_scale > 0.f ? softmax_scale : 1.0f / sqrtf((float)dim_head);
    if (core_bwd_mfma_ok(a)) {
        const int64_t tasks = nseq * heads, cap = (int64_t)rat_max_blocks() * 4;
        RAT_LAUNCH(core_bwd_mfma_kernel, (unsigned)(tasks < cap ? tasks : cap), CB_THREADS, core_bwd_mfma_smem(L), stream, a);
        return rat_check_launch("rat_attn_core_bwd (matrix pipe)");
    }
    return core_dispatch<LaunchBwd>(dim_head, a, core_grid((nseq * heads + core_pairs(L) - 1) / core_pairs(L)), core_threads(L), smem, stream);
}
