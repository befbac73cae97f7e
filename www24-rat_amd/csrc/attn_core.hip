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
#define RAT_CM_KB 4
#endif
constexpr int CM_THREADS = 256, CM_WAVES = 4, CM_LD = 12, CM_DH = 10, CM_KB = RAT_CM_KB;      // CM_KB: 16-key tiles per trip of the key loop
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
                        const float t = vs[(16 * (kt0 + jt) + 4 * g + r) * CM_LD + (m < CM_LD ? m : 0)];
                        av[r][jt] = m < CM_DH ? t : 0.f;
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
    return core_dispatch<LaunchBwd>(dim_head, a, core_grid((nseq * heads + core_pairs(L) - 1) / core_pairs(L)), core_threads(L), smem, stream);
}
