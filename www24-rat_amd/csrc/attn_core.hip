// attn_core.hip — K2d: the softmax(Q K^T * scale) V core for LONG sequences, forward and backward, on projected Q|K|V rows.
//
// RAT_m0 (RAT_m0.py:123-127) attends jointly over all T*S tokens of a sample (231 at the north-star shape): a whole
// sequence no longer fits the 64-row LDS tile of the fused kernel in attn.hip, so that variant runs the projections as
// plain MFMA GEMMs (rat_sgemm) and LayerNorm as K2c, and only this core is new.  One work-group per (sequence, head):
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

// LDS rows are unpadded [L][dh]: in the key / query walks every lane of a wave reads the SAME row (a broadcast, no bank
// conflict), and the staging writes are contiguous.
template <int DH>
__global__ void __launch_bounds__(1024) core_fwd_kernel(CoreArgs a) {
    RAT_DYN_SMEM(smem);
    const int dh = DH > 0 ? DH : a.dh, L = a.L, I = a.heads * dh;
    float* ks = reinterpret_cast<float*>(smem);          // [L][dh]
    float* vs = ks + (size_t)L * dh;                     // [L][dh]
    const float sl2 = a.scale * RAT_LOG2E;
    for (int64_t task = blockIdx.x; task < a.nseq * a.heads; task += gridDim.x) {
        const int64_t sq = task / a.heads;
        const int h = (int)(task - sq * a.heads);
        const float* base = a.qkv + h * dh;
        for (int e = threadIdx.x; e < L * dh; e += blockDim.x) {
            const int j = e / dh, c = e - j * dh;
            const float* row = base + core_token(a, sq, j) * (3 * I);
            ks[e] = row[I + c];
            vs[e] = row[2 * I + c];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            Vec<DH> q, o, kv;
            const int64_t tok = core_token(a, sq, i);
            q.load(base + tok * (3 * I), dh);
            o.zero();
            float m = -3.0e38f, l = 0.f;
            for (int j = 0; j < L; ++j) {
                kv.load(ks + (size_t)j * dh, dh);
                const float s = q.dot(kv) * sl2;
                const float mn = fmaxf(m, s);
                const float corr = rat_exp2(m - mn), p = rat_exp2(s - mn);
                l = l * corr + p;
                kv.load(vs + (size_t)j * dh, dh);
                o.scale_axpy(corr, p, kv);
                m = mn;
            }
            o.store(a.o_out + tok * I + h * dh, dh, 1.0f / l);
            if (a.lse_out != nullptr) a.lse_out[tok * a.heads + h] = m + rat_log2(l);
        }
        __syncthreads();
    }
}

template <int DH>
__global__ void __launch_bounds__(1024) core_bwd_kernel(CoreArgs a) {
    RAT_DYN_SMEM(smem);
    const int dh = DH > 0 ? DH : a.dh, L = a.L, I = a.heads * dh;
    float* qs = reinterpret_cast<float*>(smem);          // [L][dh]
    float* ks = qs + (size_t)L * dh;
    float* vs = ks + (size_t)L * dh;
    float* gs = vs + (size_t)L * dh;                     // dO
    float* ls = gs + (size_t)L * dh;                     // [L] lse
    float* ds = ls + L;                                  // [L] delta = dO . O
    const float sl2 = a.scale * RAT_LOG2E;
    for (int64_t task = blockIdx.x; task < a.nseq * a.heads; task += gridDim.x) {
        const int64_t sq = task / a.heads;
        const int h = (int)(task - sq * a.heads);
        const float* base = a.qkv + h * dh;
        const float* gbase = a.dout + h * dh;
        const float* obase = a.o + h * dh;
        for (int e = threadIdx.x; e < L * dh; e += blockDim.x) {
            const int j = e / dh, c = e - j * dh;
            const int64_t tok = core_token(a, sq, j);
            const float* row = base + tok * (3 * I);
            qs[e] = row[c];
            ks[e] = row[I + c];
            vs[e] = row[2 * I + c];
            gs[e] = gbase[tok * I + c];
        }
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            const int64_t tok = core_token(a, sq, i);
            ls[i] = a.lse_in[tok * a.heads + h];
            float dsum = 0.f;
            for (int c = 0; c < dh; ++c) dsum = fmaf(gbase[tok * I + c], obase[tok * I + c], dsum);
            ds[i] = dsum;
        }
        __syncthreads();
        float* dbase = a.dqkv + h * dh;
        // pass 1: one lane per query row -> dQ
        for (int i = threadIdx.x; i < L; i += blockDim.x) {
            Vec<DH> q, go, dq, kv;
            q.load(qs + (size_t)i * dh, dh);
            go.load(gs + (size_t)i * dh, dh);
            dq.zero();
            const float lse = ls[i], delta = ds[i];
            for (int j = 0; j < L; ++j) {
                kv.load(vs + (size_t)j * dh, dh);
                const float dp = go.dot(kv);
                kv.load(ks + (size_t)j * dh, dh);
                const float p = rat_exp2(q.dot(kv) * sl2 - lse);
                dq.axpy(p * (dp - delta), kv);
            }
            dq.store(dbase + core_token(a, sq, i) * (3 * I), dh, a.scale);
        }
        // pass 2: one lane per key row -> dK, dV
        for (int j = threadIdx.x; j < L; j += blockDim.x) {
            Vec<DH> kk, vv, dk, dv, t, qv;
            kk.load(ks + (size_t)j * dh, dh);
            vv.load(vs + (size_t)j * dh, dh);
            dk.zero();
            dv.zero();
            for (int i = 0; i < L; ++i) {
                t.load(gs + (size_t)i * dh, dh);
                const float dp = t.dot(vv);
                qv.load(qs + (size_t)i * dh, dh);
                const float p = rat_exp2(qv.dot(kk) * sl2 - ls[i]);
                dv.axpy(p, t);
                dk.axpy(p * (dp - ds[i]), qv);
            }
            const int64_t tokj = core_token(a, sq, j);
            dk.store(dbase + tokj * (3 * I) + I, dh, a.scale);
            dv.store(dbase + tokj * (3 * I) + 2 * I, dh, 1.0f);
        }
        __syncthreads();
    }
}

int core_threads(int L) {
    int t = (L + 63) / 64 * 64;
    return t > 1024 ? 1024 : t;
}

int core_check(int64_t nseq, int L, int heads, int dh, size_t smem) {
    RAT_REQUIRE(nseq > 0 && L > 0 && heads > 0 && dh > 0, "bad dims");
    RAT_REQUIRE(dh <= CORE_DH_MAX, "dim_head > 32 is not supported by the long-sequence attention core");
    RAT_REQUIRE(smem <= 160 * 1024, "sequence too long for the LDS staging of one head's K, V (and Q, dO) rows");
    return 0;
}

template <template <int> class Launch>
int core_dispatch(int dh, const CoreArgs& a, unsigned grid, int threads, size_t smem, void* stream) {
    switch (dh) {
        case 4: return Launch<4>::go(a, grid, threads, smem, stream);
        case 8: return Launch<8>::go(a, grid, threads, smem, stream);
        case 10: return Launch<10>::go(a, grid, threads, smem, stream);
        case 16: return Launch<16>::go(a, grid, threads, smem, stream);
        case 20: return Launch<20>::go(a, grid, threads, smem, stream);
        default: return Launch<0>::go(a, grid, threads, smem, stream);
    }
}
template <int DH>
struct LaunchFwd {
    static int go(const CoreArgs& a, unsigned grid, int threads, size_t smem, void* stream) {
        RAT_LAUNCH((core_fwd_kernel<DH>), grid, threads, smem, stream, a);
        return rat_check_launch("rat_attn_core_fwd");
    }
};
template <int DH>
struct LaunchBwd {
    static int go(const CoreArgs& a, unsigned grid, int threads, size_t smem, void* stream) {
        RAT_LAUNCH((core_bwd_kernel<DH>), grid, threads, smem, stream, a);
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
    const size_t smem = (size_t)2 * L * dim_head * sizeof(float);
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
    return core_dispatch<LaunchFwd>(dim_head, a, core_grid(nseq * heads), core_threads(L), smem, stream);
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
    const size_t smem = ((size_t)4 * L * dim_head + 2 * (size_t)L) * sizeof(float);
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
    return core_dispatch<LaunchBwd>(dim_head, a, core_grid(nseq * heads), core_threads(L), smem, stream);
}
