// attn.hip — K2a: one attention phase of the reference's CrossIntraEncoderBlock, forward and backward.
//
//   y = to_out(softmax(Q K^T * dh^-0.5) V) + x,   [Q|K|V] = LayerNorm(x) W_qkv^T      (RAT_m2.py:155-161,176-202)
//
// over groups ("sequences") of L tokens that a RatSeqMap addresses inside the [B][T][S][d] grid: the S field
// tokens of one sample (intra, RAT_m2.py:222-224) or the T samples at one field position (cross,
// RAT_m2.py:226-230).  The reference materialises a transposed copy for the cross phase; here it is only a
// different token stride.
//
// Work decomposition (same for both phases): a 512-thread work-group (8 waves) owns a chunk of up to 64 token
// rows = floor(64/L) whole sequences, staged in LDS:
//     xs  [64][D16+4]   LayerNorm'ed tokens (A operand of the QKV projection, 16-byte row reads)
//     qkv [64][3I16+4]  projection output; the softmax(QK^T)V result overwrites the Q columns in place
// The projections run on v_mfma_f32_16x16x4_f32 with the weights streamed from L2 as B operands (one k-block
// ahead of the MFMAs); the (L x L x dh) attention core of a sequence-head is far too small and ragged for a
// 16x16 MFMA tile at fp32 (fp32 MFMA peak == fp32 VALU peak on gfx950, and an 11x11x10 problem fills 39 % of a
// padded tile), so it runs on the VALU, one lane per (sequence, head, query) with an online softmax.
// Backward recomputes LayerNorm and QKV from x, re-derives P from the saved log-sum-exp, and keeps every
// weight gradient in MFMA accumulators across the work-group's whole chunk loop (written once per launch
// to a per-work-group slab, then summed in fixed order => deterministic).
//
// Two code paths from ONE source: template <TD, TDH>.  TD > 0 ("fast"): embedding_dim == TD and heads*dim_head
// are multiples of 16, dim_head == TDH, 16-byte aligned pointers — every bounds guard compiles away and loops
// over d / dim_head unroll.  TD == 0: any shape within the limits below, guarded loads, zero-padded tiles.
#include "rat_device.h"
#include "../../include/rat_hip.h"

#include <cstdlib>
#include <initializer_list>

namespace {

constexpr int ATT_THREADS = 512;
constexpr int ATT_WAVES = ATT_THREADS / 64;
constexpr int ATT_ROWS = 64;
constexpr int ATT_MT = ATT_ROWS / 16;
constexpr int DH_MAX = 16;        // dim_head <= 16 (every shipped config uses 10)
constexpr int QSLOTS = 8;         // persistent dW_qkv tiles per wave  (3I16/16 * D16/16 <= 64)
constexpr int OSLOTS = 4;         // persistent dW_out tiles per wave  (D16/16 * I16/16 <= 32)
constexpr int FAST_INNER = 80;    // the compiled fast shapes fix heads x dim_head = 8 x 10 (or 4 x 20, RAT_m3): the whole LDS geometry is compile-time
constexpr int fast_heads(int tdh) { return tdh > 0 ? FAST_INNER / tdh : 0; }
constexpr int CORE_UNROLL = 3;    // keys (queries) per trip of the VALU attention-core loops

struct AttnArgs {
    const float* x;
    const float* dy;
    float* y;            // forward output / backward dx
    const float* res;    // forward: residual source (x for PreNorm(Attention) + x; y itself to accumulate; nullptr: none)
    const float* add;    // backward: gradient added to the LayerNorm-backward result (dy for the residual; nullptr: none)
    int add_lds;         // backward: add == dy and out_scale == 1 -> the residual gradient is the dy tile already in LDS
    float out_scale;     // y = out_scale * to_out(...) + res   (RAT_m3's mean of two attentions: 0.5)
    float* o_save;
    float* lse_save;
    const float* ln_g;
    const float* ln_b;
    const float* w_qkv;
    const float* w_out;
    const float* b_out;
    float* slabs;        // backward: [gridDim.x][slab_stride]
    int64_t slab_stride;
    int64_t nseq, q_div, hi_stride, lo_stride, pos_stride;
    int64_t nchunks;
    int L, nsq_chunk;
    int nq;              // RatSeqMap.queries: only the first nq positions of a sequence are QUERIES that matter (L: all of them)
    int d, heads, dh;
    float eps, scale;
    int vec_x, vec_wqkv, vec_wout;
    RatDrop drop;        // Dropout behind the output projection (RAT_m2.py:186-189); the EX instantiations only
    int groups;          // attn_fwd3_kernel<GRP>: head groups looped over inside a chunk (wide heads: heads = groups x 8)
    int64_t group_tok;   // ... o_save / lse_save are [groups][group_tok][.]: token stride between two groups' slices
    unsigned long long* prof;
};

struct AttnGeom {
    int D, I, Q3, D16, I16, Q16, ldx, ldq, ldt;
    __host__ __device__ AttnGeom(int d, int heads, int dh) {
        D = d;
        I = heads * dh;
        Q3 = 3 * I;
        D16 = (D + 15) / 16 * 16;
        I16 = (I + 15) / 16 * 16;
        Q16 = (Q3 + 15) / 16 * 16;
        ldx = D16 + 4;
        ldq = Q16 + 4;
        ldt = (D16 > I16 ? D16 : I16) + 4;
    }
    size_t fwd_smem() const { return (size_t)ATT_ROWS * (ldx + ldq) * 4 + 2 * ATT_ROWS * 8; }
    size_t bwd_smem(int heads) const {
        return (size_t)ATT_ROWS * (2 * ldx + ldq + 2 * ldt) * 4 + (size_t)ATT_ROWS * (2 + 2 * heads) * 4 + 2 * ATT_ROWS * 8;
    }
    int64_t slab_floats() const { return (int64_t)Q3 * D + (int64_t)D * I + 3 * (int64_t)D; }
};

__device__ __forceinline__ int64_t seq_token(const AttnArgs& a, int64_t q, int p) {
    if (a.nseq <= 0x7fffffffLL && a.q_div <= 0x7fffffffLL) {          // 32-bit divide (a 64-bit one is a ~200-instruction sequence)
        const unsigned qq = (unsigned)q, dv = (unsigned)a.q_div, hi = qq / dv;
        return (int64_t)hi * a.hi_stride + (int64_t)(qq - hi * dv) * a.lo_stride + (int64_t)p * a.pos_stride;
    }
    return (q / a.q_div) * a.hi_stride + (q % a.q_div) * a.lo_stride + (int64_t)p * a.pos_stride;
}

// rows of this chunk -> token ids (-1 for padding rows)
__device__ __forceinline__ void map_rows(const AttnArgs& a, int64_t chunk, int64_t* rowtok, int& nsq, int& rows) {
    const int64_t q0 = chunk * a.nsq_chunk;
    const int64_t left = a.nseq - q0;
    nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
    rows = nsq * a.L;
    if (threadIdx.x < ATT_ROWS) {
        const int r = threadIdx.x;
        rowtok[r] = r < rows ? seq_token(a, q0 + r / a.L, r % a.L) : (int64_t)-1;
    }
}

// (the bf16x3 kernels use this 64-bit form too: a 32-bit map and a form that cannot be hoisted out of the chunk loop were built to get
//  rid of attn_bwd3_kernel's 12 bytes of scratch and measured 0.7-1.5 % SLOWER — profiles/round5/r5_map_rows_ab.txt,
//  tools/experiments/round5_ab_branches.txt)

// load `width` floats per row from a token-indexed global array into an LDS tile (padding rows -> 0)
__device__ __forceinline__ void load_rows(float* tile, int ld, const float* src, const int64_t* rowtok, int width,
                                          bool vec, float mul = 1.0f, const RatDrop* drop = nullptr) {
    if (drop != nullptr && drop->threshold != 0) {           // dy through the projection's Dropout: element-wise mask, scalar path
        for (int e = threadIdx.x; e < ATT_ROWS * width; e += ATT_THREADS) {
            const int r = e / width, c = e - r * width;
            const int64_t tok = rowtok[r];
            tile[(size_t)r * ld + c] = tok >= 0 ? drop->apply(src[tok * width + c], tok * width + c) * mul : 0.f;
        }
        return;
    }
    if (vec) {
        const int w4 = width >> 2;
        for (int e = threadIdx.x; e < ATT_ROWS * w4; e += ATT_THREADS) {
            const int r = e / w4, c4 = e - r * w4;
            const int64_t tok = rowtok[r];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tok >= 0) v = *reinterpret_cast<const float4*>(src + tok * width + 4 * c4);
            if (mul != 1.0f) { v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul; }
            *reinterpret_cast<float4*>(tile + (size_t)r * ld + 4 * c4) = v;
        }
    } else {
        for (int e = threadIdx.x; e < ATT_ROWS * width; e += ATT_THREADS) {
            const int r = e / width, c = e - r * width;
            const int64_t tok = rowtok[r];
            tile[(size_t)r * ld + c] = tok >= 0 ? src[tok * width + c] * mul : 0.f;
        }
    }
}

// Compile-time-width form for the fast shapes: a thread's items are staged in three sweeps — all row-map reads, then all
// global loads, then all LDS writes — so that the tile costs ONE memory round trip.  (The run-time-width loop above compiles to
// one LDS read -> global load -> LDS write dependency chain per item: seven serial round trips per chunk for x, dy and O.)
template <int WIDTH>
struct RowFetch {
    static constexpr int W4 = WIDTH / 4;
    static constexpr int NIT = (ATT_ROWS * W4 + ATT_THREADS - 1) / ATT_THREADS;
    float4 v[NIT];
    __device__ __forceinline__ void issue(const float* src, const int64_t* rowtok) {
        int64_t tok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            tok[it] = e < ATT_ROWS * W4 ? rowtok[e / W4] : -1;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tok[it] >= 0) v[it] = *reinterpret_cast<const float4*>(src + tok[it] * WIDTH + 4 * (e % W4));
        }
    }
    __device__ __forceinline__ void stash(float* tile, int ld, float mul = 1.0f) const {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            if (e < ATT_ROWS * W4) {
                float4 t = v[it];
                if (mul != 1.0f) { t.x *= mul; t.y *= mul; t.z *= mul; t.w *= mul; }
                *reinterpret_cast<float4*>(tile + (size_t)(e / W4) * ld + 4 * (e % W4)) = t;
            }
        }
    }
};

// The same idea for the generic kernels' instantiations with compile-time geometry (GD = embedding_dim, GI = heads x dim_head, GH =
// heads: the shipped MovieLens shape 10 / 20 / 2, BASELINE configs[0] 16 / 20 / 2): the chunk's x, dy, O and lse are fetched as 8-byte
// pieces, every request issued before the first LDS store — ONE memory round trip instead of one per trip of four run-time loops
// (x 2, dy 2, O 3, lse 1 at d = 10).  Needs 8-byte aligned arrays (checked by the caller, a uniform branch).
template <int GD, int GI, int GH>
struct SmallFetch {
    static constexpr int PX = GD / 2, NX = (ATT_ROWS * PX + ATT_THREADS - 1) / ATT_THREADS;
    static constexpr int PO = GI / 2, NO = (ATT_ROWS * PO + ATT_THREADS - 1) / ATT_THREADS;
    static constexpr int NL = (ATT_ROWS * GH + ATT_THREADS - 1) / ATT_THREADS;
    float2 x[NX], dy[NX], o[NO];
    float l[NL];
    static __device__ __forceinline__ float2 ld2(const float* p, bool ok) {
        return ok ? *reinterpret_cast<const float2*>(p) : make_float2(0.f, 0.f);
    }
    __device__ __forceinline__ void issue(const float* xsrc, const float* dysrc, const float* osrc, const float* lsrc, const int64_t* rowtok) {
        int64_t tx[NX], to[NO], tl[NL];
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            tx[it] = e < ATT_ROWS * PX ? rowtok[e / PX] : -1;
        }
#pragma unroll
        for (int it = 0; it < NO; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            to[it] = (osrc != nullptr && e < ATT_ROWS * PO) ? rowtok[e / PO] : -1;
        }
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            tl[it] = (lsrc != nullptr && e < ATT_ROWS * GH) ? rowtok[e / GH] : -1;
        }
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            x[it] = ld2(xsrc + tx[it] * GD + 2 * (e % PX), tx[it] >= 0);
            dy[it] = ld2(dysrc + tx[it] * GD + 2 * (e % PX), dysrc != nullptr && tx[it] >= 0);
        }
#pragma unroll
        for (int it = 0; it < NO; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            o[it] = ld2(osrc + to[it] * GI + 2 * (e % PO), to[it] >= 0);
        }
#pragma unroll
        for (int it = 0; it < NL; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            l[it] = tl[it] >= 0 ? lsrc[tl[it] * GH + e % GH] : 0.f;
        }
    }
    __device__ __forceinline__ void stash(float* xs, float* dys, float* ob, float* lses, int ldx, int ldt, float dymul) const {
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            if (e < ATT_ROWS * PX) {
                *reinterpret_cast<float2*>(xs + (size_t)(e / PX) * ldx + 2 * (e % PX)) = x[it];
                if (dys != nullptr) *reinterpret_cast<float2*>(dys + (size_t)(e / PX) * ldx + 2 * (e % PX)) = make_float2(dy[it].x * dymul, dy[it].y * dymul);
            }
        }
        if (ob != nullptr) {
#pragma unroll
            for (int it = 0; it < NO; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                if (e < ATT_ROWS * PO) *reinterpret_cast<float2*>(ob + (size_t)(e / PO) * ldt + 2 * (e % PO)) = o[it];
            }
#pragma unroll
            for (int it = 0; it < NL; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                if (e < ATT_ROWS * GH) lses[e] = l[it];
            }
        }
    }
};
__device__ __forceinline__ bool aligned8_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }


// the same touch with the token taken from the NEXT chunk's row map (double-buffered maps: no 64-bit divisions here)
__device__ __forceinline__ float prefetch_lines_map(const int64_t* next_rowtok, int slot, int nlines_row, const float* src, int width) {
    const int r = slot / nlines_row, ln = slot - r * nlines_row;
    const int64_t tok = next_rowtok[r];
    if (tok < 0) return 0.f;
    int c = ln * 32;
    if (c >= width) c = width - 1;
    return src[tok * width + c];
}

__device__ __forceinline__ void zero_cols(float* tile, int ld, int c0) {   // tile[:, c0:ld] = 0
    const int w = ld - c0;
    for (int e = threadIdx.x; e < ATT_ROWS * w; e += ATT_THREADS) tile[(size_t)(e / w) * ld + c0 + e % w] = 0.f;
}

// in-place LayerNorm of the valid rows of xs: 8 lanes per row, lane `sub` owns the COLS contiguous columns
// [sub*COLS, sub*COLS + COLS) (COLS = ceil(D/8); 16-byte LDS accesses when COLS % 4 == 0); optionally keeps mean / rstd
template <int COLS, bool VEC>
__device__ __forceinline__ void layer_norm_rows(float* xs, int ld, int D, int rows, const float* g, const float* b,
                                                float eps, float* mu_out, float* rs_out) {
    const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
    float* xr = xs + (size_t)r * ld + sub * COLS;
    const int c0 = sub * COLS;
    float xv[COLS];
    float s = 0.f;
    if (VEC && COLS % 4 == 0) {                                  // 8 * COLS == D: every lane's columns are real
#pragma unroll
        for (int k = 0; k < COLS; k += 4) {
            const float4 t = *reinterpret_cast<const float4*>(xr + k);
            xv[k] = t.x; xv[k + 1] = t.y; xv[k + 2] = t.z; xv[k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < COLS; ++k) xv[k] = c0 + k < D ? xr[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < COLS; ++k) s += (c0 + k < D) ? xv[k] : 0.f;
    const float mean = rat_group_sum<8>(s) / (float)D;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        const float t = (c0 + k < D) ? xv[k] - mean : 0.f;
        v += t * t;
    }
    const float rstd = 1.0f / sqrtf(rat_group_sum<8>(v) / (float)D + eps);
    if (r < rows) {
#pragma unroll
        for (int k = 0; k < COLS; ++k)
            if (c0 + k < D) xr[k] = (xv[k] - mean) * rstd * g[c0 + k] + b[c0 + k];
    }
    if (mu_out != nullptr && sub == 0) {
        mu_out[r] = mean;
        rs_out[r] = rstd;
    }
}

// tile[rows][0:width] (+ residual rows of `res`, token-indexed) -> dst rows, 16-byte coalesced when vec
// dst = mul * tile + res (res may be nullptr, or dst itself: every element is read and written by the same thread)
__device__ __forceinline__ void store_rows_residual(float* dst, const float* tile, int ld, const float* res,
                                                    const int64_t* rowtok, int rows, int width, bool vec, float mul,
                                                    const RatDrop* drop = nullptr) {
    if (drop != nullptr && drop->threshold != 0) {           // y = mul * Dropout(tile) + res, element-wise mask
        for (int e = threadIdx.x; e < rows * width; e += ATT_THREADS) {
            const int r = e / width, c = e - r * width;
            const int64_t tok = rowtok[r];
            const float v = drop->apply(tile[(size_t)r * ld + c], tok * width + c) * mul;
            dst[tok * width + c] = res != nullptr ? v + res[tok * width + c] : v;
        }
        return;
    }
    if (vec) {
        const int w4 = width >> 2;
        for (int e = threadIdx.x; e < rows * w4; e += ATT_THREADS) {
            const int r = e / w4, c4 = e - r * w4;
            const int64_t tok = rowtok[r];
            float4 v = *reinterpret_cast<const float4*>(tile + (size_t)r * ld + 4 * c4);
            if (mul != 1.0f) { v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul; }
            if (res != nullptr) {
                const float4 x = *reinterpret_cast<const float4*>(res + tok * width + 4 * c4);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            *reinterpret_cast<float4*>(dst + tok * width + 4 * c4) = v;
        }
    } else {
        for (int e = threadIdx.x; e < rows * width; e += ATT_THREADS) {
            const int r = e / width, c = e - r * width;
            const int64_t tok = rowtok[r];
            const float v = tile[(size_t)r * ld + c] * mul;
            dst[tok * width + c] = res != nullptr ? v + res[tok * width + c] : v;
        }
    }
}

// per-head vectors of the attention core: compile-time dim_head (8-byte LDS accesses) or runtime <= DH_MAX
template <int TDH>
struct HeadVec {
    static constexpr int N = TDH > 0 ? TDH : DH_MAX;
    float v[N];
    __device__ __forceinline__ void load(const float* p, int dh) {
        if (TDH > 0 && TDH % 2 == 0) {
#pragma unroll
            for (int c = 0; c < N; c += 2) {
                const float2 t = *reinterpret_cast<const float2*>(p + c);
                v[c] = t.x;
                v[c + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < N; ++c) v[c] = (TDH > 0 || c < dh) ? p[c] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* p, int dh, float scale) const {
        if (TDH > 0 && TDH % 2 == 0) {
#pragma unroll
            for (int c = 0; c < N; c += 2) *reinterpret_cast<float2*>(p + c) = make_float2(v[c] * scale, v[c + 1] * scale);
        } else {
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (TDH > 0 || c < dh) p[c] = v[c] * scale;
        }
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int c = 0; c < N; ++c) v[c] = 0.f;
    }
    __device__ __forceinline__ float dot(const HeadVec& o) const {       // padded lanes are 0 on both sides
#if !defined(RAT_EMU)
        if (N % 2 == 0) {                                                // two partial sums -> v_pk_fma_f32 (half the VALU issue slots)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 acc = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < N; c += 2) {
                const f32x2 x = {v[c], v[c + 1]}, y = {o.v[c], o.v[c + 1]};
                acc = __builtin_elementwise_fma(x, y, acc);
            }
            return acc.x + acc.y;
        }
#endif
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = 0; c + 1 < N; c += 2) {
            s0 = fmaf(v[c], o.v[c], s0);
            s1 = fmaf(v[c + 1], o.v[c + 1], s1);
        }
        if (N % 2) s0 = fmaf(v[N - 1], o.v[N - 1], s0);
        return s0 + s1;
    }
    __device__ __forceinline__ void axpy(float a, const HeadVec& x) {    // v += a * x
#pragma unroll
        for (int c = 0; c < N; ++c) v[c] = fmaf(a, x.v[c], v[c]);
    }
    __device__ __forceinline__ void scale_axpy(float s, float a, const HeadVec& x) {   // v = v * s + a * x
#pragma unroll
        for (int c = 0; c < N; ++c) v[c] = fmaf(a, x.v[c], v[c] * s);
    }
};

// Width-20 heads (RAT_m3: heads/2 heads of 2*dim_head) on the fast geometry run as two 10-wide HALVES on adjacent lanes: the
// per-(sequence, head, row) task count doubles back to what the 8 x 10 shape has (504 / 440 of 512 lanes instead of 252 /
// 220), each lane keeps only its half of q / o / dq / dk / dv, and the one thing a score needs from the partner lane — the
// other half of a dot product — is a single cross-lane add.
template <bool PAIRED, class V>
__device__ __forceinline__ float head_dot(const V& a, const V& b) {
    float s = a.dot(b);
    if (PAIRED) s += __shfl_xor(s, 1, 64);
    return s;
}

// LDS copies of the two weight matrices for the compile-time-geometry instantiations (d <= 16): wq_s [Q16][20] row = Q|K|V column n, the d
// weights of that column (rows >= 3 I and columns >= d zero), wo_s [16][I16 + 4] row = output feature (rows >= d zero).  As RatLdsRows they
// are the B operand of the forward projections, as RatLdsCols of the backward's dO = dy W_out and d(LN out) = dQKV W_qkv.
constexpr int CT_LDW = 20;
__host__ __device__ inline size_t ct_weight_floats(const AttnGeom& g) { return (size_t)g.Q16 * CT_LDW + (size_t)16 * (g.I16 + 4); }
__device__ __forceinline__ void ct_stage_weights(const AttnArgs& a, const AttnGeom& g, float* wq_s, float* wo_s) {
    const int total = (int)ct_weight_floats(g);
    for (int e = threadIdx.x; e < total; e += ATT_THREADS) wq_s[e] = 0.f;     // (wo_s follows wq_s)
    __syncthreads();
    for (int e = threadIdx.x; e < g.Q3 * g.D; e += ATT_THREADS) wq_s[(e / g.D) * CT_LDW + e % g.D] = a.w_qkv[e];
    if (a.w_out != nullptr)
        for (int e = threadIdx.x; e < g.D * g.I; e += ATT_THREADS) wo_s[(e / g.I) * (g.I16 + 4) + e % g.I] = a.w_out[e];
}

// ------------------------------------------------------------------------------------------------ forward
// EX = false: the plain PreNorm(Attention)(x) + x layer (residual = x, output scale 1) — the _ex constants fold away
// GC: columns per lane of the 8-lanes-per-row phases in the generic (TD = 0) kernels — 16 serves embedding_dim <= 128; the instantiations
// with GC = 2 serve embedding_dim <= 16 (the shipped MovieLens / Tmall geometries, d = 10) with an eighth of the per-lane column state
// GH: head count of a generic instantiation as a compile-time constant (0 = run-time): every LDS stride of the head side folds
// GD: likewise the embedding dimension of a generic instantiation (0 = run-time; unlike TD it promises no 16-byte alignment)
template <int TD, int TDH, bool EX = true, int GC = 16, int GH = 0, int GD = 0>
__global__ void __launch_bounds__(ATT_THREADS) attn_fwd_kernel(AttnArgs a) {
    constexpr bool FAST = TD > 0;
    constexpr int COLS = FAST ? (TD + 7) / 8 : GC;
    RAT_DYN_SMEM(smem);
    const int heads_c = FAST ? fast_heads(TDH) : (GH > 0 ? GH : a.heads);
    const AttnGeom g(FAST ? TD : (GD > 0 ? GD : a.d), heads_c, TDH > 0 ? TDH : a.dh);
    float* xs = reinterpret_cast<float*>(smem);
    float* qkv = xs + (size_t)ATT_ROWS * g.ldx;
    int64_t* rowtok = reinterpret_cast<int64_t*>(qkv + (size_t)ATT_ROWS * g.ldq);
    const int L = a.L, D = g.D, I = g.I, dh = TDH > 0 ? TDH : a.dh;
    const int ldx = g.ldx, ldq = g.ldq;

    // Compile-time geometry (GH, GD: the shipped d = 10 shapes, BASELINE configs[0]): both weight matrices are copied to LDS once per
    // work-group (ct_stage_weights) and — two heads — every 16 x 16 tile of a projection is a wave's task of its own.  At these sizes a GEMM
    // phase is a handful of MFMAs: fetching its weight fragments from L2 per tile and leaving half the waves without a task WAS the phase
    // (same-box A/B: profiles/round5/r5_small_d_ab.txt).
    constexpr bool CTW = !FAST && GH > 0 && GD > 0 && TDH > 0;
    constexpr int MTQ = (CTW && GH <= 2) ? 1 : ATT_MT, MTO = (CTW && GH <= 2) ? 1 : 2;
    float* const wq_s = reinterpret_cast<float*>(rowtok + 2 * ATT_ROWS);      // CTW: [Q16][20] to_qkv.weight, [16][I16 + 4] to_out.weight
    float* const wo_s = wq_s + (size_t)g.Q16 * CT_LDW;
    if (CTW) ct_stage_weights(a, g, wq_s, wo_s);
    zero_cols(xs, ldx, D);
    zero_cols(qkv, ldq, g.Q3);
    __syncthreads();
    RAT_PROF_DECL

    int64_t* const rowtok0 = rowtok;                          // double-buffered row maps (chunk c+1's map is written during chunk c),
    //                                                           addressed as base + parity * 64 so that they stay provably-LDS pointers
    //                                                           (an array of two pointers indexed by parity compiled to FLAT loads)
    {
        int nsq0, rows0;
        map_rows(a, blockIdx.x, rowtok0, nsq0, rows0);
    }
    __syncthreads();
    int parity = 0;
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x, parity ^= 1) {
        rowtok = rowtok0 + parity * ATT_ROWS;
        int nsq, rows;
        {
            const int64_t q0 = chunk * a.nsq_chunk;
            const int64_t left = a.nseq - q0;
            nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
            rows = nsq * a.L;
        }
        if (FAST) {
            RowFetch<FAST ? TD : 4> fx;
            fx.issue(a.x, rowtok);
            fx.stash(xs, ldx);
        } else if (GD > 0 && GH > 0 && GD % 2 == 0 && TDH > 0 && aligned8_dev(a.x)) {
            SmallFetch<(GD > 0 ? GD : 2), (GH > 0 && TDH > 0 ? GH * TDH : 2), (GH > 0 ? GH : 1)> sf;
            sf.issue(a.x, nullptr, nullptr, nullptr, rowtok);
            sf.stash(xs, nullptr, nullptr, nullptr, ldx, 0, 1.0f);
        } else {
            load_rows(xs, ldx, a.x, rowtok, D, a.vec_x != 0);
        }
        __syncthreads();
        layer_norm_rows<COLS, FAST && (8 * COLS == TD)>(xs, ldx, D, rows, a.ln_g, a.ln_b, a.eps, nullptr, nullptr);
        if (chunk + gridDim.x < a.nchunks) {
            int nsq1, rows1;
            map_rows(a, chunk + gridDim.x, (rowtok0 + (parity ^ 1) * ATT_ROWS), nsq1, rows1);
        }
        __syncthreads();
        const int mt_valid = (rows + 15) / 16;
        RAT_PROF_MARK(0);

        // Q|K|V = LN(x) W_qkv^T  -> qkv[rows][0:3I]
        {
            const RatLdsRows A{xs, ldx};
            auto epi = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
                if (FAST || col < g.Q3)
#pragma unroll
                    for (int r = 0; r < 4; ++r) qkv[(size_t)rat_acc_row(mt, r) * ldq + col] = acc[r];
            };
            if (CTW) rat_gemm_phase<false, MTQ, ATT_WAVES, ATT_MT, 0>(A, RatLdsRows{wq_s, CT_LDW}, mt_valid, g.Q16 / 16, g.D16 / 16, epi);
            else rat_gemm_phase<FAST, ATT_MT, ATT_WAVES, ATT_MT, (FAST ? TD / 16 : 0)>(A, RatGlobalWnkT<!FAST>{a.w_qkv, g.Q3, D, D, a.vec_wqkv != 0}, mt_valid, g.Q16 / 16, g.D16 / 16, epi);
        }
        __syncthreads();
        RAT_PROF_MARK(1);

        // softmax(Q K^T * scale) V, one lane per (sequence, head, query); result replaces Q in place
        float pf = 0.f;
        {
            const int nl = (D * 4 + 127) / 128;
            if ((int)threadIdx.x < ATT_ROWS * nl && chunk + gridDim.x < a.nchunks)
                pf = prefetch_lines_map((rowtok0 + (parity ^ 1) * ATT_ROWS), threadIdx.x, nl, a.x, D);
        }
        constexpr bool PAIRED = FAST && TDH == 20;
        constexpr int VW = PAIRED ? 10 : TDH;                  // per-lane vector width
        typedef HeadVec<VW> HV;
        const int ntasks = nsq * heads_c * L * (PAIRED ? 2 : 1);
        const float sl2 = a.scale * RAT_LOG2E;
        // (A 4x4x1-MFMA formulation of this core — one block per (sequence, head, 4 queries) — and a 16x16x4 one were built
        //  and measured slower than this VALU loop at L = 11 / 21: tools/experiments/README.md.)
        // PAIRED: every lane of every wave runs every trip (idle lanes shadow the last task and store nothing) so that the
        // cross-lane add in head_dot always sees a complete wave
        for (int task0 = threadIdx.x; PAIRED ? task0 - (int)threadIdx.x < ntasks : task0 < ntasks; task0 += ATT_THREADS) {
            const bool live = !PAIRED || task0 < ntasks;
            const int task = live ? task0 : ntasks - 1;
            const int half = PAIRED ? (task & 1) : 0, t2 = PAIRED ? task >> 1 : task;
            const int i = t2 % L;
            const int h = (t2 / L) % heads_c;
            const int sq = t2 / (L * heads_c);
            const int row_i = sq * L + i;
            float* qp = qkv + (size_t)row_i * ldq + h * dh + half * VW;
            HV q, o, kv;
            q.load(qp, dh);
            o.zero();
            float m = -INFINITY, l = 0.f;
            const float* kbase = qkv + (size_t)(sq * L) * ldq + I + h * dh + half * VW;
            // keys three at a time: all six K / V rows are requested before anything waits (one exposed LDS latency per three
            // keys instead of two per key) and the three score dot products are independent chains; the online-softmax
            // recurrence itself runs in the original key order, so the result is bit-identical to the one-key loop
            int j = 0;
            for (; j + CORE_UNROLL <= L; j += CORE_UNROLL) {
                HV kk[CORE_UNROLL], vv[CORE_UNROLL];
#pragma unroll
                for (int u = 0; u < CORE_UNROLL; ++u) {
                    const float* kp = kbase + (size_t)(j + u) * ldq;
                    kk[u].load(kp, dh);
                    vv[u].load(kp + I, dh);
                }
                float sc[CORE_UNROLL];
#pragma unroll
                for (int u = 0; u < CORE_UNROLL; ++u) sc[u] = head_dot<PAIRED>(q, kk[u]) * sl2;   // scores in log2 units
#pragma unroll
                for (int u = 0; u < CORE_UNROLL; ++u) {
                    const float mn = fmaxf(m, sc[u]);
                    const float corr = rat_exp2(m - mn);
                    const float p = rat_exp2(sc[u] - mn);
                    l = l * corr + p;
                    o.scale_axpy(corr, p, vv[u]);
                    m = mn;
                }
            }
            for (; j < L; ++j) {
                const float* kp = kbase + (size_t)j * ldq;
                kv.load(kp, dh);
                const float s = head_dot<PAIRED>(q, kv) * sl2;
                const float mn = fmaxf(m, s);
                const float corr = rat_exp2(m - mn);
                const float p = rat_exp2(s - mn);
                l = l * corr + p;
                kv.load(kp + I, dh);
                o.scale_axpy(corr, p, kv);
                m = mn;
            }
            const float inv = 1.0f / l;
            if (live) {
                o.store(qp, dh, inv);
                const int64_t tok = rowtok[row_i];
                if (a.o_save != nullptr) o.store(a.o_save + tok * I + h * dh + half * VW, dh, inv);
                if (a.lse_save != nullptr && half == 0) a.lse_save[tok * heads_c + h] = m + rat_log2(l);   // log2-domain log-sum-exp
            }
        }
        __syncthreads();
        RAT_PROF_MARK(2);
        // y = O W_out^T + b_out + x   (or y = O + x when Attention has no output projection).  The projection tile is
        // staged in xs (free since the QKV projection) so that the residual add and the store are whole-row, 16-byte accesses.
        if (a.w_out != nullptr) {
            const RatLdsRows A{qkv, ldq};
            auto epi = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
                if (FAST || col < D) {
                    const float bias = a.b_out[col];
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[(size_t)rat_acc_row(mt, r) * ldx + col] = acc[r] + bias;
                }
            };
            if (CTW) rat_gemm_phase<false, MTO, ATT_WAVES, ATT_MT, 0>(A, RatLdsRows{wo_s, g.I16 + 4}, mt_valid, g.D16 / 16, g.I16 / 16, epi);
            else rat_gemm_phase<FAST, 2, ATT_WAVES, ATT_MT, 0>(A, RatGlobalWnkT<!FAST>{a.w_out, D, I, I, a.vec_wout != 0}, mt_valid, g.D16 / 16, g.I16 / 16, epi);
            __syncthreads();
            store_rows_residual(a.y, xs, ldx, EX ? a.res : a.x, rowtok, rows, D, FAST || a.vec_x != 0, EX ? a.out_scale : 1.0f,
                                EX ? &a.drop : nullptr);
        } else {
            store_rows_residual(a.y, qkv, ldq, EX ? a.res : a.x, rowtok, rows, D, FAST || a.vec_x != 0, EX ? a.out_scale : 1.0f);
        }
        __syncthreads();
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));                              // keep the prefetch load alive (its value is irrelevant)
#endif
        RAT_PROF_MARK(3);
    }
    RAT_PROF_FLUSH(a.prof, 0);
}

// ----------------------------------------------------------------------------------------------- backward
template <int TD, int TDH, bool EX = true, int GC = 16, int GH = 0, int GD = 0>
__global__ void __launch_bounds__(ATT_THREADS) attn_bwd_kernel(AttnArgs a) {
    constexpr bool FAST = TD > 0;
    constexpr int COLS = FAST ? (TD + 7) / 8 : GC;
    RAT_DYN_SMEM(smem);
    const AttnGeom g(FAST ? TD : (GD > 0 ? GD : a.d), FAST ? fast_heads(TDH) : (GH > 0 ? GH : a.heads), TDH > 0 ? TDH : a.dh);
    const int L = a.L, D = g.D, I = g.I, dh = TDH > 0 ? TDH : a.dh, H = FAST ? fast_heads(TDH) : (GH > 0 ? GH : a.heads);
    const int ldx = g.ldx, ldq = g.ldq, ldt = g.ldt;
    float* xs = reinterpret_cast<float*>(smem);                 // [64][ldx]  LayerNorm(x)
    float* dys = xs + (size_t)ATT_ROWS * ldx;                   // [64][ldx]  dL/dy
    float* qkv = dys + (size_t)ATT_ROWS * ldx;                  // [64][ldq]  Q|K|V, later dQ|dK|dV
    float* ob = qkv + (size_t)ATT_ROWS * ldq;                   // [64][ldt]  O, later dQ
    float* dob = ob + (size_t)ATT_ROWS * ldt;                   // [64][ldt]  dO, later d(LayerNorm out)
    float* mu = dob + (size_t)ATT_ROWS * ldt;                   // [64]
    float* rs = mu + ATT_ROWS;                                  // [64]
    float* lses = rs + ATT_ROWS;                                // [64][H]
    float* dlt = lses + (size_t)ATT_ROWS * H;                   // [64][H]   rowsum(dO * O)
    int64_t* rowtok = reinterpret_cast<int64_t*>(dlt + (size_t)ATT_ROWS * H);
    const bool has_out = a.w_out != nullptr;
    // compile-time d <= 16 (one column tile): the d(LayerNorm out) GEMM splits its CONTRACTION over the waves (phase 5)
    constexpr bool KSPLIT = !FAST && GD > 0 && GD <= 16 && GH > 0 && ((GH * (TDH > 0 ? TDH : 16) + 15) / 16 * 16) >= 32;

    // persistent parameter-gradient accumulators
    f32x4 accq[QSLOTS], acco[OSLOTS];
#pragma unroll
    for (int s = 0; s < QSLOTS; ++s) accq[s] = rat_zero4();
#pragma unroll
    for (int s = 0; s < OSLOTS; ++s) acco[s] = rat_zero4();
    float dgam[COLS], dbet[COLS];
#pragma unroll
    for (int k = 0; k < COLS; ++k) dgam[k] = dbet[k] = 0.f;
    float dbo = 0.f;
    const int q_tn = g.D16 / 16, q_tiles = (g.Q16 / 16) * q_tn;       // dW_qkv tiles: (3I16/16) x (D16/16)
    const int o_tn = g.D16 / 16, o_tiles = (g.I16 / 16) * o_tn;       // dW_out^T tiles: (I16/16) x (D16/16)

    constexpr bool CTW = !FAST && GH > 0 && GD > 0 && TDH > 0;   // weights in LDS, two heads: one GEMM task per tile (see attn_fwd_kernel)
    constexpr int MTQ = (CTW && GH <= 2) ? 1 : ATT_MT, MTO = (CTW && GH <= 2) ? 1 : 2;
    float* const wq_s = reinterpret_cast<float*>(rowtok + 2 * ATT_ROWS);
    float* const wo_s = wq_s + (size_t)g.Q16 * CT_LDW;
    if (CTW) ct_stage_weights(a, g, wq_s, wo_s);
    zero_cols(xs, ldx, D);
    zero_cols(dys, ldx, D);
    zero_cols(qkv, ldq, g.Q3);
    zero_cols(ob, ldt, 0);
    zero_cols(dob, ldt, 0);
    __syncthreads();
    RAT_PROF_DECL

    int64_t* const rowtok0 = rowtok;                          // row maps are double-buffered (chunk c+1's map is written during chunk c);
    //                                                           base + parity * 64 keeps them provably-LDS pointers (see forward)
    {
        int nsq0, rows0;
        map_rows(a, blockIdx.x, rowtok0, nsq0, rows0);
    }
    __syncthreads();
    // LayerNorm backward multiplies this lane's COLS columns by gamma in every chunk: the values are fetched ONCE here.  (Loaded
    // inside the phase they compiled into COLS predicated single-dword loads, each in its own basic block behind a branch — a serial
    // chain of L2 round trips per chunk.)
    float lng[COLS];
    {
        const int c0g = (threadIdx.x & 7) * COLS;
#pragma unroll
        for (int k = 0; k < COLS; ++k) lng[k] = (FAST || c0g + k < D) ? a.ln_g[c0g + k] : 0.f;
    }
    int parity = 0;
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x, parity ^= 1) {
        rowtok = rowtok0 + parity * ATT_ROWS;
        int nsq, rows;
        {
            const int64_t q0 = chunk * a.nsq_chunk;
            const int64_t left = a.nseq - q0;
            nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
            rows = nsq * a.L;
        }
        RAT_PROF_MARK(0);
        if (FAST) {                                              // x, dy and O rows of the chunk in ONE memory round trip
            RowFetch<FAST ? TD : 4> fx, fdy;
            RowFetch<FAST ? FAST_INNER : 4> fo;
            fx.issue(a.x, rowtok);
            fdy.issue(a.dy, rowtok);
            fo.issue(a.o_save, rowtok);
            fx.stash(xs, ldx);
            fdy.stash(dys, ldx, EX ? a.out_scale : 1.0f);
            fo.stash(ob, ldt);
            if (EX && a.drop.threshold != 0) {                   // rare: re-stage dy through the Dropout mask (after the plain stash)
                __syncthreads();
                load_rows(dys, ldx, a.dy, rowtok, D, false, a.out_scale, &a.drop);
            }
        } else if (GD > 0 && GH > 0 && GD % 2 == 0 && TDH > 0 && (!EX || a.drop.threshold == 0) && aligned8_dev(a.x) && aligned8_dev(a.dy) &&
                   aligned8_dev(a.o_save)) {
            SmallFetch<(GD > 0 ? GD : 2), (GH > 0 && TDH > 0 ? GH * TDH : 2), (GH > 0 ? GH : 1)> sf;
            sf.issue(a.x, a.dy, a.o_save, a.lse_save, rowtok);
            sf.stash(xs, dys, ob, lses, ldx, ldt, EX ? a.out_scale : 1.0f);
        } else {
            load_rows(xs, ldx, a.x, rowtok, D, a.vec_x != 0);
            load_rows(dys, ldx, a.dy, rowtok, D, a.vec_x != 0, EX ? a.out_scale : 1.0f, EX ? &a.drop : nullptr);
            load_rows(ob, ldt, a.o_save, rowtok, I, (I % 4) == 0 && a.vec_x != 0);
        }
        if (FAST || !(GD > 0 && GH > 0 && GD % 2 == 0 && TDH > 0 && (!EX || a.drop.threshold == 0) && aligned8_dev(a.x) && aligned8_dev(a.dy) &&
                      aligned8_dev(a.o_save)))
        for (int e = threadIdx.x; e < ATT_ROWS * H; e += ATT_THREADS) {
            const int64_t tok = rowtok[e / H];
            lses[e] = tok >= 0 ? a.lse_save[tok * H + e % H] : 0.f;
        }
        __syncthreads();
        RAT_PROF_MARK(1);
        layer_norm_rows<COLS, FAST && (8 * COLS == TD)>(xs, ldx, D, rows, a.ln_g, a.ln_b, a.eps, mu, rs);
        if (chunk + gridDim.x < a.nchunks) {
            int nsq1, rows1;
            map_rows(a, chunk + gridDim.x, (rowtok0 + (parity ^ 1) * ATT_ROWS), nsq1, rows1);
        }
        __syncthreads();
        const int mt_valid = (rows + 15) / 16;
        RAT_PROF_MARK(2);

        // (1) recompute Q|K|V
        {
            const RatLdsRows A{xs, ldx};
            auto epi = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
                if (FAST || col < g.Q3)
#pragma unroll
                    for (int r = 0; r < 4; ++r) qkv[(size_t)rat_acc_row(mt, r) * ldq + col] = acc[r];
            };
            if (CTW) rat_gemm_phase<false, MTQ, ATT_WAVES, ATT_MT, 0>(A, RatLdsRows{wq_s, CT_LDW}, mt_valid, g.Q16 / 16, g.D16 / 16, epi);
            else rat_gemm_phase<FAST, ATT_MT, ATT_WAVES, ATT_MT, (FAST ? TD / 16 : 0)>(A, RatGlobalWnkT<!FAST>{a.w_qkv, g.Q3, D, D, a.vec_wqkv != 0}, mt_valid, g.Q16 / 16, g.D16 / 16, epi);
        }
        RAT_PROF_MARK(3);
        if (has_out) {
            // (2) dO = dy W_out  (dob[rows][0:I])
            const RatLdsRows A{dys, ldx};
            auto epi = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
                if (FAST || col < I)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dob[(size_t)rat_acc_row(mt, r) * ldt + col] = acc[r];
            };
            if (CTW) rat_gemm_phase<false, MTO, ATT_WAVES, ATT_MT, 0>(A, RatLdsCols{wo_s, g.I16 + 4}, mt_valid, g.I16 / 16, g.D16 / 16, epi);
            else rat_gemm_phase<FAST, 2, ATT_WAVES, ATT_MT, (FAST ? TD / 16 : 0)>(A, RatGlobalWknT<!FAST>{a.w_out, D, I, I}, mt_valid, g.I16 / 16, g.D16 / 16, epi);
            RAT_PROF_MARK(4);
            // (3) dW_out += dy^T O ; db_out += colsum(dy)
            const RatLdsCols At{ob, ldt};                          // transposed tile grid (I16/16 x D16/16): dW_out^T = O^T dy,
            const RatLdsCols Bt{dys, ldx};                         // so that the column-tile count (D/16) divides the wave count
            if (FAST && mt_valid == ATT_MT) rat_wave_gemm_ct<OSLOTS, ATT_WAVES, (FAST ? 5 * (TD / 16) : 8), (FAST ? TD / 16 : 1), ATT_MT>(acco, At, Bt);
            else if (FAST) rat_wave_gemm_ct<OSLOTS, ATT_WAVES, (FAST ? 5 * (TD / 16) : 8), (FAST ? TD / 16 : 1), 0>(acco, At, Bt, mt_valid);
            else rat_wave_gemm_slots<OSLOTS, ATT_WAVES, 0>(acco, At, Bt, o_tiles, o_tn, mt_valid);
            {   // db_out partials: thread = (column, row group); the row groups are combined once, after the chunk loop
                const int nrg = ATT_THREADS / D, col = threadIdx.x % D, rg = threadIdx.x / D;
                if (rg < nrg)
                    for (int r = rg; r < rows; r += nrg) dbo += dys[(size_t)r * ldx + col];
            }
        } else {
            for (int e = threadIdx.x; e < ATT_ROWS * D; e += ATT_THREADS) {
                const int r = e / D, c = e - r * D;
                dob[(size_t)r * ldt + c] = dys[(size_t)r * ldx + c];
            }
        }
        __syncthreads();
        RAT_PROF_MARK(5);

        // (4) attention backward, pass 1: one lane per query row -> delta, dQ (written over O)
        constexpr bool PAIRED = FAST && TDH == 20;
        constexpr int VW = PAIRED ? 10 : TDH;
        typedef HeadVec<VW> HV;
        const int ntasks = nsq * H * L * (PAIRED ? 2 : 1);
        const float sl2 = a.scale * RAT_LOG2E;
        float pf = 0.f;
        {   // passes 1 and 2 touch LDS only: the next chunk's x / dy / O / lse lines travel HBM -> L2 meanwhile
            const int nlx = (D * 4 + 127) / 128, nlo = (I * 4 + 127) / 128;
            int t = threadIdx.x;
            if (chunk + gridDim.x < a.nchunks) {
                const int64_t* nrt = (rowtok0 + (parity ^ 1) * ATT_ROWS);          // written before the barrier that opened this phase
                if (t < ATT_ROWS * nlx) pf = prefetch_lines_map(nrt, t, nlx, a.x, D);
                else if ((t -= ATT_ROWS * nlx) < ATT_ROWS * nlx) pf = prefetch_lines_map(nrt, t, nlx, a.dy, D);
                else if ((t -= ATT_ROWS * nlx) < ATT_ROWS * nlo) pf = prefetch_lines_map(nrt, t, nlo, a.o_save, I);
                else if ((t -= ATT_ROWS * nlo) < ATT_ROWS) pf = prefetch_lines_map(nrt, t, 1, a.lse_save, H);
            }
        }
        for (int task0 = threadIdx.x; PAIRED ? task0 - (int)threadIdx.x < ntasks : task0 < ntasks; task0 += ATT_THREADS) {
            const bool live = !PAIRED || task0 < ntasks;           // PAIRED: idle lanes shadow the last task (see forward)
            const int task = live ? task0 : ntasks - 1;
            const int half = PAIRED ? (task & 1) : 0, t2 = PAIRED ? task >> 1 : task;
            const int i = t2 % L;
            const int h = (t2 / L) % H;
            const int sq = t2 / (L * H);
            const int row_i = sq * L + i;
            const int ho = h * dh + half * VW;
            float* op = ob + (size_t)row_i * ldt + ho;
            HV q, go, dq, kv;
            q.load(qkv + (size_t)row_i * ldq + ho, dh);
            go.load(dob + (size_t)row_i * ldt + ho, dh);
            kv.load(op, dh);
            const float delta = head_dot<PAIRED>(go, kv);
            dq.zero();
            if (live) dlt[row_i * H + h] = delta;                // (both halves write the same value)
            const float lse = lses[row_i * H + h];
            const float* kbase = qkv + (size_t)(sq * L) * ldq + I + ho;
            int j = 0;
            for (; j < L; ++j) {
                const float* kp = kbase + (size_t)j * ldq;
                kv.load(kp + I, dh);
                const float dp = head_dot<PAIRED>(go, kv);
                kv.load(kp, dh);
                const float p = rat_exp2(head_dot<PAIRED>(q, kv) * sl2 - lse);
                dq.axpy(p * (dp - delta), kv);
            }
            if (live) dq.store(op, dh, a.scale);
        }
        __syncthreads();
        RAT_PROF_MARK(6);
        // pass 2: one lane per key row -> dK, dV (written over K, V)
        for (int task0 = threadIdx.x; PAIRED ? task0 - (int)threadIdx.x < ntasks : task0 < ntasks; task0 += ATT_THREADS) {
            const bool live = !PAIRED || task0 < ntasks;
            const int task = live ? task0 : ntasks - 1;
            const int half = PAIRED ? (task & 1) : 0, t2 = PAIRED ? task >> 1 : task;
            const int j = t2 % L;
            const int h = (t2 / L) % H;
            const int sq = t2 / (L * H);
            const int ho = h * dh + half * VW;
            float* kp = qkv + (size_t)(sq * L + j) * ldq + I + ho;
            HV kk, vv, dk, dv, t;
            kk.load(kp, dh);
            vv.load(kp + I, dh);
            dk.zero();
            dv.zero();
            int i = 0;
            for (; i < L; ++i) {
                const int row_i = sq * L + i;
                t.load(dob + (size_t)row_i * ldt + ho, dh);
                const float dp = head_dot<PAIRED>(t, vv);
                const float lse = lses[row_i * H + h], delta = dlt[row_i * H + h];
                HV qv;
                qv.load(qkv + (size_t)row_i * ldq + ho, dh);
                const float p = rat_exp2(head_dot<PAIRED>(qv, kk) * sl2 - lse);
                dv.axpy(p, t);
                dk.axpy(p * (dp - delta), qv);
            }
            if (live) {
                dk.store(kp, dh, a.scale);
                dv.store(kp + I, dh, 1.0f);
            }
        }
        __syncthreads();
        // dQ (in ob) -> Q columns of qkv: qkv now holds d[Q|K|V]
        if (FAST) {
            const int w4 = I >> 2;
            for (int e = threadIdx.x; e < ATT_ROWS * w4; e += ATT_THREADS) {
                const int r = e / w4, c4 = e - r * w4;
                *reinterpret_cast<float4*>(qkv + (size_t)r * ldq + 4 * c4) = *reinterpret_cast<const float4*>(ob + (size_t)r * ldt + 4 * c4);
            }
        } else {
            for (int e = threadIdx.x; e < rows * I; e += ATT_THREADS) {
                const int r = e / I, c = e - r * I;
                qkv[(size_t)r * ldq + c] = ob[(size_t)r * ldt + c];
            }
        }
        __syncthreads();
        RAT_PROF_MARK(7);

        // (5) d(LN out) = dQKV W_qkv -> dob[rows][0:D] ; (6) dW_qkv += dQKV^T LN(x)
        {
            const RatLdsRows A{qkv, ldq};
            const RatGlobalWknT<!FAST> Bw{a.w_qkv, g.Q3, D, D};
            if (FAST && TD == 64) {
                // N = 64 gives only 4 column tiles for 8 waves: pair the waves on the K extent (3I = 15 k-blocks -> 8 + 7),
                // each with all 4 row tiles (16 MFMAs per k-block keep the 2-deep L2 prefetch of B ahead of the math),
                // and combine the two partial tiles through LDS.
                const int w = rat_wave(), nt = w & 3, half = w >> 2;
                const int kbt = g.Q16 / 16, mid = (kbt + 1) / 2;
                f32x4 acc[ATT_MT];
#pragma unroll
                for (int i = 0; i < ATT_MT; ++i) acc[i] = rat_zero4();
                rat_wave_gemm_col<ATT_MT, 0>(acc, A, Bw, 0, nt, half ? kbt : mid, half ? mid : 0);
                const int col = rat_acc_col(nt);
                float* part = half ? ob : dob;                  // ob is free since dQ moved into qkv; LN-bwd adds the two partials
#pragma unroll
                for (int i = 0; i < ATT_MT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[(size_t)rat_acc_row(i, r) * ldt + col] = acc[i][r];
            } else if (KSPLIT) {
                // ONE column tile: as (row-tile pair x column tile) tasks that is 2 tasks for 8 waves, each a chain over all 3 I / 16
                // k-blocks (stamps at the Tmall shape: 18 % of the kernel).  Four-way split of the contraction instead: wave = (row-tile
                // pair w & 1, K part w >> 1); the partial tiles land side by side in dob / ob (free since dQ moved into qkv), LayerNorm
                // backward adds them.  (Only with compile-time geometry: in the run-time-dimension kernel the extra live state spilled.)
                const int w = rat_wave(), mb = w & 1, part = w >> 1;
                const int kbt = g.Q16 / 16, k0 = part * kbt / 4, k1 = (part + 1) * kbt / 4;
                f32x4 acc[2] = {rat_zero4(), rat_zero4()};
                if (k1 > k0 && CTW) rat_wave_gemm_col<2, 0>(acc, A, RatLdsCols{wq_s, CT_LDW}, 2 * mb, 0, k1, k0);
                else if (k1 > k0) rat_wave_gemm_col<2, 0>(acc, A, Bw, 2 * mb, 0, k1, k0);
                float* pt = (part < 2 ? dob : ob) + 16 * (part & 1);
                const int col = rat_acc_col(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[(size_t)rat_acc_row(2 * mb + i, r) * ldt + col] = acc[i][r];
            } else {
                rat_gemm_phase<FAST, 2, ATT_WAVES, ATT_MT, 0>(A, Bw, mt_valid, g.D16 / 16, g.Q16 / 16, [&](int mt, int nt, const f32x4& acc) {
                    const int col = rat_acc_col(nt);
                    if (FAST || col < D)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dob[(size_t)rat_acc_row(mt, r) * ldt + col] = acc[r];
                });
            }
            RAT_PROF_MARK(8);
            const RatLdsCols At{qkv, ldq};
            const RatLdsCols Bt{xs, ldx};
            if (FAST && mt_valid == ATT_MT) rat_wave_gemm_ct<QSLOTS, ATT_WAVES, (FAST ? 15 * (TD / 16) : 8), (FAST ? TD / 16 : 1), ATT_MT>(accq, At, Bt);
            else if (FAST) rat_wave_gemm_ct<QSLOTS, ATT_WAVES, (FAST ? 15 * (TD / 16) : 8), (FAST ? TD / 16 : 1), 0>(accq, At, Bt, mt_valid);
            else rat_wave_gemm_slots<QSLOTS, ATT_WAVES, 0>(accq, At, Bt, q_tiles, q_tn, mt_valid);
        }
        __syncthreads();
        RAT_PROF_MARK(9);

        // (7) LayerNorm backward + residual: dx = dy + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxn * gamma
        //     8 lanes per row, lane `sub` owns COLS contiguous columns -> 16-byte global / LDS accesses
        {
            const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
            const int c0 = sub * COLS;
            const bool valid = r < rows;
            const int64_t tok = valid ? rowtok[r] : 0;
            const float mean = mu[r], rstd = rs[r];
            float xh[COLS], gg[COLS], out[COLS], ad[COLS];
            float s1 = 0.f, s2 = 0.f;
            // the gradient added to the LayerNorm-backward result: the dy tile (PreNorm(Attention) + x), another tensor, or nothing
            const bool add_lds = EX ? a.add_lds != 0 : true;
            if (FAST && COLS % 4 == 0 && !add_lds) {           // a separate gradient tensor: 16-byte loads of this lane's columns
#pragma unroll
                for (int k = 0; k < COLS; k += 4) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid && a.add != nullptr) t = *reinterpret_cast<const float4*>(a.add + tok * D + c0 + k);
                    ad[k] = t.x; ad[k + 1] = t.y; ad[k + 2] = t.z; ad[k + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < COLS; ++k) {
                    const int c = c0 + k;
                    const bool in = (FAST || c < D) && valid;
                    if (add_lds) ad[k] = in ? dys[(size_t)r * ldx + c] : 0.f;
                    else ad[k] = (in && a.add != nullptr) ? a.add[tok * D + c] : 0.f;
                }
            }
            if (FAST && COLS % 4 == 0) {
#pragma unroll
                for (int k = 0; k < COLS; k += 4) {
                    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid) xv = *reinterpret_cast<const float4*>(a.x + tok * D + c0 + k);
                    float4 gv = *reinterpret_cast<const float4*>(dob + (size_t)r * ldt + c0 + k);
                    if (TD == 64) {                                 // second K-half of the split dXn GEMM
                        const float4 g2 = *reinterpret_cast<const float4*>(ob + (size_t)r * ldt + c0 + k);
                        gv.x += g2.x; gv.y += g2.y; gv.z += g2.z; gv.w += g2.w;
                    }
                    xh[k] = xv.x; xh[k + 1] = xv.y; xh[k + 2] = xv.z; xh[k + 3] = xv.w;
                    gg[k] = gv.x; gg[k + 1] = gv.y; gg[k + 2] = gv.z; gg[k + 3] = gv.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < COLS; ++k) {
                    const int c = c0 + k;
                    xh[k] = (c < D && valid) ? a.x[tok * D + c] : 0.f;
                    gg[k] = (c < D && valid) ? dob[(size_t)r * ldt + c] : 0.f;
                    if (KSPLIT && c < D && valid)                   // the other three K parts of the split dXn GEMM
                        gg[k] += dob[(size_t)r * ldt + 16 + c] + (ob[(size_t)r * ldt + c] + ob[(size_t)r * ldt + 16 + c]);
                }
            }
#pragma unroll
            for (int k = 0; k < COLS; ++k) {
                const int c = c0 + k;
                const bool on = (FAST || c < D) && valid;
                xh[k] = on ? (xh[k] - mean) * rstd : 0.f;
                gg[k] = on ? gg[k] : 0.f;
                const float gw = on ? gg[k] * lng[k] : 0.f;
                s1 += gw;
                s2 += gw * xh[k];
            }
            s1 = rat_group_sum<8>(s1) / (float)D;
            s2 = rat_group_sum<8>(s2) / (float)D;
#pragma unroll
            for (int k = 0; k < COLS; ++k) {
                const int c = c0 + k;
                const bool on = (FAST || c < D) && valid;
                const float gw = on ? gg[k] * lng[k] : 0.f;
                out[k] = on ? ad[k] + rstd * (gw - s1 - xh[k] * s2) : 0.f;
                dgam[k] += gg[k] * xh[k];
                dbet[k] += gg[k];
            }
            if (valid) {
                if (FAST && COLS % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < COLS; k += 4)
                        *reinterpret_cast<float4*>(a.y + tok * D + c0 + k) = make_float4(out[k], out[k + 1], out[k + 2], out[k + 3]);
                } else {
#pragma unroll
                    for (int k = 0; k < COLS; ++k)
                        if (c0 + k < D) a.y[tok * D + c0 + k] = out[k];
                }
            }
        }
        __syncthreads();
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));
#endif
        RAT_PROF_MARK(10);
    }
    RAT_PROF_FLUSH(a.prof, 12);

    // ---- write this work-group's parameter-gradient slab: [dW_qkv | dW_out | db_out | dgamma | dbeta]
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
    float* s_wqkv = slab;
    float* s_wout = s_wqkv + (int64_t)g.Q3 * D;
    float* s_bout = s_wout + (int64_t)D * I;
    float* s_gam = s_bout + D;
    float* s_bet = s_gam + D;
#pragma unroll
    for (int s = 0; s < QSLOTS; ++s) {
        const int id = rat_wave() + ATT_WAVES * s;
        if (id < q_tiles) {
            const int col = rat_acc_col(id % q_tn);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rat_acc_row(id / q_tn, r);
                if (row < g.Q3 && col < D) s_wqkv[(int64_t)row * D + col] = accq[s][r];
            }
        }
    }
    if (has_out) {
#pragma unroll
        for (int s = 0; s < OSLOTS; ++s) {
            const int id = rat_wave() + ATT_WAVES * s;
            if (id < o_tiles) {
                const int col = rat_acc_col(id % o_tn);           // output-feature index (row of W_out)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rat_acc_row(id / o_tn, r);     // inner index (column of W_out)
                    if (row < I && col < D) s_wout[(int64_t)col * I + row] = acco[s][r];
                }
            }
        }
        {
            __syncthreads();
            float* red0 = dys;                                   // free now: [nrg][D] partials
            const int nrg = ATT_THREADS / D, col = threadIdx.x % D, rg = threadIdx.x / D;
            if (rg < nrg) red0[rg * D + col] = dbo;
            __syncthreads();
            if (threadIdx.x < D) {
                float sacc = 0.f;
                for (int k = 0; k < nrg; ++k) sacc += red0[k * D + threadIdx.x];
                s_bout[threadIdx.x] = sacc;
            }
        }
    }
    // dgamma / dbeta: 64 row-slots x D partials -> LDS -> column sums
    float* red = xs;                                             // [64][ldx] is free now
    {
        const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int c = sub * COLS + k;
            if (c < D) red[(size_t)r * ldx + c] = dgam[k];
        }
        __syncthreads();
        if (threadIdx.x < D) {
            float sacc = 0.f;
            for (int rr = 0; rr < ATT_ROWS; ++rr) sacc += red[(size_t)rr * ldx + threadIdx.x];
            s_gam[threadIdx.x] = sacc;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int c = sub * COLS + k;
            if (c < D) red[(size_t)r * ldx + c] = dbet[k];
        }
        __syncthreads();
        if (threadIdx.x < D) {
            float sacc = 0.f;
            for (int rr = 0; rr < ATT_ROWS; ++rr) sacc += red[(size_t)rr * ldx + threadIdx.x];
            s_bet[threadIdx.x] = sacc;
        }
    }
}

#include "attn_wide.h"      // wide heads at a small embedding dimension, one launch per direction (exact fp32)
#include "attn_b3.h"        // the bf16x3 kernels of the north-star geometry

int check_dims(const RatSeqMap* map, int d, int heads, int dim_head, bool backward) {
    RAT_REQUIRE(map != nullptr, "null seq map");
    RAT_REQUIRE(d > 0 && heads > 0 && dim_head > 0, "bad dims");
    RAT_REQUIRE(dim_head <= DH_MAX || dim_head == 20, "dim_head > 16 (other than 20) is not supported by this kernel");
    RAT_REQUIRE(d <= 128, "embedding_dim > 128 is not supported by this kernel");
    RAT_REQUIRE(map->L >= 1 && map->L <= ATT_ROWS, "sequence length (K+1 or F+1) must be in [1, 64]");
    RAT_REQUIRE(map->nseq >= 1 && map->q_div >= 1, "bad seq map");
    const AttnGeom g(d, heads, dim_head);
    RAT_REQUIRE(g.fwd_smem() <= 160 * 1024, "heads*dim_head too large for the LDS tile (forward)");
    if (backward) {
        RAT_REQUIRE(g.bwd_smem(heads) <= 160 * 1024, "heads*dim_head too large for the LDS tile (backward)");
        RAT_REQUIRE((g.Q16 / 16) * (g.D16 / 16) <= QSLOTS * ATT_WAVES,
                    "3*heads*dim_head x embedding_dim exceeds the in-register dW_qkv accumulator budget");
        RAT_REQUIRE((g.D16 / 16) * (g.I16 / 16) <= OSLOTS * ATT_WAVES,
                    "embedding_dim x heads*dim_head exceeds the in-register dW_out accumulator budget");
    }
    return 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void fill_common(AttnArgs& a, const RatAttnParams* w, const RatSeqMap* map, int d, int heads, int dim_head,
                 float ln_eps) {
    a.ln_g = w->ln_g;
    a.ln_b = w->ln_b;
    a.w_qkv = w->w_qkv;
    a.w_out = w->w_out;
    a.b_out = w->b_out;
    a.nseq = map->nseq;
    a.q_div = map->q_div;
    a.hi_stride = map->hi_stride;
    a.lo_stride = map->lo_stride;
    a.pos_stride = map->pos_stride;
    a.L = map->L;
    a.nq = (map->queries > 0 && map->queries < map->L) ? map->queries : map->L;
    a.nsq_chunk = ATT_ROWS / map->L;
    a.nchunks = (map->nseq + a.nsq_chunk - 1) / a.nsq_chunk;
    a.d = d;
    a.heads = heads;
    a.dh = dim_head;
    a.eps = ln_eps;
    a.scale = 1.0f / sqrtf((float)dim_head);
    a.out_scale = 1.0f;
    const int I = heads * dim_head;
    a.vec_wqkv = (d % 4 == 0) && aligned16(w->w_qkv);
    a.vec_wout = (I % 4 == 0) && aligned16(w->w_out);
    a.prof = rat_prof_buffer();
}

// the compile-time dim_head instantiations move per-head vectors 8 bytes at a time, o_save rows included
bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

// which compiled fast shape (if any) serves these dimensions
int fast_dim(const AttnArgs& a, std::initializer_list<const void*> ptrs) {
    if (!((a.dh == 10 || a.dh == 20) && a.heads == fast_heads(a.dh)) || a.w_out == nullptr) return 0;
    for (const void* p : ptrs)
        if (p != nullptr && !aligned16(p)) return 0;
    if (!aligned16(a.w_qkv) || !aligned16(a.w_out)) return 0;
    return (a.d == 64 || a.d == 16) ? a.d : 0;
}

}  // namespace

extern "C" int rat_attn_fwd(const float* x, float* y, float* o_save, float* lse_save, const RatAttnParams* w_host,
                            const RatSeqMap* map_host, int d, int heads, int dim_head, float ln_eps, void* stream) {
    return rat_attn_fwd_ex(x, x, y, o_save, lse_save, w_host, map_host, d, heads, dim_head, 0.f, 1.f, ln_eps, 0.f, 0, RAT_ARITH_F32,
                           nullptr, 0, stream);
}

// bf16x3 kernels exist for the north-star geometry (embedding_dim 64, 8 heads x 10) and, inside the same 64-wide tiles, for
// embedding_dim 40 / 48 / 56 with the same heads (DPAD: the shipped KKBox config is d = 40); every other shape runs the exact-fp32
// kernels whatever `arith` says
static bool b3_dim(int d) { return d == B3_D || d == 40 || d == 48 || d == 56; }
// (round 6: also 4 heads x 20 at embedding_dim 64 — what RAT_m3 runs at the north-star config: the same projections and planes, NH = 4)
static bool b3_geom(int d, int heads, int dim_head) {
    return (b3_dim(d) && heads == B3_H && dim_head == B3_DH) || (d == B3_D && heads == 4 && dim_head == 2 * B3_DH);
}
static bool b3_shape(int d, int heads, int dim_head, const RatAttnParams* w) {
    return b3_geom(d, heads, dim_head) && w->w_out != nullptr;
}
// PH instantiation of attn_bwd3_kernel: P of a chunk inside the dy planes' 24 KB
static bool b3_ph_fits(int L, int nsq_chunk, int heads = B3_H) { return (size_t)nsq_chunk * heads * L * L * 4 <= (size_t)3 * B3_XP; }
// the matrix-pipe backward core (b3_bwd_core_mfma) by sequence length — see its comment for the measurements behind the rule; the
// attn_bwd_core_mfma knob forces it on (1, any L <= 32) or off (0)
static int b3_matrix_core(int L) {                     // -> 0 (VALU passes) | NIT (all key tiles of a query tile at once) | 13 (three tiles, key tiles inner)
    const int k = rat_knob(RAT_KNOB_ATTN_BWD_CORE_MFMA);
    if (L > 48 || k == 0) return 0;
    // 33 ... 48 tokens (round 6; one sequence per chunk): three tiles at once spill 64-66 VGPRs and lose to the VALU passes (1.24 against
    // 1.17 ms at L = 41); with the key tiles as the inner loop (b3_bwd_core_mfma_kt) the core wins by 4 % from 40 tokens on — r6_attn_L41_ab.txt.
    // At one and two tiles that form is SLOWER than the all-at-once form (L 31: 1.77 against 1.72 ms; L 21 / 11: x 1.2) and is not built in.
    if (L > 32) return (k == 1 || L >= 40) ? 13 : 0;
    return (k == 1 || L >= 28) ? (L > 16 ? 2 : 1) : 0;
}
// the matrix-pipe FORWARD core (b3_fwd_core_mfma): attn_fwd_core_mfma knob 0 = by length (28 ... 32 and 40 ... 48 tokens), 2 = forced on (L <= 48), 3 = off
// (1 selected round 3's bf16x3 core until round 6 — tools/experiments/attn_fwd3m_kernel.hip.txt — and now means 0)
static int b3_fwd_matrix_core(int L, int nq) {
    const int k = rat_knob(RAT_KNOB_ATTN_FWD_CORE_MFMA);
    if (L > 48 || nq < L || k == 3) return 0;
    // by length: 28 ... 32 tokens (two 16-row tiles, >= 88 % full) and — round 6 — 40 ... 48 (three tiles, one sequence per chunk: the VALU loop
    // leaves 36 % of the lanes idle there and walks 41 keys per lane; BASELINE configs[3]'s intra-sample sequences, F = 40)
    if (k != 2 && !((L >= 28 && L <= 32) || L >= 40)) return 0;
    return L > 32 ? 3 : (L > 16 ? 2 : 1);
}
static bool b3_ph_enabled() {                          // on unless the attn_bwd_ph knob is 0 (same-box A/B: L = 11 1.2477 -> 1.2322 ms, -1.2 %)
    return rat_knob(RAT_KNOB_ATTN_BWD_PH) != 0;
}
// `valid` of the split jobs whose N is the embedding dimension (rat_split_weights: n_valid)
static int b3_nvalid(int d) { return d == B3_D ? 0 : d; }
// ... and only while every token's byte offset in the widest array (o_save: 320 B per token) fits 32 bits (b3_ld4)
static bool b3_off32_ok(const RatSeqMap* m) {
    if (m->hi_stride < 0 || m->lo_stride < 0 || m->pos_stride < 0) return false;
    const int64_t qd = m->q_div, hi = (m->nseq - 1) / qd, lo = m->nseq - 1 < qd - 1 ? m->nseq - 1 : qd - 1;
    const double max_tok = (double)hi * (double)m->hi_stride + (double)lo * (double)m->lo_stride + (double)(m->L - 1) * (double)m->pos_stride;
    return (max_tok + 1.0) * (double)(B3_I * 4) < 4294967296.0;
}

extern "C" size_t rat_attn_fwd_workspace(int d, int heads, int dim_head) {
    return b3_geom(d, heads, dim_head) ? B3_W_QKV + B3_W_OUT : 0;
}

// RatAttnParams.planes: [W_qkv | W_out^T | W_qkv^T | W_out] fragment planes (the backward's three first, the forward's second one last)
extern "C" size_t rat_attn_planes_bytes(int d, int heads, int dim_head) {
    return b3_geom(d, heads, dim_head) ? B3_W_BYTES : 0;
}
extern "C" int rat_attn_split_jobs(const RatAttnParams* w_host, int d, int heads, int dim_head, void* planes, RatSplitJob* jobs_out) {
    RAT_REQUIRE(w_host && jobs_out, "null pointer");
    if (!b3_shape(d, heads, dim_head, w_host) || planes == nullptr) return 0;
    RAT_REQUIRE(aligned16(planes) && w_host->w_qkv, "planes must be 16-byte aligned");
    char* ws = static_cast<char*>(planes);
    // (d < 64: K = d is padded to 64 by the split itself; where N is the embedding dimension the planes cover 64 rows, d of them real)
    jobs_out[0] = RatSplitJob{w_host->w_qkv, ws, B3_Q3, d, d, 0, 0, 0};                                                        // forward + backward
    jobs_out[1] = RatSplitJob{w_host->w_out, ws + B3_W_QKV, B3_I, d, B3_I, 1, 0, 0};                                          // backward: dO
    jobs_out[2] = RatSplitJob{w_host->w_qkv, ws + B3_W_QKV + B3_W_OUTT, B3_D, B3_Q3, d, 1, 0, b3_nvalid(d)};                  // backward: d(LN out)
    jobs_out[3] = RatSplitJob{w_host->w_out, ws + B3_W_QKV + B3_W_OUTT + B3_W_QKVT, B3_D, B3_I, B3_I, 0, 0, b3_nvalid(d)};     // forward: out-proj
    return 4;
}

extern "C" int rat_attn_fwd_ex(const float* x, const float* res, float* y, float* o_save, float* lse_save,
                               const RatAttnParams* w_host, const RatSeqMap* map_host, int d, int heads, int dim_head,
                               float softmax_scale, float out_scale, float ln_eps, float dropout_p, uint64_t dropout_seed,
                               int arith, float* workspace, size_t workspace_bytes, void* stream) {
    if (check_dims(map_host, d, heads, dim_head, false)) return -1;
    RAT_REQUIRE(x && y && w_host && w_host->ln_g && w_host->ln_b && w_host->w_qkv, "null pointer");
    RAT_REQUIRE(w_host->w_out != nullptr || heads * dim_head == d, "missing w_out");
    AttnArgs a{};
    fill_common(a, w_host, map_host, d, heads, dim_head, ln_eps);
    if (softmax_scale > 0.f) a.scale = softmax_scale;
    a.out_scale = out_scale;
    a.res = res;
    a.x = x;
    a.y = y;
    a.o_save = o_save;
    a.lse_save = lse_save;
    a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(res);
    const AttnGeom g(d, heads, dim_head);
    // (+ the LDS copies of the weights for the instantiations with compile-time geometry: attn_fwd_kernel CTW)
    const bool ct_shape = dim_head == 10 && ((d == 10 && (heads == 8 || heads == 2)) || (d == 16 && heads == 2));
    const size_t smem = g.fwd_smem() + (ct_shape ? ct_weight_floats(g) * sizeof(float) : 0);
    const int per_cu = (int)((160 * 1024) / smem) >= 2 ? 2 : 1;
    const unsigned blocks = (unsigned)(a.nchunks < rat_max_blocks() * per_cu ? a.nchunks : rat_max_blocks() * per_cu);
    RAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
    if (dropout_p > 0.f && w_host->w_out != nullptr)              // Attention.to_out is Identity without a projection: no Dropout there
        a.drop = RatDrop{dropout_seed, (uint32_t)((double)dropout_p * 4294967296.0), 1.0f / (1.0f - dropout_p), w_host->drop_seed_dev};
    const int fast = fast_dim(a, {x, y, res, o_save, lse_save});
    const bool plain = res == x && out_scale == 1.0f && a.drop.threshold == 0;
    const bool have_planes = w_host->planes != nullptr && aligned16(w_host->planes);
    const bool dpad = d != B3_D && b3_dim(d) && heads == fast_heads(dim_head) && aligned16(x) && aligned16(y) && aligned16(res) &&
                      aligned16(o_save) && aligned16(lse_save) && aligned16(w_host->w_qkv) && aligned16(w_host->w_out);
    if (arith == RAT_ARITH_BF16X3 && (fast == 64 || dpad) && b3_shape(d, heads, dim_head, w_host) && b3_off32_ok(map_host) &&
        (have_planes || (workspace != nullptr && workspace_bytes >= B3_W_QKV + B3_W_OUT && aligned16(workspace)))) {
        const char* p_qkv;
        const char* p_out;
        if (have_planes) {                                       // split once per step by the caller (rat_split_weights_batch)
            p_qkv = static_cast<const char*>(w_host->planes);
            p_out = p_qkv + B3_W_QKV + B3_W_OUTT + B3_W_QKVT;
        } else {
            char* ws = reinterpret_cast<char*>(workspace);
            if (rat_launch_split_weights(w_host->w_qkv, B3_Q3, d, d, 0, ws, stream) ||
                rat_launch_split_weights(w_host->w_out, B3_D, B3_I, B3_I, 0, ws + B3_W_QKV, stream, 0, b3_nvalid(d))) return -1;
            p_qkv = ws;
            p_out = ws + B3_W_QKV;
        }
        Attn3W W{};
        W.qkv = RatWPlanes{reinterpret_cast<const rat_u4*>(p_qkv), 2};
        W.out = RatWPlanes{reinterpret_cast<const rat_u4*>(p_out), 3};
        const unsigned b3_blocks = (unsigned)(a.nchunks < rat_max_blocks() ? a.nchunks : rat_max_blocks());
        if (heads == 4) {                                        // 4 heads x 20 (RAT_m3): one general instantiation
            RAT_LAUNCH((attn_fwd3_kernel<true, false, false, false, 0, 4>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
            return rat_check_launch("rat_attn_fwd (bf16x3, 4 heads)");
        }
        if (dpad) {
            if (plain && a.nq < a.L) RAT_LAUNCH((attn_fwd3_kernel<false, true, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
            else if (plain) RAT_LAUNCH((attn_fwd3_kernel<false, false, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
            else RAT_LAUNCH((attn_fwd3_kernel<true, false, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        } else if (plain && a.nq < a.L) RAT_LAUNCH((attn_fwd3_kernel<false, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (plain && b3_fwd_matrix_core(a.L, a.nq) == 3)
            RAT_LAUNCH((attn_fwd3_kernel<false, false, false, false, 3>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (plain && b3_fwd_matrix_core(a.L, a.nq) == 2)
            RAT_LAUNCH((attn_fwd3_kernel<false, false, false, false, 2>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (plain && b3_fwd_matrix_core(a.L, a.nq) == 1)      // (sequences of at most 16 tokens: only when the knob forces it)
            RAT_LAUNCH((attn_fwd3_kernel<false, false, false, false, 1>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (plain) RAT_LAUNCH((attn_fwd3_kernel<false>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (b3_fwd_matrix_core(a.L, a.L) == 3)                // (EX computes every position whatever `queries` says)
            RAT_LAUNCH((attn_fwd3_kernel<true, false, false, false, 3>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (b3_fwd_matrix_core(a.L, a.L) == 2)
            RAT_LAUNCH((attn_fwd3_kernel<true, false, false, false, 2>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else RAT_LAUNCH((attn_fwd3_kernel<true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);   // (computes every position)
        return rat_check_launch("rat_attn_fwd (bf16x3)");
    }
    const bool head_vec = aligned8(o_save);            // else the run-time dim_head kernel (4-byte accesses), which stops at DH_MAX
    RAT_REQUIRE(head_vec || dim_head <= DH_MAX, "o_save must be 8-byte aligned for dim_head 20");
    if (!head_vec) RAT_LAUNCH((attn_fwd_kernel<0, 0>), blocks, ATT_THREADS, smem, stream, a);
    else if (fast == 64 && dim_head == 10 && plain) RAT_LAUNCH((attn_fwd_kernel<64, 10, false>), blocks, ATT_THREADS, smem, stream, a);
    else if (fast == 64 && dim_head == 10) RAT_LAUNCH((attn_fwd_kernel<64, 10>), blocks, ATT_THREADS, smem, stream, a);
    else if (fast == 16 && dim_head == 10) RAT_LAUNCH((attn_fwd_kernel<16, 10>), blocks, ATT_THREADS, smem, stream, a);
    else if (fast == 64 && dim_head == 20) RAT_LAUNCH((attn_fwd_kernel<64, 20>), blocks, ATT_THREADS, smem, stream, a);
    else if (dim_head == 20) RAT_LAUNCH((attn_fwd_kernel<0, 20>), blocks, ATT_THREADS, smem, stream, a);
    else if (dim_head == 10 && d == 10 && heads == 8) RAT_LAUNCH((attn_fwd_kernel<0, 10, true, 2, 8, 10>), blocks, ATT_THREADS, smem, stream, a);   // shipped Tmall (groups of 8 heads)
    else if (dim_head == 10 && d == 10 && heads == 2) RAT_LAUNCH((attn_fwd_kernel<0, 10, true, 2, 2, 10>), blocks, ATT_THREADS, smem, stream, a);   // shipped MovieLens
    else if (dim_head == 10 && d == 16 && heads == 2) RAT_LAUNCH((attn_fwd_kernel<0, 10, true, 2, 2, 16>), blocks, ATT_THREADS, smem, stream, a);   // BASELINE configs[0]
    else if (dim_head == 10 && d <= 16) RAT_LAUNCH((attn_fwd_kernel<0, 10, true, 2>), blocks, ATT_THREADS, smem, stream, a);
    else if (dim_head == 10) RAT_LAUNCH((attn_fwd_kernel<0, 10>), blocks, ATT_THREADS, smem, stream, a);   // e.g. the shipped KKBox d = 40
    else RAT_LAUNCH((attn_fwd_kernel<0, 0>), blocks, ATT_THREADS, smem, stream, a);
    return rat_check_launch("rat_attn_fwd");
}

// ---- wide heads in ONE forward launch (round 5): heads = G x 8, dim_head 10, embedding_dim 64 — BASELINE configs[4], the shipped Tmall
// config's 32 heads at d = 64.  `planes`: G x the full plane set [W_qkv | W_out^T | W_qkv^T | W_out] of the groups' weight slices, filled by
// the jobs of rat_attn_groups_split_jobs (rows g*80.. of the Q, K and V blocks of to_qkv.weight; columns g*80.. of to_out.weight — no
// permuted copy of the weights is needed); slice g of it is a valid RatAttnParams.planes for rat_attn_bwd_ex on that group.  o_save / lse_save: [G][ntok][80] / [G][ntok][8], group-major, so that every group's slice is what
// rat_attn_bwd_ex expects for a launch on that group.
static int b3_groups(int d, int heads, int dim_head) {
    return (d == B3_D && dim_head == B3_DH && heads > B3_H && heads % B3_H == 0 && heads / B3_H <= 8) ? heads / B3_H : 0;
}
// ... and at a small embedding dimension (the shipped Tmall geometry, d = 10): exact-fp32 kernels, weights addressed in place (no planes)
// heads per group: 8 x 10, or (round 6: RAT_m3's heads / 2 heads of width 2 dim_head) 4 x 20 — a group's inner width is 80 either way
static int wide_heads(int dim_head) { return dim_head == WG_DH ? WG_H : (dim_head == 2 * WG_DH ? WG_H / 2 : 0); }
static int wide_groups(int d, int heads, int dim_head) {
    const int wh = wide_heads(dim_head);
    return (d >= 1 && d <= 16 && wh > 0 && heads > wh && heads % wh == 0 && heads / wh <= 8) ? heads / wh : 0;
}
// bit 0: rat_attn_fwd_groups serves these dimensions; bit 1: rat_attn_bwd_groups does too
extern "C" int rat_attn_groups_supported(int d, int heads, int dim_head) {
    if (b3_groups(d, heads, dim_head)) return 1;
    const int G = wide_groups(d, heads, dim_head);
    return G == 0 ? 0 : (G <= WG_MAXG ? 3 : 1);
}
extern "C" size_t rat_attn_groups_planes_bytes(int d, int heads, int dim_head) {
    return (size_t)b3_groups(d, heads, dim_head) * B3_GRP_PLANES;
}
extern "C" int rat_attn_groups_split_jobs(const RatAttnParams* w_host, int d, int heads, int dim_head, void* planes, RatSplitJob* jobs_out) {
    RAT_REQUIRE(w_host && jobs_out, "null pointer");
    const int G = b3_groups(d, heads, dim_head);
    if (G == 0 || planes == nullptr || w_host->w_out == nullptr) return 0;
    RAT_REQUIRE(aligned16(planes) && w_host->w_qkv, "planes must be 16-byte aligned");
    const int I = heads * dim_head;
    int n = 0;
    const int rows = (B3_I << 8) | (I << 20);                    // RatSplitJob.perm: the group's Q | K | V rows = 80 out of every I (rat_split_row)
    for (int g = 0; g < G; ++g) {                                // the four jobs of rat_attn_split_jobs on group g's slices, read in place
        char* ws = static_cast<char*>(planes) + (size_t)g * B3_GRP_PLANES;
        const float* wq = w_host->w_qkv + (size_t)g * B3_I * d;
        const float* wo = w_host->w_out + (size_t)g * B3_I;
        jobs_out[n++] = RatSplitJob{wq, ws, B3_Q3, d, d, 0, rows, 0};                                              // forward + backward
        jobs_out[n++] = RatSplitJob{wo, ws + B3_W_QKV, B3_I, d, I, 1, 0, 0};                                       // backward: dO
        jobs_out[n++] = RatSplitJob{wq, ws + B3_W_QKV + B3_W_OUTT, B3_D, B3_Q3, d, 1, rows, 0};                    // backward: d(LN out)
        jobs_out[n++] = RatSplitJob{wo, ws + B3_W_QKV + B3_W_OUTT + B3_W_QKVT, B3_D, B3_I, I, 0, 0, 0};            // forward: out-proj
    }
    return n;
}
extern "C" int rat_attn_fwd_groups(const float* x, const float* res, float* y, float* o_save, float* lse_save, int64_t ntok,
                                   const RatAttnParams* w_host, const void* planes, const RatSeqMap* map_host, int d, int heads,
                                   int dim_head, float softmax_scale, float out_scale, float ln_eps, float dropout_p,
                                   uint64_t dropout_seed, void* stream) {
    const int Gw = wide_groups(d, heads, dim_head);
    if (Gw > 0) {                                                // small embedding_dim: exact fp32, weights in place, `planes` unused
        const int wh = wide_heads(dim_head);
        if (check_dims(map_host, d, wh, dim_head, false)) return -1;
        RAT_REQUIRE(x && y && w_host && w_host->ln_g && w_host->ln_b && w_host->w_qkv && w_host->w_out && w_host->b_out, "null pointer");
        RAT_REQUIRE((o_save == nullptr) == (lse_save == nullptr) && ntok > 0, "o_save and lse_save come together, [G][ntok][.]");
        AttnArgs a{};
        fill_common(a, w_host, map_host, d, wh, dim_head, ln_eps);
        a.vec_wout = ((heads * dim_head) % 4 == 0) && aligned16(w_host->w_out);
        a.vec_wqkv = (aligned8(x) && aligned16(w_host->w_out)) ? 1 : 0;   // (read by the kernel as "aligned for the one-round-trip loads")
        if (softmax_scale > 0.f) a.scale = softmax_scale;
        a.out_scale = out_scale;
        a.res = res;
        a.x = x;
        a.y = y;
        a.o_save = o_save;
        a.lse_save = lse_save;
        a.groups = Gw;
        a.group_tok = ntok;
        a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(y) && aligned16(res);
        RAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
        if (dropout_p > 0.f)
            a.drop = RatDrop{dropout_seed, (uint32_t)((double)dropout_p * 4294967296.0), 1.0f / (1.0f - dropout_p), w_host->drop_seed_dev};
        const size_t smem = AttnGeom(d, wh, dim_head).fwd_smem() + (size_t)WG_WO_FLOATS * 4;
        const int per_cu = (int)((160 * 1024) / smem) >= 2 ? 2 : 1;
        const unsigned blocks = (unsigned)(a.nchunks < rat_max_blocks() * per_cu ? a.nchunks : rat_max_blocks() * per_cu);
        if (wh == 4 && d == 10) RAT_LAUNCH((attn_fwd_wide_kernel<10, 4>), blocks, ATT_THREADS, smem, stream, a);
        else if (wh == 4) RAT_LAUNCH((attn_fwd_wide_kernel<0, 4>), blocks, ATT_THREADS, smem, stream, a);
        else if (d == 10) RAT_LAUNCH((attn_fwd_wide_kernel<10>), blocks, ATT_THREADS, smem, stream, a);
        else RAT_LAUNCH((attn_fwd_wide_kernel<0>), blocks, ATT_THREADS, smem, stream, a);
        return rat_check_launch("rat_attn_fwd_groups (small d)");
    }
    const int G = b3_groups(d, heads, dim_head);
    RAT_REQUIRE(G > 0, "rat_attn_fwd_groups serves dim_head 10 and 16 ... 64 heads in groups of 8 at embedding_dim 64 or <= 16 (there also 8 ... 32 heads of width 20)");
    if (check_dims(map_host, d, B3_H, dim_head, false)) return -1;
    RAT_REQUIRE(x && y && planes && w_host && w_host->ln_g && w_host->ln_b && w_host->b_out, "null pointer");
    RAT_REQUIRE((o_save == nullptr) == (lse_save == nullptr) && ntok > 0, "o_save and lse_save come together, [G][ntok][.]");
    RAT_REQUIRE(aligned16(x) && aligned16(y) && aligned16(res) && aligned16(o_save) && aligned16(lse_save) && aligned16(planes),
                "tensors must be 16-byte aligned");
    RAT_REQUIRE(b3_off32_ok(map_host), "token offsets beyond 32 bits");
    AttnArgs a{};
    fill_common(a, w_host, map_host, d, B3_H, dim_head, ln_eps);
    if (softmax_scale > 0.f) a.scale = softmax_scale;
    a.out_scale = out_scale;
    a.res = res;
    a.x = x;
    a.y = y;
    a.o_save = o_save;
    a.lse_save = lse_save;
    a.groups = G;
    a.group_tok = ntok;
    a.vec_x = 1;
    RAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
    if (dropout_p > 0.f)
        a.drop = RatDrop{dropout_seed, (uint32_t)((double)dropout_p * 4294967296.0), 1.0f / (1.0f - dropout_p), w_host->drop_seed_dev};
    Attn3W W{};
    W.qkv = RatWPlanes{reinterpret_cast<const rat_u4*>(planes), 2};
    W.out = RatWPlanes{reinterpret_cast<const rat_u4*>(static_cast<const char*>(planes) + B3_W_QKV + B3_W_OUTT + B3_W_QKVT), 3};
    const unsigned blocks = (unsigned)(a.nchunks < rat_max_blocks() ? a.nchunks : rat_max_blocks());
    if (b3_fwd_matrix_core(a.L, a.L) == 2) RAT_LAUNCH((attn_fwd3_kernel<true, false, false, true, 2>), blocks, ATT_THREADS, B3_FWD_WOUT, stream, a, W);
    else RAT_LAUNCH((attn_fwd3_kernel<true, false, false, true>), blocks, ATT_THREADS, B3_FWD_WOUT, stream, a, W);
    return rat_check_launch("rat_attn_fwd_groups");
}

// The backward of the same layer in one launch — small embedding dimensions only (rat_attn_groups_supported bit 1; see attn_bwd_wide_kernel
// for why embedding_dim 64 cannot).  o_save / lse_save: what rat_attn_fwd_groups saved; gradients in the layer's full-width layout.
extern "C" size_t rat_attn_bwd_groups_workspace(int d, int heads, int dim_head) {
    const int G = wide_groups(d, heads, dim_head);
    if (G == 0 || G > WG_MAXG) return 0;
    return (size_t)256 * (size_t)AttnGeom(d, heads, dim_head).slab_floats() * sizeof(float);
}
extern "C" int rat_attn_bwd_groups(const float* x, const float* dy, const float* add, const float* o_save, const float* lse_save, int64_t ntok,
                                   float* dx, const RatAttnParams* w_host, const RatAttnParams* grads_host, float* workspace,
                                   size_t workspace_bytes, const RatSeqMap* map_host, int d, int heads, int dim_head, float softmax_scale,
                                   float out_scale, float ln_eps, float dropout_p, uint64_t dropout_seed, void* stream) {
    const int G = wide_groups(d, heads, dim_head);
    RAT_REQUIRE(G > 0 && G <= WG_MAXG, "rat_attn_bwd_groups serves embedding_dim <= 16 and 2 ... 4 head groups of 8 x 10 or 4 x 20");
    const int wh = wide_heads(dim_head);
    if (check_dims(map_host, d, wh, dim_head, true)) return -1;
    RAT_REQUIRE(x && dy && o_save && lse_save && dx && w_host && grads_host && workspace && ntok > 0, "null pointer");
    RAT_REQUIRE(w_host->ln_g && w_host->ln_b && w_host->w_qkv && w_host->w_out && w_host->b_out, "null parameter");
    RAT_REQUIRE(grads_host->ln_g && grads_host->ln_b && grads_host->w_qkv && grads_host->w_out && grads_host->b_out, "null gradient");
    RAT_REQUIRE(workspace_bytes >= rat_attn_bwd_groups_workspace(d, heads, dim_head), "workspace too small");
    AttnArgs a{};
    fill_common(a, w_host, map_host, d, wh, dim_head, ln_eps);
    if (softmax_scale > 0.f) a.scale = softmax_scale;
    a.out_scale = out_scale;
    a.add = add;
    RAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
    if (dropout_p > 0.f)
        a.drop = RatDrop{dropout_seed, (uint32_t)((double)dropout_p * 4294967296.0), 1.0f / (1.0f - dropout_p), w_host->drop_seed_dev};
    a.add_lds = (add == dy && out_scale == 1.0f && a.drop.threshold == 0) ? 1 : 0;
    a.x = x;
    a.dy = dy;
    a.y = dx;
    a.o_save = const_cast<float*>(o_save);
    a.lse_save = const_cast<float*>(lse_save);
    a.groups = G;
    a.group_tok = ntok;
    a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(dy);
    // (read by the kernel as "every array is aligned for the one-round-trip loads": 8-byte rows of x / dy / W_qkv, 16-byte O / W_out)
    a.vec_wqkv = (aligned8(x) && aligned8(dy) && aligned16(o_save) && aligned8(w_host->w_qkv) && aligned16(w_host->w_out)) ? 1 : 0;
    const AttnGeom gfull(d, heads, dim_head);
    a.slabs = workspace;
    a.slab_stride = gfull.slab_floats();
    const int blocks = (int)(a.nchunks < rat_max_blocks() ? a.nchunks : rat_max_blocks());
    const size_t smem = AttnGeom(d, wh, dim_head).bwd_smem(wh) + (size_t)(WG_WQ_FLOATS + WG_WO_FLOATS) * 4;
    if (wh == 4 && d == 10) RAT_LAUNCH((attn_bwd_wide_kernel<10, 4>), blocks, ATT_THREADS, smem, stream, a);
    else if (wh == 4) RAT_LAUNCH((attn_bwd_wide_kernel<0, 4>), blocks, ATT_THREADS, smem, stream, a);
    else if (d == 10) RAT_LAUNCH((attn_bwd_wide_kernel<10>), blocks, ATT_THREADS, smem, stream, a);
    else RAT_LAUNCH((attn_bwd_wide_kernel<0>), blocks, ATT_THREADS, smem, stream, a);
    if (rat_check_launch("rat_attn_bwd_groups")) return -1;
    const int D = d, I = heads * dim_head;
    float* outs[5] = {grads_host->w_qkv, grads_host->w_out, grads_host->b_out, grads_host->ln_g, grads_host->ln_b};
    const int64_t sizes[5] = {(int64_t)3 * I * D, (int64_t)D * I, D, D, D};
    const int64_t offs[5] = {0, (int64_t)3 * I * D, (int64_t)3 * I * D + (int64_t)D * I, (int64_t)3 * I * D + (int64_t)D * I + D,
                             (int64_t)3 * I * D + (int64_t)D * I + 2 * D};
    return rat_launch_reduce_slabs(workspace, blocks, a.slab_stride, outs, offs, sizes, 5, stream);
}

// 1 when the fused kernels (forward AND backward) serve these dimensions, 0 when the caller has to take the composed path
// (K2c LayerNorm -> rat_sgemm -> rat_attn_core_*_map -> rat_sgemm): sequences above 64 tokens, or heads*dim_head too wide for
// the LDS tile / the in-register weight-gradient accumulators (the shipped Tmall config: 32 heads x 10).
extern "C" int rat_attn_fused_supported(int d, int heads, int dim_head, int L) {
    if (d <= 0 || heads <= 0 || dim_head <= 0 || L < 1 || L > ATT_ROWS) return 0;
    if (!(dim_head <= DH_MAX || dim_head == 20) || d > 128) return 0;
    const AttnGeom g(d, heads, dim_head);
    if (g.fwd_smem() > 160 * 1024 || g.bwd_smem(heads) > 160 * 1024) return 0;
    if ((g.Q16 / 16) * (g.D16 / 16) > QSLOTS * ATT_WAVES || (g.D16 / 16) * (g.I16 / 16) > OSLOTS * ATT_WAVES) return 0;
    return 1;
}

extern "C" size_t rat_attn_bwd_workspace(int d, int heads, int dim_head) {
    const AttnGeom g(d, heads, dim_head);
    const size_t slabs = (size_t)256 * (size_t)g.slab_floats() * sizeof(float);
    const bool b3 = b3_geom(d, heads, dim_head);                             // + the pre-split weight fragments of the bf16x3 kernel
    return slabs + (b3 ? B3_W_QKV + B3_W_OUTT + B3_W_QKVT : 0);
}

extern "C" int rat_attn_bwd(const float* x, const float* dy, const float* o_save, const float* lse_save, float* dx,
                            const RatAttnParams* w_host, const RatAttnParams* grads_host, float* workspace,
                            size_t workspace_bytes, const RatSeqMap* map_host, int d, int heads, int dim_head,
                            float ln_eps, void* stream) {
    return rat_attn_bwd_ex(x, dy, dy, o_save, lse_save, dx, w_host, grads_host, workspace, workspace_bytes, map_host, d, heads,
                           dim_head, 0.f, 1.f, ln_eps, 0.f, 0, RAT_ARITH_F32, stream);
}

extern "C" int rat_attn_bwd_ex(const float* x, const float* dy, const float* add, const float* o_save, const float* lse_save,
                               float* dx, const RatAttnParams* w_host, const RatAttnParams* grads_host, float* workspace,
                               size_t workspace_bytes, const RatSeqMap* map_host, int d, int heads, int dim_head,
                               float softmax_scale, float out_scale, float ln_eps, float dropout_p, uint64_t dropout_seed, int arith,
                               void* stream) {
    if (check_dims(map_host, d, heads, dim_head, true)) return -1;
    RAT_REQUIRE(x && dy && o_save && lse_save && dx && w_host && grads_host && workspace, "null pointer");
    RAT_REQUIRE(w_host->w_out != nullptr || heads * dim_head == d, "missing w_out");
    RAT_REQUIRE(workspace_bytes >= rat_attn_bwd_workspace(d, heads, dim_head), "workspace too small");
    AttnArgs a{};
    fill_common(a, w_host, map_host, d, heads, dim_head, ln_eps);
    if (softmax_scale > 0.f) a.scale = softmax_scale;
    a.out_scale = out_scale;
    a.add = add;
    RAT_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
    if (dropout_p > 0.f && w_host->w_out != nullptr)
        a.drop = RatDrop{dropout_seed, (uint32_t)((double)dropout_p * 4294967296.0), 1.0f / (1.0f - dropout_p), w_host->drop_seed_dev};
    a.add_lds = (add == dy && out_scale == 1.0f && a.drop.threshold == 0) ? 1 : 0;
    a.x = x;
    a.dy = dy;
    a.y = dx;
    a.o_save = const_cast<float*>(o_save);
    a.lse_save = const_cast<float*>(lse_save);
    a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(o_save);
    const AttnGeom g(d, heads, dim_head);
    a.slabs = workspace;
    a.slab_stride = g.slab_floats();
    const int blocks = (int)(a.nchunks < rat_max_blocks() ? a.nchunks : rat_max_blocks());
    const bool ct_shape = dim_head == 10 && ((d == 10 && (heads == 8 || heads == 2)) || (d == 16 && heads == 2));
    const size_t smem = g.bwd_smem(heads) + (ct_shape ? ct_weight_floats(g) * sizeof(float) : 0);     // (attn_bwd_kernel CTW)
    const int fast = fast_dim(a, {x, dy, add, o_save, dx});
    const bool dpad = d != B3_D && b3_dim(d) && heads == fast_heads(dim_head) && aligned16(x) && aligned16(dy) && aligned16(add) &&
                      aligned16(o_save) && aligned16(dx) && aligned16(w_host->w_qkv) && aligned16(w_host->w_out);
    if (arith == RAT_ARITH_BF16X3 && (fast == 64 || dpad) && b3_shape(d, heads, dim_head, w_host) && b3_off32_ok(map_host) && aligned16(workspace) &&
        (a.slab_stride * 256 * 4) % 16 == 0) {
        const char* ws;
        if (w_host->planes != nullptr && aligned16(w_host->planes)) {     // split once per step by the caller (rat_split_weights_batch)
            ws = static_cast<const char*>(w_host->planes);
        } else {
            char* wsw = reinterpret_cast<char*>(workspace) + (size_t)256 * a.slab_stride * sizeof(float);
            if (rat_launch_split_weights(w_host->w_qkv, B3_Q3, d, d, 0, wsw, stream) ||
                rat_launch_split_weights(w_host->w_out, B3_I, d, B3_I, 1, wsw + B3_W_QKV, stream) ||
                rat_launch_split_weights(w_host->w_qkv, B3_D, B3_Q3, d, 1, wsw + B3_W_QKV + B3_W_OUTT, stream, 0, b3_nvalid(d))) return -1;
            ws = wsw;
        }
        Attn3W W{};
        W.qkv = RatWPlanes{reinterpret_cast<const rat_u4*>(ws), 2};
        W.outT = RatWPlanes{reinterpret_cast<const rat_u4*>(ws + B3_W_QKV), 2};
        W.qkvT = RatWPlanes{reinterpret_cast<const rat_u4*>(ws + B3_W_QKV + B3_W_OUTT), 8};
        if (heads == 4) {                                        // 4 heads x 20 (RAT_m3): the general form, P handed to pass 2 where it fits
            if (b3_ph_fits(a.L, a.nsq_chunk, 4) && b3_ph_enabled())
                RAT_LAUNCH((attn_bwd3_kernel<true, false, false, true, 0, 4>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
            else RAT_LAUNCH((attn_bwd3_kernel<true, false, false, false, 0, 4>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        } else if (dpad) {
            if (a.add_lds && a.nq < a.L) RAT_LAUNCH((attn_bwd3_kernel<false, true, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
            else if (a.add_lds) RAT_LAUNCH((attn_bwd3_kernel<false, false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
            else RAT_LAUNCH((attn_bwd3_kernel<true, false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        } else if (a.add_lds && a.nq < a.L) RAT_LAUNCH((attn_bwd3_kernel<false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else if (b3_matrix_core(a.L) && a.nq >= a.L) {
            switch (b3_matrix_core(a.L) * 2 + (a.add_lds ? 0 : 1)) {   // (code, EX)
                case 2: RAT_LAUNCH((attn_bwd3_kernel<false, false, false, false, 1>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W); break;
                case 3: RAT_LAUNCH((attn_bwd3_kernel<true, false, false, false, 1>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W); break;
                case 4: RAT_LAUNCH((attn_bwd3_kernel<false, false, false, false, 2>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W); break;
                case 5: RAT_LAUNCH((attn_bwd3_kernel<true, false, false, false, 2>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W); break;
                case 26: RAT_LAUNCH((attn_bwd3_kernel<false, false, false, false, 13>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W); break;
                default: RAT_LAUNCH((attn_bwd3_kernel<true, false, false, false, 13>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W); break;
            }
        }
        else if (a.add_lds && b3_ph_fits(a.L, a.nsq_chunk) && b3_ph_enabled())
            RAT_LAUNCH((attn_bwd3_kernel<false, false, false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else if (a.nq >= a.L && b3_ph_fits(a.L, a.nsq_chunk) && b3_ph_enabled())      // (wide heads: groups 1 ... G - 1 add onto dx, not dy)
            RAT_LAUNCH((attn_bwd3_kernel<true, false, false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else if (a.add_lds) RAT_LAUNCH((attn_bwd3_kernel<false>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else RAT_LAUNCH((attn_bwd3_kernel<true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);       // (computes every position)
    } else if (!aligned8(o_save)) {                    // run-time dim_head kernel: 4-byte accesses, dim_head <= DH_MAX
        RAT_REQUIRE(dim_head <= DH_MAX, "o_save must be 8-byte aligned for dim_head 20");
        RAT_LAUNCH((attn_bwd_kernel<0, 0>), blocks, ATT_THREADS, smem, stream, a);
    } else if (fast == 64 && dim_head == 10 && a.add_lds) RAT_LAUNCH((attn_bwd_kernel<64, 10, false>), blocks, ATT_THREADS, smem, stream, a);
    else if (fast == 64 && dim_head == 10) RAT_LAUNCH((attn_bwd_kernel<64, 10>), blocks, ATT_THREADS, smem, stream, a);
    else if (fast == 16 && dim_head == 10) RAT_LAUNCH((attn_bwd_kernel<16, 10>), blocks, ATT_THREADS, smem, stream, a);
    else if (fast == 64 && dim_head == 20) RAT_LAUNCH((attn_bwd_kernel<64, 20>), blocks, ATT_THREADS, smem, stream, a);
    else if (dim_head == 20) RAT_LAUNCH((attn_bwd_kernel<0, 20>), blocks, ATT_THREADS, smem, stream, a);
    else if (dim_head == 10 && d == 10 && heads == 8) RAT_LAUNCH((attn_bwd_kernel<0, 10, true, 2, 8, 10>), blocks, ATT_THREADS, smem, stream, a);   // shipped Tmall (groups of 8 heads)
    else if (dim_head == 10 && d == 10 && heads == 2) RAT_LAUNCH((attn_bwd_kernel<0, 10, true, 2, 2, 10>), blocks, ATT_THREADS, smem, stream, a);   // shipped MovieLens
    else if (dim_head == 10 && d == 16 && heads == 2) RAT_LAUNCH((attn_bwd_kernel<0, 10, true, 2, 2, 16>), blocks, ATT_THREADS, smem, stream, a);   // BASELINE configs[0]
    else if (dim_head == 10 && d <= 16) RAT_LAUNCH((attn_bwd_kernel<0, 10, true, 2>), blocks, ATT_THREADS, smem, stream, a);
    else if (dim_head == 10) RAT_LAUNCH((attn_bwd_kernel<0, 10>), blocks, ATT_THREADS, smem, stream, a);   // e.g. the shipped KKBox d = 40
    else RAT_LAUNCH((attn_bwd_kernel<0, 0>), blocks, ATT_THREADS, smem, stream, a);
    if (rat_check_launch("rat_attn_bwd")) return -1;
    const int D = d, I = heads * dim_head;
    float* outs[5] = {grads_host->w_qkv, grads_host->w_out, grads_host->b_out, grads_host->ln_g, grads_host->ln_b};
    const int64_t sizes[5] = {(int64_t)3 * I * D, w_host->w_out ? (int64_t)D * I : 0, w_host->w_out ? D : 0, D, D};
    const int64_t offs[5] = {0, (int64_t)3 * I * D, (int64_t)3 * I * D + (int64_t)D * I,
                             (int64_t)3 * I * D + (int64_t)D * I + D, (int64_t)3 * I * D + (int64_t)D * I + 2 * D};
    return rat_launch_reduce_slabs(workspace, blocks, a.slab_stride, outs, offs, sizes, 5, stream);
}
