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

// The same for the bf16x3 kernels, which only run when every token's byte offset fits 32 bits (b3_off32_ok): the whole map in 32-bit
// arithmetic (exact modulo 2^32, and the true values are below it).  The 64-bit form kept a loop-invariant 64-bit product per lane
// alive across the chunk loop of attn_bwd3_kernel — the one value that kernel spilled to scratch (round 4's resource report: 12 bytes
// of scratch, a `scratch_load_dwordx2 ... Folded Reload` inside the loop, which waits for vmcnt(0)).
__device__ __forceinline__ void map_rows(const AttnArgs& a, int64_t chunk, int64_t* rowtok, int& nsq, int& rows);
__device__ __forceinline__ void map_rows32(const AttnArgs& a, int64_t chunk, int64_t* rowtok, int& nsq, int& rows) {
    const int64_t q0 = chunk * a.nsq_chunk;
    const int64_t left = a.nseq - q0;
    nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
    rows = nsq * a.L;
    if (threadIdx.x < ATT_ROWS) {
        const unsigned r = threadIdx.x, Lu = (unsigned)a.L;
        const unsigned sq = r / Lu, p = r - sq * Lu;
        const unsigned qq = (unsigned)q0 + sq, dv = a.q_div > 0x7fffffffLL ? 0x7fffffffu : (unsigned)a.q_div, hi = qq / dv;
        const unsigned tok = hi * (unsigned)a.hi_stride + (qq - hi * dv) * (unsigned)a.lo_stride + p * (unsigned)a.pos_stride;
        rowtok[r] = (int)r < rows ? (int64_t)tok : (int64_t)-1;
    }
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

// the 64-bit map with the lane's row number made opaque at every call: nothing of it can be hoisted out of the chunk loop (A/B only)
__device__ __forceinline__ void map_rows_nohoist(const AttnArgs& a, int64_t chunk, int64_t* rowtok, int& nsq, int& rows) {
    const int64_t q0 = chunk * a.nsq_chunk;
    const int64_t left = a.nseq - q0;
    nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
    rows = nsq * a.L;
    if (threadIdx.x < ATT_ROWS) {
        int r = threadIdx.x;
#ifndef RAT_EMU
        asm volatile("" : "+v"(r));
#endif
        rowtok[r] = r < rows ? seq_token(a, q0 + r / a.L, r % a.L) : (int64_t)-1;
    }
}
// Which form the bf16x3 kernels use is a same-box A/B decision (profiles/round5/r5_map_rows_ab.txt): RAT_MAP_FWD / RAT_MAP_BWD
// 0 = 64-bit (round 4), 1 = 32-bit, 2 = 64-bit without hoisting.
#ifndef RAT_MAP_FWD
#define RAT_MAP_FWD 0
#endif
#ifndef RAT_MAP_BWD
#define RAT_MAP_BWD 0
#endif
template <int MODE>
__device__ __forceinline__ void map_rows_b3(const AttnArgs& a, int64_t chunk, int64_t* rowtok, int& nsq, int& rows) {
    if (MODE == 1) map_rows32(a, chunk, rowtok, nsq, rows);
    else if (MODE == 2) map_rows_nohoist(a, chunk, rowtok, nsq, rows);
    else map_rows(a, chunk, rowtok, nsq, rows);
}

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

// Warm L2 with the NEXT chunk's rows: every thread touches one dword of one 128-byte line (64 rows x `bytes` per
// source).  Issued at the start of a phase that performs no other global loads, so the HBM round trip hides behind it
// and the next iteration's tile loads hit L2.  Returns the touched value; the caller keeps it alive until loop end.
__device__ __forceinline__ float prefetch_lines(const AttnArgs& a, int64_t chunk, int slot, int nlines_row, const float* src,
                                                int width) {
    // slot in [0, 64 * nlines_row): row = slot / nlines_row, line = slot % nlines_row
    if (chunk >= a.nchunks) return 0.f;
    const int r = slot / nlines_row, ln = slot - r * nlines_row;
    const int64_t q0 = chunk * a.nsq_chunk;
    const int64_t left = a.nseq - q0;
    const int nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
    if (r >= nsq * a.L) return 0.f;
    const int64_t tok = seq_token(a, q0 + r / a.L, r % a.L);
    int c = ln * 32;
    if (c >= width) c = width - 1;
    return src[tok * width + c];
}

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

// =============================================================================================================================
// Wide heads at a SMALL embedding dimension, exact fp32 — the shipped Tmall geometry (configs/RAT_m2/tmall_x1_002/model_config.yaml:
// embedding_dim 10, 32 heads x 10).  heads * dim_head = 320 does not fit the fused kernels' LDS tile, so the layer ran as G = heads / 8
// launches of attn_fwd_kernel / attn_bwd_kernel<0, 10, true, 2, 8, 10> on 8 heads each.  These two kernels run the WHOLE layer in one
// launch each: a chunk is loaded and normalised once and the head groups are looped over inside it (RAT_m2.py:192-202: the heads only
// meet in to_out).  Unlike at embedding_dim 64 (attn_bwd3_kernel: 80 accumulator tiles per GROUP) the backward loop fits here too:
// with ONE 16-wide column tile for d <= 16 a group's weight gradients are 15 + 5 accumulator tiles, four groups' are the 80 tiles =
// 48 VGPRs per lane that one group needs at d = 64.  Group g's weights are addressed in place: rows g*80.. of the Q, K and V blocks of
// to_qkv.weight (80 = 5 tiles of 16, so a tile never straddles two blocks), columns g*80.. of to_out.weight.  The parameter-gradient
// slabs are written in the layer's FULL layout ([3 I][d], [d][I], I = 80 G), so the gradients land in place as well.
// LDS map = the 8-head generic kernels' (xs [64][20], dys [64][20], qkv [64][244], ob / dob [64][84], ...) + the current group's weight
// slices (wide_stage_wq / _wo): 74 KB forward (two work-groups per CU), 138 KB backward.
constexpr int WG_H = 8, WG_DH = 10, WG_I = WG_H * WG_DH, WG_Q3 = 3 * WG_I, WG_LDX = 20, WG_LDQ = WG_Q3 + 4, WG_LDT = WG_I + 4, WG_COLS = 2,
              WG_MAXG = 4;
static_assert(WG_I % 16 == 0, "a 16-row weight tile must not straddle the Q / K / V blocks");

// B[k][n] = to_qkv.weight[row(n)][k] for the group's 240 Q|K|V columns n (the recomputed / forward projection)
struct WideWqkvNk {
    const float* w;
    int itot, g, D;
    __device__ __forceinline__ float4 operator()(int tile, int kb) const {
        const int l = rat_lane(), part = tile / (WG_I / 16);
        const int row = part * itot + g * WG_I + (tile - part * (WG_I / 16)) * 16 + (l & 15);
        const int k = kb * 16 + 4 * (l >> 4);
        const float* p = w + (size_t)row * D + k;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 0 < D) r.x = p[0];
        if (k + 1 < D) r.y = p[1];
        if (k + 2 < D) r.z = p[2];
        if (k + 3 < D) r.w = p[3];
        return r;
    }
};
// B[k][n] = to_qkv.weight[row(k)][n] for the group's 240 Q|K|V columns k (d(LayerNorm out) = dQKV W_qkv)
struct WideWqkvKn {
    const float* w;
    int itot, g, D;
    __device__ __forceinline__ float4 operator()(int tile, int kb) const {
        const int l = rat_lane(), part = kb / (WG_I / 16);
        const int row = part * itot + g * WG_I + (kb - part * (WG_I / 16)) * 16 + 4 * (l >> 4);
        const int n = tile * 16 + (l & 15);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < D) {
            const float* p = w + (size_t)row * D + n;
            r = make_float4(p[0], p[(size_t)D], p[(size_t)2 * D], p[(size_t)3 * D]);
        }
        return r;
    }
};

// Group g's weight slices -> LDS, once per (chunk, group): the GEMM phases then read their B operands from LDS instead of chasing them
// through L2 one dependent round trip per 16 x 16 tile (at these sizes a phase is a handful of MFMAs: the fetch latency WAS the phase).
//   wq_s [240][12]: row n = the group's Q|K|V column n, the d <= 10 weights of that row (12-float rows: the fourth k-quad of a fragment reads the next
//     row's first floats — finite, and multiplied by the zero padding columns of the activation tile; embedding_dim 11 ... 16 keeps to L2);
//   wo_s [16][80]: row k = output feature (rows >= d stay zero), the group's 80 columns of to_out.weight.
constexpr int WG_LDWQ = 12, WG_WQ_FLOATS = WG_Q3 * WG_LDWQ + 16, WG_WO_FLOATS = 16 * WG_I;
__device__ __forceinline__ void wide_stage_wq(float* wq_s, const float* w_qkv, int itot, int g, int D) {
    for (int e = threadIdx.x; e < WG_Q3 * D; e += ATT_THREADS) {
        const int r = e / D, c = e - r * D, part = r / WG_I;
        wq_s[r * WG_LDWQ + c] = w_qkv[(size_t)(part * itot + g * WG_I + (r - part * WG_I)) * D + c];
    }
}
__device__ __forceinline__ void wide_stage_wo(float* wo_s, const float* w_out, int itot, int g, int D) {
    for (int e = threadIdx.x; e < D * WG_I; e += ATT_THREADS) {
        const int k = e / WG_I, n = e - k * WG_I;
        wo_s[e] = w_out[(size_t)k * itot + g * WG_I + n];
    }
}

// The same two copies plus the chunk's O / lse / x / dy tiles with EVERY request issued before anything is stored (compile-time trip
// counts): one memory round trip per group instead of one per loop trip — the run-time loops above compile to load -> wait -> LDS store
// chains, 11 serial L2 round trips per group at the Tmall shape.  d = 10 with 8-byte aligned rows / 16-byte aligned O and W_out only.
struct WideGroupFetch {
    static constexpr int NO = (ATT_ROWS * (WG_I / 4) + ATT_THREADS - 1) / ATT_THREADS;      // float4 pieces of the O tile per thread (3)
    static constexpr int NQ = (WG_Q3 * 5 + ATT_THREADS - 1) / ATT_THREADS;                   // float2 pieces of the group's W_qkv rows (3)
    float4 o[NO], wo;
    float2 wq[NQ];
    float lse;
    __device__ __forceinline__ void issue(const float* o_g, const float* lse_g, const int64_t* rowtok, const float* w_qkv, const float* w_out,
                                          int itot, int g, bool with_o) {
        constexpr int W4 = WG_I / 4;
        if (with_o) {
            int64_t tok[NO];
#pragma unroll
            for (int it = 0; it < NO; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                tok[it] = e < ATT_ROWS * W4 ? rowtok[e / W4] : -1;
            }
            const int64_t tl = rowtok[threadIdx.x / WG_H];
#pragma unroll
            for (int it = 0; it < NO; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                o[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tok[it] >= 0) o[it] = *reinterpret_cast<const float4*>(o_g + tok[it] * WG_I + 4 * (e % W4));
            }
            lse = tl >= 0 ? lse_g[tl * WG_H + threadIdx.x % WG_H] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            wq[it] = make_float2(0.f, 0.f);
            if (w_qkv != nullptr && e < WG_Q3 * 5) {
                const int r = e / 5, c2 = e - r * 5, part = r / WG_I;
                wq[it] = *reinterpret_cast<const float2*>(w_qkv + (size_t)(part * itot + g * WG_I + (r - part * WG_I)) * 10 + 2 * c2);
            }
        }
        wo = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)threadIdx.x < 10 * W4) {
            const int k = threadIdx.x / W4, n4 = threadIdx.x - k * W4;
            wo = *reinterpret_cast<const float4*>(w_out + (size_t)k * itot + g * WG_I + 4 * n4);
        }
    }
    __device__ __forceinline__ void stash(float* ob, float* lses, float* wq_s, float* wo_s, bool with_o) const {
        constexpr int W4 = WG_I / 4;
        if (with_o) {
#pragma unroll
            for (int it = 0; it < NO; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                if (e < ATT_ROWS * W4) *reinterpret_cast<float4*>(ob + (size_t)(e / W4) * WG_LDT + 4 * (e % W4)) = o[it];
            }
            lses[threadIdx.x] = lse;
        }
        if (wq_s != nullptr) {
#pragma unroll
            for (int it = 0; it < NQ; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                if (e < WG_Q3 * 5) *reinterpret_cast<float2*>(wq_s + (e / 5) * WG_LDWQ + 2 * (e % 5)) = wq[it];
            }
        }
        if ((int)threadIdx.x < 10 * W4) *reinterpret_cast<float4*>(wo_s + 4 * threadIdx.x) = wo;
    }
};
static_assert(ATT_ROWS * WG_H == ATT_THREADS, "one lse element per thread");
// this thread's 8-byte piece of a [tokens][10] row (threads < 64 * 5)
__device__ __forceinline__ float2 wide_row_piece(const float* src, const int64_t* rowtok) {
    float2 v = make_float2(0.f, 0.f);
    if ((int)threadIdx.x < ATT_ROWS * 5) {
        const int64_t tok = rowtok[threadIdx.x / 5];
        if (tok >= 0) v = *reinterpret_cast<const float2*>(src + tok * 10 + 2 * (threadIdx.x % 5));
    }
    return v;
}
__device__ __forceinline__ void wide_row_stash(float* tile, const float2& v, float mul = 1.0f) {
    if ((int)threadIdx.x < ATT_ROWS * 5)
        *reinterpret_cast<float2*>(tile + (size_t)(threadIdx.x / 5) * WG_LDX + 2 * (threadIdx.x % 5)) = make_float2(v.x * mul, v.y * mul);
}

// touch one dword of every 128-byte line of `width`-float rows of the chunk whose map is `rt` (see prefetch_lines_map)
__device__ __forceinline__ float wide_touch(const int64_t* rt, int& t, const float* src, int width) {
    const int nl = (width * 4 + 127) / 128;
    float v = 0.f;
    if (t >= 0 && t < ATT_ROWS * nl) v = prefetch_lines_map(rt, t, nl, src, width);
    t -= ATT_ROWS * nl;
    return v;
}

// GD: the embedding dimension as a compile-time constant (10: the shipped geometry; 0: run-time, any d <= 16)
template <int GD>
__global__ void __launch_bounds__(ATT_THREADS, 4) attn_fwd_wide_kernel(AttnArgs a) {   // (4 waves per SIMD: two work-groups per CU)
    RAT_DYN_SMEM(smem);
    const int D = GD > 0 ? GD : a.d, L = a.L, G = a.groups, itot = G * WG_I;
    float* xs = reinterpret_cast<float*>(smem);                  // [64][20] LayerNorm(x), read by every group; at the end the y tile
    float* qkv = xs + (size_t)ATT_ROWS * WG_LDX;                 // [64][244] Q|K|V of the current group; O replaces Q
    int64_t* const rowtok0 = reinterpret_cast<int64_t*>(qkv + (size_t)ATT_ROWS * WG_LDQ);
    float* const wo_s = reinterpret_cast<float*>(rowtok0 + 2 * ATT_ROWS);   // [16][80] the current group's columns of to_out.weight
    for (int e = threadIdx.x; e < WG_WO_FLOATS; e += ATT_THREADS) wo_s[e] = 0.f;
    zero_cols(xs, WG_LDX, D);
    zero_cols(qkv, WG_LDQ, WG_Q3);
    {
        int nsq0, rows0;
        map_rows(a, blockIdx.x, rowtok0, nsq0, rows0);
    }
    __syncthreads();
    int parity = 0;
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x, parity ^= 1) {
        const int64_t* rowtok = rowtok0 + parity * ATT_ROWS;
        const int64_t* rowtok_next = rowtok0 + (parity ^ 1) * ATT_ROWS;
        int nsq, rows;
        {
            const int64_t q0 = chunk * a.nsq_chunk;
            const int64_t left = a.nseq - q0;
            nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
            rows = nsq * a.L;
        }
        const bool fast = GD == 10 && a.vec_wqkv != 0;          // (vec_wqkv: the host found every array aligned for the one-round-trip loads)
        if (fast) wide_row_stash(xs, wide_row_piece(a.x, rowtok));
        else load_rows(xs, WG_LDX, a.x, rowtok, D, a.vec_x != 0);
        __syncthreads();
        layer_norm_rows<WG_COLS, false>(xs, WG_LDX, D, rows, a.ln_g, a.ln_b, a.eps, nullptr, nullptr);
        const bool more = chunk + gridDim.x < a.nchunks;
        if (more) {
            int nsq1, rows1;
            map_rows(a, chunk + gridDim.x, rowtok0 + (parity ^ 1) * ATT_ROWS, nsq1, rows1);
        }
        __syncthreads();
        const int mt_valid = (rows + 15) / 16;
        f32x4 yacc[1][1] = {{rat_zero4()}};                      // waves 0-3: row tile w of the output projection, summed over the groups
        float pf = 0.f;
        for (int grp = 0; grp < G; ++grp) {
            if (fast) {                                          // (read two barriers from here)
                WideGroupFetch gf;
                gf.issue(nullptr, nullptr, rowtok, nullptr, a.w_out, itot, grp, false);
                gf.stash(nullptr, nullptr, nullptr, wo_s, false);
            } else {
                wide_stage_wo(wo_s, a.w_out, itot, grp, D);
            }
            // Q|K|V = LN(x) W_qkv[group]^T
            {
                const RatLdsRows A{xs, WG_LDX};
                const WideWqkvNk Bw{a.w_qkv, itot, grp, D};
                rat_gemm_phase<false, ATT_MT, ATT_WAVES, ATT_MT, 0>(A, Bw, mt_valid, WG_Q3 / 16, 1, [&](int mt, int nt, const f32x4& acc) {
                    const int col = rat_acc_col(nt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) qkv[(size_t)rat_acc_row(mt, r) * WG_LDQ + col] = acc[r];
                });
            }
            __syncthreads();
            if (grp == G - 1 && more) {                          // the core touches LDS only: the next chunk's x lines travel meanwhile
                int t = threadIdx.x;
                pf += wide_touch(rowtok_next, t, a.x, D);
            }
            // softmax(Q K^T * scale) V, one lane per (sequence, head, query) — attn_fwd_kernel's loop at compile-time dim_head 10
            typedef HeadVec<WG_DH> HV;
            float* const o_save = a.o_save != nullptr ? a.o_save + (int64_t)grp * a.group_tok * WG_I : nullptr;
            float* const lse_save = a.lse_save != nullptr ? a.lse_save + (int64_t)grp * a.group_tok * WG_H : nullptr;
            const int ntasks = nsq * WG_H * L;
            const float sl2 = a.scale * RAT_LOG2E;
            for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
                const int i = task % L;
                const int h = (task / L) % WG_H;
                const int sq = task / (L * WG_H);
                const int row_i = sq * L + i;
                float* qp = qkv + (size_t)row_i * WG_LDQ + h * WG_DH;
                HV q, o, kv;
                q.load(qp, WG_DH);
                o.zero();
                float m = -INFINITY, l = 0.f;
                const float* kbase = qkv + (size_t)(sq * L) * WG_LDQ + WG_I + h * WG_DH;
                int j = 0;
                for (; j + CORE_UNROLL <= L; j += CORE_UNROLL) {
                    HV kk[CORE_UNROLL], vv[CORE_UNROLL];
#pragma unroll
                    for (int u = 0; u < CORE_UNROLL; ++u) {
                        const float* kp = kbase + (size_t)(j + u) * WG_LDQ;
                        kk[u].load(kp, WG_DH);
                        vv[u].load(kp + WG_I, WG_DH);
                    }
                    float sc[CORE_UNROLL];
#pragma unroll
                    for (int u = 0; u < CORE_UNROLL; ++u) sc[u] = q.dot(kk[u]) * sl2;
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
                    const float* kp = kbase + (size_t)j * WG_LDQ;
                    kv.load(kp, WG_DH);
                    const float s = q.dot(kv) * sl2;
                    const float mn = fmaxf(m, s);
                    const float corr = rat_exp2(m - mn);
                    const float p = rat_exp2(s - mn);
                    l = l * corr + p;
                    kv.load(kp + WG_I, WG_DH);
                    o.scale_axpy(corr, p, kv);
                    m = mn;
                }
                const float inv = 1.0f / l;
                o.store(qp, WG_DH, inv);
                const int64_t tok = rowtok[row_i];
                if (o_save != nullptr) o.store(o_save + tok * WG_I + h * WG_DH, WG_DH, inv);
                if (lse_save != nullptr) lse_save[tok * WG_H + h] = m + rat_log2(l);
            }
            __syncthreads();
            // partial output projection O_g W_out[:, group]^T into the accumulators of waves 0-3 (B from the staged LDS copy)
            if (rat_wave() < mt_valid && rat_wave() < ATT_MT) {
                const RatLdsRows A{qkv, WG_LDQ};
                const RatLdsRows Bw{wo_s, WG_I};
                rat_wave_gemm<1, 1>(yacc, A, Bw, rat_wave(), 0, 1, 1, WG_I / 16);
            }
            __syncthreads();                                     // (the next group's projection overwrites the O columns)
        }
        if (rat_wave() < ATT_MT) {                               // y tile = sum of the partials + bias, staged in xs (dead since the last Q|K|V)
            const int col = rat_acc_col(0);
            if (col < D) {
                const float bias = a.b_out[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[(size_t)rat_acc_row(rat_wave(), r) * WG_LDX + col] = yacc[0][0][r] + bias;
            }
        }
        __syncthreads();
        store_rows_residual(a.y, xs, WG_LDX, a.res, rowtok, rows, D, a.vec_x != 0, a.out_scale, &a.drop);
        __syncthreads();
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));
#endif
    }
}

template <int GD>
__global__ void __launch_bounds__(ATT_THREADS) attn_bwd_wide_kernel(AttnArgs a) {
    RAT_DYN_SMEM(smem);
    const int D = GD > 0 ? GD : a.d, L = a.L, G = a.groups, itot = G * WG_I;
    float* xs = reinterpret_cast<float*>(smem);                  // [64][20] LayerNorm(x)            (every group)
    float* dys = xs + (size_t)ATT_ROWS * WG_LDX;                 // [64][20] dL/dy                    (every group)
    float* qkv = dys + (size_t)ATT_ROWS * WG_LDX;                // [64][244] Q|K|V, later dQ|dK|dV   (per group)
    float* ob = qkv + (size_t)ATT_ROWS * WG_LDQ;                 // [64][84] O, later dQ, later two partial d(LN out) tiles
    float* dob = ob + (size_t)ATT_ROWS * WG_LDT;                 // [64][84] dO, later two partial d(LN out) tiles
    float* mu = dob + (size_t)ATT_ROWS * WG_LDT;
    float* rs = mu + ATT_ROWS;
    float* lses = rs + ATT_ROWS;                                 // [64][8]
    float* dlt = lses + (size_t)ATT_ROWS * WG_H;                 // [64][8]
    int64_t* const rowtok0 = reinterpret_cast<int64_t*>(dlt + (size_t)ATT_ROWS * WG_H);
    float* const wq_s = reinterpret_cast<float*>(rowtok0 + 2 * ATT_ROWS);   // [240][12] the current group's rows of to_qkv.weight (wide_stage_wq)
    float* const wo_s = wq_s + WG_WQ_FLOATS;                                // [16][80]  ... and its columns of to_out.weight
    const bool lds_w = D <= 10;                                  // (embedding_dim 11 ... 16: the fragments' k-quads would not fit 12-float rows)
    for (int e = threadIdx.x; e < WG_WQ_FLOATS + WG_WO_FLOATS; e += ATT_THREADS) wq_s[e] = 0.f;

    // persistent parameter-gradient accumulators: group g's dW_qkv tiles {w, w + 8} of 15 and its dW_out^T tile w of 5
    f32x4 accq0[2], accq1[2], accq2[2], accq3[2], acco0[1], acco1[1], acco2[1], acco3[1];
#pragma unroll
    for (int s = 0; s < 2; ++s) accq0[s] = accq1[s] = accq2[s] = accq3[s] = rat_zero4();
    acco0[0] = acco1[0] = acco2[0] = acco3[0] = rat_zero4();
    float dgam[WG_COLS], dbet[WG_COLS], lng[WG_COLS];
    const int c0 = (threadIdx.x & 7) * WG_COLS;
#pragma unroll
    for (int k = 0; k < WG_COLS; ++k) {
        dgam[k] = dbet[k] = 0.f;
        lng[k] = c0 + k < D ? a.ln_g[c0 + k] : 0.f;
    }
    float dbo = 0.f;
    zero_cols(xs, WG_LDX, D);
    zero_cols(dys, WG_LDX, D);
    zero_cols(qkv, WG_LDQ, WG_Q3);
    zero_cols(ob, WG_LDT, 0);
    zero_cols(dob, WG_LDT, 0);
    {
        int nsq0, rows0;
        map_rows(a, blockIdx.x, rowtok0, nsq0, rows0);
    }
    __syncthreads();
    RAT_PROF_DECL
    int parity = 0;
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x, parity ^= 1) {
        const int64_t* rowtok = rowtok0 + parity * ATT_ROWS;
        const int64_t* rowtok_next = rowtok0 + (parity ^ 1) * ATT_ROWS;
        int nsq, rows;
        {
            const int64_t q0 = chunk * a.nsq_chunk;
            const int64_t left = a.nseq - q0;
            nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
            rows = nsq * a.L;
        }
        const bool more = chunk + gridDim.x < a.nchunks;
        // ---- once per chunk: x -> LayerNorm, dy (through the projection's Dropout and output scale)
        const bool fast = GD == 10 && a.vec_wqkv != 0;          // (vec_wqkv: the host found every array aligned for the one-round-trip loads)
        if (fast && a.drop.threshold == 0) {
            const float2 vx = wide_row_piece(a.x, rowtok), vd = wide_row_piece(a.dy, rowtok);
            wide_row_stash(xs, vx);
            wide_row_stash(dys, vd, a.out_scale);
        } else {
            load_rows(xs, WG_LDX, a.x, rowtok, D, a.vec_x != 0);
            load_rows(dys, WG_LDX, a.dy, rowtok, D, a.vec_x != 0, a.out_scale, &a.drop);
        }
        __syncthreads();
        layer_norm_rows<WG_COLS, false>(xs, WG_LDX, D, rows, a.ln_g, a.ln_b, a.eps, mu, rs);
        if (more) {
            int nsq1, rows1;
            map_rows(a, chunk + gridDim.x, rowtok0 + (parity ^ 1) * ATT_ROWS, nsq1, rows1);
        }
        {   // db_out partials: thread = (column, row group); combined once, after the chunk loop
            const int nrg = ATT_THREADS / D, col = threadIdx.x % D, rg = threadIdx.x / D;
            if (rg < nrg)
                for (int r = rg; r < rows; r += nrg) dbo += dys[(size_t)r * WG_LDX + col];
        }
        const int mt_valid = (rows + 15) / 16;
        float gsum[WG_COLS];                                     // d(LayerNorm out) of this thread's columns, summed over the groups
#pragma unroll
        for (int k = 0; k < WG_COLS; ++k) gsum[k] = 0.f;
        float pf = 0.f;
        RAT_PROF_MARK(0);
        for (int grp = 0; grp < G; ++grp) {
            const float* const o_g = a.o_save + (int64_t)grp * a.group_tok * WG_I;
            const float* const lse_g = a.lse_save + (int64_t)grp * a.group_tok * WG_H;
            if (fast) {                                          // O, lse and the group's weight slices: one round trip
                WideGroupFetch gf;
                gf.issue(o_g, lse_g, rowtok, a.w_qkv, a.w_out, itot, grp, true);
                gf.stash(ob, lses, wq_s, wo_s, true);
            } else {
                load_rows(ob, WG_LDT, o_g, rowtok, WG_I, false);
                if (lds_w) wide_stage_wq(wq_s, a.w_qkv, itot, grp, D);
                wide_stage_wo(wo_s, a.w_out, itot, grp, D);
                for (int e = threadIdx.x; e < ATT_ROWS * WG_H; e += ATT_THREADS) {
                    const int64_t tok = rowtok[e / WG_H];
                    lses[e] = tok >= 0 ? lse_g[tok * WG_H + e % WG_H] : 0.f;
                }
            }
            __syncthreads();                                     // (also: LayerNorm of xs, the dy tile — first group)
            RAT_PROF_MARK(1);
            // (1) recompute Q|K|V   (2) dO = dy W_out[:, group]   (3) dW_out^T[group] += O^T dy
            {
                const RatLdsRows A{xs, WG_LDX};
                auto epi = [&](int mt, int nt, const f32x4& acc) {
                    const int col = rat_acc_col(nt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) qkv[(size_t)rat_acc_row(mt, r) * WG_LDQ + col] = acc[r];
                };
                if (lds_w) rat_gemm_phase<false, ATT_MT, ATT_WAVES, ATT_MT, 0>(A, RatLdsRows{wq_s, WG_LDWQ}, mt_valid, WG_Q3 / 16, 1, epi);
                else rat_gemm_phase<false, ATT_MT, ATT_WAVES, ATT_MT, 0>(A, WideWqkvNk{a.w_qkv, itot, grp, D}, mt_valid, WG_Q3 / 16, 1, epi);
            }
            {
                const RatLdsRows A{dys, WG_LDX};
                const RatLdsCols Bw{wo_s, WG_I};                 // B[k][n] = W_out[k][group column n], rows k >= d are zero
                rat_gemm_phase<false, 2, ATT_WAVES, ATT_MT, 0>(A, Bw, mt_valid, WG_I / 16, 1, [&](int mt, int nt, const f32x4& acc) {
                    const int col = rat_acc_col(nt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dob[(size_t)rat_acc_row(mt, r) * WG_LDT + col] = acc[r];
                });
                const RatLdsCols At{ob, WG_LDT};
                const RatLdsCols Bt{dys, WG_LDX};
                switch (grp) {
                    case 0: rat_wave_gemm_slots<1, ATT_WAVES, 0>(acco0, At, Bt, WG_I / 16, 1, mt_valid); break;
                    case 1: rat_wave_gemm_slots<1, ATT_WAVES, 0>(acco1, At, Bt, WG_I / 16, 1, mt_valid); break;
                    case 2: rat_wave_gemm_slots<1, ATT_WAVES, 0>(acco2, At, Bt, WG_I / 16, 1, mt_valid); break;
                    default: rat_wave_gemm_slots<1, ATT_WAVES, 0>(acco3, At, Bt, WG_I / 16, 1, mt_valid); break;
                }
            }
            __syncthreads();
            RAT_PROF_MARK(2);
            // (4) attention backward on the VALU: attn_bwd_kernel's two passes at compile-time dim_head 10
            typedef HeadVec<WG_DH> HV;
            const int ntasks = nsq * WG_H * L;
            const float sl2 = a.scale * RAT_LOG2E;
            {   // both passes touch LDS only: the lines this block loads next travel HBM -> L2 meanwhile
                int t = threadIdx.x;
                if (grp + 1 < G) {
                    pf += wide_touch(rowtok, t, o_g + a.group_tok * WG_I, WG_I);
                    pf += wide_touch(rowtok, t, lse_g + a.group_tok * WG_H, WG_H);
                } else if (more) {
                    pf += wide_touch(rowtok_next, t, a.x, D);
                    pf += wide_touch(rowtok_next, t, a.dy, D);
                    pf += wide_touch(rowtok_next, t, a.o_save, WG_I);
                    pf += wide_touch(rowtok_next, t, a.lse_save, WG_H);
                }
            }
            for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
                const int i = task % L;
                const int h = (task / L) % WG_H;
                const int sq = task / (L * WG_H);
                const int row_i = sq * L + i;
                const int ho = h * WG_DH;
                float* op = ob + (size_t)row_i * WG_LDT + ho;
                HV q, go, dq, kv;
                q.load(qkv + (size_t)row_i * WG_LDQ + ho, WG_DH);
                go.load(dob + (size_t)row_i * WG_LDT + ho, WG_DH);
                kv.load(op, WG_DH);
                const float delta = go.dot(kv);
                dq.zero();
                dlt[row_i * WG_H + h] = delta;
                const float lse = lses[row_i * WG_H + h];
                const float* kbase = qkv + (size_t)(sq * L) * WG_LDQ + WG_I + ho;
                for (int j = 0; j < L; ++j) {
                    const float* kp = kbase + (size_t)j * WG_LDQ;
                    kv.load(kp + WG_I, WG_DH);
                    const float dp = go.dot(kv);
                    kv.load(kp, WG_DH);
                    const float p = rat_exp2(q.dot(kv) * sl2 - lse);
                    dq.axpy(p * (dp - delta), kv);
                }
                dq.store(op, WG_DH, a.scale);
            }
            __syncthreads();
            RAT_PROF_MARK(3);
            for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
                const int j = task % L;
                const int h = (task / L) % WG_H;
                const int sq = task / (L * WG_H);
                const int ho = h * WG_DH;
                float* kp = qkv + (size_t)(sq * L + j) * WG_LDQ + WG_I + ho;
                HV kk, vv, dk, dv, t;
                kk.load(kp, WG_DH);
                vv.load(kp + WG_I, WG_DH);
                dk.zero();
                dv.zero();
                for (int i = 0; i < L; ++i) {
                    const int row_i = sq * L + i;
                    t.load(dob + (size_t)row_i * WG_LDT + ho, WG_DH);
                    const float dp = t.dot(vv);
                    const float lse = lses[row_i * WG_H + h], delta = dlt[row_i * WG_H + h];
                    HV qv;
                    qv.load(qkv + (size_t)row_i * WG_LDQ + ho, WG_DH);
                    const float p = rat_exp2(qv.dot(kk) * sl2 - lse);
                    dv.axpy(p, t);
                    dk.axpy(p * (dp - delta), qv);
                }
                dk.store(kp, WG_DH, a.scale);
                dv.store(kp + WG_I, WG_DH, 1.0f);
            }
            __syncthreads();
            RAT_PROF_MARK(4);
            for (int e = threadIdx.x; e < rows * WG_I; e += ATT_THREADS) {     // dQ (in ob) -> the Q columns: qkv = d[Q|K|V]
                const int r = e / WG_I, c = e - r * WG_I;
                qkv[(size_t)r * WG_LDQ + c] = ob[(size_t)r * WG_LDT + c];
            }
            __syncthreads();
            RAT_PROF_MARK(5);
            // (5) d(LN out) partials = dQKV W_qkv[group]: ONE column tile, the contraction (15 k-blocks) split four ways over the waves
            //     (wave = (row-tile pair w & 1, K part w >> 1)); the partial tiles land side by side in dob / ob (dead now)
            {
                const RatLdsRows A{qkv, WG_LDQ};
                const int w = rat_wave(), mb = w & 1, part = w >> 1;
                constexpr int KBT = WG_Q3 / 16;
                const int k0 = part * KBT / 4, k1 = (part + 1) * KBT / 4;
                f32x4 acc[2] = {rat_zero4(), rat_zero4()};
                if (lds_w) rat_wave_gemm_col<2, 0>(acc, A, RatLdsCols{wq_s, WG_LDWQ}, 2 * mb, 0, k1, k0);   // (columns >= d of the tile: finite, never read)
                else rat_wave_gemm_col<2, 0>(acc, A, WideWqkvKn{a.w_qkv, itot, grp, D}, 2 * mb, 0, k1, k0);
                float* pt = (part < 2 ? dob : ob) + 16 * (part & 1);
                const int col = rat_acc_col(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[(size_t)rat_acc_row(2 * mb + i, r) * WG_LDT + col] = acc[i][r];
            }
            // (6) dW_qkv[group] += dQKV^T LN(x)
            {
                const RatLdsCols At{qkv, WG_LDQ};
                const RatLdsCols Bt{xs, WG_LDX};
                switch (grp) {
                    case 0: rat_wave_gemm_slots<2, ATT_WAVES, 0>(accq0, At, Bt, WG_Q3 / 16, 1, mt_valid); break;
                    case 1: rat_wave_gemm_slots<2, ATT_WAVES, 0>(accq1, At, Bt, WG_Q3 / 16, 1, mt_valid); break;
                    case 2: rat_wave_gemm_slots<2, ATT_WAVES, 0>(accq2, At, Bt, WG_Q3 / 16, 1, mt_valid); break;
                    default: rat_wave_gemm_slots<2, ATT_WAVES, 0>(accq3, At, Bt, WG_Q3 / 16, 1, mt_valid); break;
                }
            }
            __syncthreads();
            RAT_PROF_MARK(6);
            {   // this thread's columns of the four partial tiles -> the running sum over the groups
                const int r = threadIdx.x >> 3;
#pragma unroll
                for (int k = 0; k < WG_COLS; ++k) {
                    const int c = c0 + k;
                    if (c < D && r < rows)
                        gsum[k] += (dob[(size_t)r * WG_LDT + c] + dob[(size_t)r * WG_LDT + 16 + c]) +
                                   (ob[(size_t)r * WG_LDT + c] + ob[(size_t)r * WG_LDT + 16 + c]);
                }
            }
            __syncthreads();                                     // (the next group's O overwrites ob)
            RAT_PROF_MARK(7);
        }
        // ---- once per chunk: LayerNorm backward + the added gradient
        {
            const int r = threadIdx.x >> 3;
            const bool valid = r < rows;
            const int64_t tok = valid ? rowtok[r] : 0;
            const float mean = mu[r], rstd = rs[r];
            float xh[WG_COLS], out[WG_COLS], ad[WG_COLS];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < WG_COLS; ++k) {
                const int c = c0 + k;
                const bool on = c < D && valid;
                if (a.add_lds) ad[k] = on ? dys[(size_t)r * WG_LDX + c] : 0.f;
                else ad[k] = (on && a.add != nullptr) ? a.add[tok * D + c] : 0.f;
                xh[k] = on ? (a.x[tok * D + c] - mean) * rstd : 0.f;
                gsum[k] = on ? gsum[k] : 0.f;
                const float gw = gsum[k] * lng[k];
                s1 += gw;
                s2 += gw * xh[k];
            }
            s1 = rat_group_sum<8>(s1) / (float)D;
            s2 = rat_group_sum<8>(s2) / (float)D;
#pragma unroll
            for (int k = 0; k < WG_COLS; ++k) {
                const int c = c0 + k;
                const bool on = c < D && valid;
                const float gw = gsum[k] * lng[k];
                out[k] = on ? ad[k] + rstd * (gw - s1 - xh[k] * s2) : 0.f;
                dgam[k] += gsum[k] * xh[k];
                dbet[k] += gsum[k];
                if (on) a.y[tok * D + c] = out[k];
            }
        }
        __syncthreads();
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));
#endif
        RAT_PROF_MARK(8);
    }
    RAT_PROF_FLUSH(a.prof, 84);

    // ---- this work-group's parameter-gradient slab in the layer's FULL layout: [dW_qkv [3 I][d] | dW_out [d][I] | db_out | dgamma | dbeta]
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
    float* s_wqkv = slab;
    float* s_wout = s_wqkv + (int64_t)3 * itot * D;
    float* s_bout = s_wout + (int64_t)D * itot;
    float* s_gam = s_bout + D;
    float* s_bet = s_gam + D;
    {
        const int w = rat_wave(), col = rat_acc_col(0);
        auto put_q = [&](int grp, const f32x4 (&acc)[2]) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int id = w + ATT_WAVES * s;                // Q|K|V column tile of the group
                if (id < WG_Q3 / 16 && col < D && grp < G) {
                    const int part = id / (WG_I / 16), base = part * itot + grp * WG_I + (id - part * (WG_I / 16)) * 16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_wqkv[(int64_t)(base + rat_acc_row(0, r)) * D + col] = acc[s][r];
                }
            }
        };
        auto put_o = [&](int grp, const f32x4 (&acc)[1]) {
            if (w < WG_I / 16 && col < D && grp < G) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_wout[(int64_t)col * itot + grp * WG_I + rat_acc_row(w, r)] = acc[0][r];
            }
        };
        put_q(0, accq0); put_q(1, accq1); put_q(2, accq2); put_q(3, accq3);
        put_o(0, acco0); put_o(1, acco1); put_o(2, acco2); put_o(3, acco3);
    }
    {
        __syncthreads();
        float* red0 = dys;                                       // [nrg][D] partials
        const int nrg = ATT_THREADS / D, col = threadIdx.x % D, rg = threadIdx.x / D;
        if (rg < nrg) red0[rg * D + col] = dbo;
        __syncthreads();
        if ((int)threadIdx.x < D) {
            float sacc = 0.f;
            for (int k = 0; k < nrg; ++k) sacc += red0[k * D + threadIdx.x];
            s_bout[threadIdx.x] = sacc;
        }
    }
    float* red = xs;                                             // [64][20] is free now
    {
        const int r = threadIdx.x >> 3;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < WG_COLS; ++k)
                if (c0 + k < D) red[(size_t)r * WG_LDX + c0 + k] = which == 0 ? dgam[k] : dbet[k];
            __syncthreads();
            if ((int)threadIdx.x < D) {
                float sacc = 0.f;
                for (int rr = 0; rr < ATT_ROWS; ++rr) sacc += red[(size_t)rr * WG_LDX + threadIdx.x];
                (which == 0 ? s_gam : s_bet)[threadIdx.x] = sacc;
            }
        }
    }
}

// =============================================================================================================================
// bf16x3 variants of the north-star geometry (embedding_dim 64, 8 heads x 10): every projection runs on v_mfma_f32_16x16x32_bf16
// with 3-way split operands (rat_device.h "bf16x3": fp32-class accuracy at 2.7x the fp32-MFMA rate).  What changes against the
// kernels above is only how the GEMM operands are held and fetched:
//   * activations that feed a GEMM live in LDS as three bf16 PLANES, split ONCE by the thread that produces them (LayerNorm
//     output, the dy tile, the attention output O, dQ|dK|dV) — the GEMM loops contain no VALU work, only 16-byte LDS reads (row
//     operands), transposed 4 x 16 block reads (ds_read_b64_tr_b16: the token-contraction operands of the weight gradients) and
//     16-byte L2 loads of pre-split weight fragments;
//   * Q|K|V, dO and O stay fp32 tiles for the VALU attention core, which is unchanged (same instruction sequence => the softmax
//     statistics, the saved O / log-sum-exp and the pass structure are those of the exact-fp32 kernels).
constexpr int B3_D = 64, B3_I = 80, B3_Q3 = 240, B3_H = 8, B3_DH = 10;
constexpr int B3_LDQ = B3_Q3 + 4;                      // fp32 Q|K|V tile row (floats)
constexpr int B3_XP = 64 * 128;                         // one plane of a [64][64] tile (128-byte rows, swizzled)
constexpr int B3_OP = 64 * 160 + 64;                    // one plane of a [64][80] tile (160-byte rows) + slack for the padded K step
constexpr int B3_QP = 64 * 480;                         // one plane of a [64][240] tile (480-byte rows)
typedef RatPlanes<128, 7, B3_XP> PlanesX;
typedef RatPlanes<160, 0, B3_OP> PlanesO;
typedef RatPlanes<480, 0, B3_QP> PlanesQ;

struct Attn3W {                                         // pre-split weight fragments (rat_launch_split_weights)
    RatWPlanes qkv;      // B[k = d][n = qkv col]      = w_qkv[n][k]      N 240, K 64   (Q|K|V projection)
    RatWPlanes out;      // B[k = inner][n = d]        = w_out[n][k]      N 64,  K 80   (output projection, forward)
    RatWPlanes outT;     // B[k = d][n = inner]        = w_out[k][n]      N 80,  K 64   (dO = dy W_out, backward)
    RatWPlanes qkvT;     // B[k = qkv col][n = d]      = w_qkv[k][n]      N 64,  K 240  (d LN-out = dQKV W_qkv, backward)
};
constexpr size_t B3_W_QKV = (size_t)15 * 2 * 3 * 1024, B3_W_OUT = (size_t)4 * 3 * 3 * 1024, B3_W_OUTT = (size_t)5 * 2 * 3 * 1024,
                 B3_W_QKVT = (size_t)4 * 8 * 3 * 1024;
constexpr size_t B3_W_BYTES = B3_W_QKV + B3_W_OUT + B3_W_OUTT + B3_W_QKVT;

constexpr size_t B3_FWD_LSE = (size_t)3 * B3_XP + (size_t)64 * B3_LDQ * 4 + (size_t)3 * B3_OP + 2 * 64 * 8;   // [64][8] log-sum-exp of the chunk
constexpr size_t B3_FWD_WOUT = B3_FWD_LSE + (size_t)64 * B3_H * 4;      // the output projection's fragment planes, LDS-resident (36 KB)
constexpr size_t b3_fwd_smem() { return B3_FWD_WOUT + B3_W_OUT; }
constexpr size_t B3_GRP_PLANES = B3_W_BYTES;                           // rat_attn_fwd_groups: a head group's planes = the full RatAttnParams.planes set
//                                                                        [W_qkv | W_out^T | W_qkv^T | W_out], so that the backward's launch on the group takes them too
static_assert(b3_fwd_smem() <= 160 * 1024, "LDS budget (forward)");
// weight fragment planes held in LDS (same [n tile][K step][plane][lane] x 16 B layout as RatWPlanes): a fragment is three 16-byte
// LDS reads instead of a round trip to L2.  The forward kernel has 41 KB of LDS to spare, W_out's planes are 36 KB.
struct RatWPlanesLds {
    const char* base;
    int steps;
    __device__ __forceinline__ RatB3 operator()(int nt, int s) const {
        const char* p = base + ((size_t)(nt * steps + s) * 3) * 1024 + 16 * rat_lane();
        return RatB3{rat_as_bf16x8(*reinterpret_cast<const rat_u4*>(p)), rat_as_bf16x8(*reinterpret_cast<const rat_u4*>(p + 1024)),
                     rat_as_bf16x8(*reinterpret_cast<const rat_u4*>(p + 2048))};
    }
};

// LayerNorm of one row piece into planes: thread (row = tid / 8, sub = tid % 8) owns the 8 columns [8 sub, 8 sub + 8) = exactly
// one 16-byte piece, handed over in two float4 (loaded by the caller, usually a whole chunk ahead); same arithmetic, in the same
// order, as layer_norm_rows above.
// DPAD (embedding_dim d < 64, a multiple of 8, run inside the 64-wide tiles): `colok` says whether this thread's 8 columns exist; the
// pieces beyond d arrive as zeros (they add nothing to the mean), are left out of the variance, and leave as zeros (gamma = beta = 0).
template <bool DPAD = false>
__device__ __forceinline__ void b3_layer_norm_to_planes(bool valid, const float4& v0, const float4& v1, float eps, const PlanesX& xp,
                                                        const float (&gam)[8], const float (&bet)[8], float* mu_out, float* rs_out,
                                                        int dreal = B3_D, bool colok = true) {
    const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const float dn = DPAD ? (float)dreal : (float)B3_D;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += xv[k];
    const float mean = rat_group_sum<8>(s) / dn;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float t = xv[k] - mean;
        v += t * t;
    }
    if (DPAD) v = colok ? v : 0.f;
    const float rstd = 1.0f / sqrtf(rat_group_sum<8>(v) / dn + eps);
    float y[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = valid ? (xv[k] - mean) * rstd * gam[k] + bet[k] : 0.f;
    rat_u4 h, m, l;
    rat_split8(make_float4(y[0], y[1], y[2], y[3]), make_float4(y[4], y[5], y[6], y[7]), h, m, l);
    xp.store(r, sub, h, m, l);
    if (mu_out != nullptr && sub == 0) {
        mu_out[r] = mean;
        rs_out[r] = rstd;
    }
}
// this thread's piece of a token-indexed [.][64] tensor for the chunk whose row map is `rowtok` (zeros for padding rows)
// Token-indexed global accesses of the bf16x3 kernels: UNIFORM base (the kernel argument, in SGPRs) + 32-bit byte offset per lane.
// The 64-bit form (base + lane offset hoisted out of the chunk loop as a VGPR pair per array) got spilled, and every reload is a
// scratch load that waits for vmcnt(0): the loads of a phase went out one HBM round trip at a time.  The host only launches these
// kernels when every byte offset fits 32 bits (b3_off32_ok).  Loads are unconditional (padding rows read token 0 and are zeroed
// afterwards), so that nothing waits before the last load of the phase has been issued.
__device__ __forceinline__ float4 b3_ld4(const float* base, uint32_t byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float b3_ld1(const float* base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void b3_st4(float* base, uint32_t byte_off, const float4& v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ void b3_zero_unless(bool valid, float4& v) {
    v.x = valid ? v.x : 0.f; v.y = valid ? v.y : 0.f; v.z = valid ? v.z : 0.f; v.w = valid ? v.w : 0.f;
}
// this thread's 8-column piece of its row: byte offset of the piece in a [tokens][64] array
__device__ __forceinline__ uint32_t b3_piece_off(int64_t tok) {
    return (uint32_t)(tok >= 0 ? tok : 0) * (uint32_t)(B3_D * 4) + 32u * (threadIdx.x & 7);
}
// DPAD: rows are d floats; a thread whose piece does not exist points at piece 0 (its loads are unconditional and zeroed afterwards)
__device__ __forceinline__ uint32_t b3_piece_off_d(int64_t tok, int d, bool colok) {
    return (uint32_t)(tok >= 0 ? tok : 0) * (uint32_t)(d * 4) + (colok ? 32u * (threadIdx.x & 7) : 0u);
}
// the [64][80] O tile: 1280 float4 over 512 threads; element e -> row e / 20, float4 e % 20
struct B3RowFetchO {
    static constexpr int W4 = B3_I / 4;
    static constexpr int NIT = (ATT_ROWS * W4 + ATT_THREADS - 1) / ATT_THREADS;
    float4 v[NIT];
    unsigned valid;
    __device__ __forceinline__ void issue(const float* src, const int64_t* rowtok) {
        uint32_t off[NIT];
        valid = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            const int r = e < ATT_ROWS * W4 ? e / W4 : 0;
            const int64_t tok = rowtok[r];
            const bool ok = e < ATT_ROWS * W4 && tok >= 0;
            valid |= ok ? 1u << it : 0u;
            off[it] = (uint32_t)(ok ? tok : 0) * (uint32_t)(B3_I * 4) + 16u * (uint32_t)(e % W4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            // the saved O rows are read exactly once: non-temporal (same-box A/B in the step, round 4: attn_bwd3 L21 1290 -> 1267 us)
            v[it] = rat_ld4_stream(reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + off[it]));
        }
    }
    __device__ __forceinline__ void stash(float* tile, int ld) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = threadIdx.x + ATT_THREADS * it;
            if (e < ATT_ROWS * W4) {
                b3_zero_unless((valid >> it) & 1u, v[it]);
                *reinterpret_cast<float4*>(tile + (size_t)(e / W4) * ld + 4 * (e % W4)) = v[it];
            }
        }
    }
};

__device__ __forceinline__ void b3_load_piece(const float* src, const int64_t* rowtok, float4& v0, float4& v1, int d = B3_D, bool colok = true) {
    const int64_t tok = rowtok[threadIdx.x >> 3];
    v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tok >= 0 && colok) {
        v0 = *reinterpret_cast<const float4*>(src + tok * d + 8 * (threadIdx.x & 7));
        v1 = *reinterpret_cast<const float4*>(src + tok * d + 8 * (threadIdx.x & 7) + 4);
    }
}

// C[64][16 NT] = A (planes, row operand, KS K-steps) x B (weight fragments).  Wave w owns the row-tile pair {2 (w >> 2), +1} and the
// column tiles (w & 3) + 4 i: its A fragments are read once; the B fragment of the NEXT (column tile, K step) is requested before the
// MFMAs of the current one.  REV: column tiles are dealt from the other end ((3 - w & 3) + 4 i), so that two back-to-back phases with
// 4 k + 3 and 4 k + 1 column tiles (Q|K|V: 15, dO: 5) give every wave the same number of tiles in total.
// Same-box A/B of the alternatives (tools/ab_attn.sh, tools/experiments/): a whole column tile of B in flight: +5 % (registers);
// all four row tiles on one wave (half the L2 traffic, A re-read per column tile): +50 %; REV: -2 %.
template <int KS, bool REV = false, class PA, class BW, class Epi>
__device__ __forceinline__ void b3_gemm_rows(const PA& A, const BW& Bw, int n_tiles, const Epi& epi) {
    const int w = rat_wave(), mt0 = 2 * (w >> 2);
    int nt = REV ? 3 - (w & 3) : (w & 3);
    if (nt >= n_tiles) return;
    RatB3 b = Bw(nt, 0);
    RatB3 a[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < KS; ++s) a[i][s] = A.row_frag(mt0 + i, s);
    for (; nt < n_tiles; nt += 4) {
        f32x4 acc[2] = {rat_zero4(), rat_zero4()};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool last = s == KS - 1;
            const RatB3 bn = Bw(last ? (nt + 4 < n_tiles ? nt + 4 : nt) : nt, last ? 0 : s + 1);
            const RatB3 as[2] = {a[0][s], a[1][s]};
            rat_mfma3_block<2>(acc, as, b);
            b = bn;
        }
        epi(mt0, nt, acc[0]);
        epi(mt0 + 1, nt, acc[1]);
    }
}

// QSUB: RatSeqMap.queries < L is honoured (a separate instantiation: the ordinary one must not carry a second trip count)
// DPAD: embedding_dim 40 / 48 / 56 inside the 64-wide tiles (zero-padded weight planes; see b3_layer_norm_to_planes)
// ---- the attention-FORWARD core on the matrix pipe, exact fp32 (round 5; the backward's twin is b3_bwd_core_mfma) ----------------------
// Every (sequence, head) pair of the chunk is ONE wave's job on v_mfma_f32_16x16x4_f32, per 16-query tile:
//   S^T = K Q^T (A = K rows, B = Q rows; k = dim_head 10 -> 12, three steps): the accumulator of key tile jt holds, in lane (g, m),
//   S^T[key 16 jt + 4 g + r][query m] — a query's scores over ALL keys sit in the registers of the four lanes (g, m), so the row
//   softmax is in-register maxima / sums plus two cross-row swaps (b3m_rows_max / _sum); and the SAME registers are the B operand
//   of O^T = V^T P^T: k-step (jt, r) contracts over the keys {16 jt + 4 g + r : g} with A = V[that key][c = m] — the probabilities
//   never leave their registers (no LDS round trip, no shuffles, nothing split: this is what the bf16x3 core of attn_fwd3m_kernel
//   spent its VALU time on).  7 NIT MFMAs per query tile (NIT = 16-row tiles per sequence), ~40 VALU instructions of softmax.
// Dispatch by length like the backward (b3_fwd_matrix_core): sequences of 28 ... 32 tokens (BASELINE configs[4]: K = 30 -> L = 31),
// where the 32 x 32 tile is 94 % full; at L = 21 / 11 the VALU loop stays (profiles/round5/r5_attn_fwd_core_mfma_ab.txt).
__device__ __forceinline__ float b3m_rows_max(float v);
__device__ __forceinline__ float b3m_rows_sum(float v);
template <int NIT>
__device__ __forceinline__ void b3_fwd_core_mfma(float* qkv, float* lse_s, int L, int nsq, float scale) {
    const int w = rat_wave(), l = rat_lane(), g = l >> 4, m = l & 15;
    const float sl2 = scale * RAT_LOG2E;
    const int npairs = nsq * B3_H;
    for (int pair = w; pair < npairs; pair += ATT_WAVES) {
        const int h = pair % B3_H, sq = pair / B3_H;
        const int r0 = sq * L, cq = h * B3_DH, ck = B3_I + cq, cv = 2 * B3_I + cq;
        // per pair: K as the A operand of S^T (lane: K[key 16 jt + m][k 4 ks + g]) and V as the A operand of O^T (lane: V[key 16 jt + 4 g + r][c m]);
        // loads at their natural address (rows / columns past the operand stay inside the tile), masked by a select
        float ak[3][NIT], av[4][NIT];
#pragma unroll
        for (int jt = 0; jt < NIT; ++jt) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) ak[ks][jt] = qkv[(size_t)(r0 + 16 * jt + m) * B3_LDQ + ck + 4 * ks + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r][jt] = qkv[(size_t)(r0 + 16 * jt + 4 * g + r) * B3_LDQ + cv + m];
        }
#pragma unroll
        for (int jt = 0; jt < NIT; ++jt) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) ak[ks][jt] = (4 * ks + g < B3_DH && 16 * jt + m < L) ? ak[ks][jt] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r][jt] = (m < B3_DH && 16 * jt + 4 * g + r < L) ? av[r][jt] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i0 = 16 * it;
            if (i0 >= L) break;                                   // (wave-uniform)
            const bool qok = i0 + m < L;
            float bq[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) bq[ks] = qkv[(size_t)(r0 + i0 + m) * B3_LDQ + cq + 4 * ks + g];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) bq[ks] = (4 * ks + g < B3_DH && qok) ? bq[ks] : 0.f;
            f32x4 st[NIT];
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt) st[jt] = rat_zero4();
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int jt = 0; jt < NIT; ++jt) st[jt] = RAT_MFMA16(ak[ks][jt], bq[ks], st[jt]);
            // softmax over the keys of query column m: st[jt][r] = S^T[key 16 jt + 4 g + r][query i0 + m]
            float mx = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[jt][r] = 16 * jt + 4 * g + r < L ? st[jt][r] * sl2 : -INFINITY;
                    mx = fmaxf(mx, st[jt][r]);
                }
            mx = b3m_rows_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[jt][r] = rat_exp2(st[jt][r] - mx);          // (keys beyond L: exp2(-inf) = 0)
                    sum += st[jt][r];
                }
            sum = b3m_rows_sum(sum);
            // O^T[c][query] = sum over keys V[key][c] P^T[key][query]: the accumulators ARE the B operand, k-step (jt, r)
            f32x4 ot = rat_zero4();
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot = RAT_MFMA16(av[r][jt], st[jt][r], ot);
            const float inv = 1.0f / sum;
            if (qok) {                                            // ot[r] = O[query i0 + m][c = 4 g + r] (unnormalised); O replaces Q in place
                float* op = qkv + (size_t)(r0 + i0 + m) * B3_LDQ + cq + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < B3_DH) op[r] = ot[r] * inv;
                if (g == 0) lse_s[(r0 + i0 + m) * B3_H + h] = mx + rat_log2(sum);
            }
        }
    }
}

// GRP (wide heads, round 5): heads = a.groups x 8.  The head groups are independent given LayerNorm(x), so ONE launch loads and
// normalises a chunk once and then loops over the groups — Q|K|V projection, attention core, O -> planes / o_save, output projection
// per group, the projection's partial sums kept in the accumulator registers across the loop — and adds bias, Dropout and the residual
// once at the end: one LayerNorm / x load / y read-modify-write per chunk instead of one per group launch (rat_attn_fwd_groups).
// Group g's fragment planes are W.qkv / W.out + g x B3_GRP_PLANES; W_out's come from L2 (four groups' planes do not fit the LDS).
// MCF (1 / 2 = 16-row tiles per sequence): the attention core on the matrix pipe (b3_fwd_core_mfma) instead of the VALU loop; every position
// a query, sequences of at most 32 tokens
template <bool EX, bool QSUB = false, bool DPAD = false, bool GRP = false, int MCF = 0>
__global__ void __launch_bounds__(ATT_THREADS) attn_fwd3_kernel(AttnArgs a, Attn3W W) {
    static_assert(!GRP || (EX && !QSUB && !DPAD), "the group loop is written for the general (EX) form at embedding_dim 64");
    static_assert(MCF == 0 || !QSUB, "the matrix core computes every query");
    RAT_DYN_SMEM(smem);
    const PlanesX xp{smem};                                                 // LayerNorm(x) planes; later the fp32 output staging tile
    float* qkv = reinterpret_cast<float*>(smem + 3 * B3_XP);                // [64][244] fp32 Q|K|V; O overwrites Q
    const PlanesO op{smem + 3 * B3_XP + 64 * B3_LDQ * 4};                   // O planes (row operand of the output projection)
    int64_t* const rowtok0 = reinterpret_cast<int64_t*>(smem + 3 * B3_XP + 64 * B3_LDQ * 4 + 3 * B3_OP);
    float* ys = reinterpret_cast<float*>(smem);                             // [64][68] over the (then dead) x planes
    float* const lse_s = reinterpret_cast<float*>(smem + B3_FWD_LSE);       // the chunk's log-sum-exp, saved as whole rows below
    constexpr int LDY = B3_D + 4;
    const int L = a.L;
    // W_out's fragment planes: global -> LDS once per work-group (every chunk's output projection then reads them from LDS)
    if (!GRP)
        for (int e = threadIdx.x; e < (int)(B3_W_OUT / 16); e += ATT_THREADS)
            reinterpret_cast<rat_u4*>(smem + B3_FWD_WOUT)[e] = W.out.base[e];
    const RatWPlanesLds wout_lds{smem + B3_FWD_WOUT, 3};

    for (int e = threadIdx.x; e < 3 * B3_OP / 4; e += ATT_THREADS) reinterpret_cast<float*>(op.base)[e] = 0.f;   // incl. the slack
    for (int e = threadIdx.x; e < 64 * (B3_LDQ - B3_Q3); e += ATT_THREADS) qkv[(e >> 2) * B3_LDQ + B3_Q3 + (e & 3)] = 0.f;
    const int dreal = DPAD ? a.d : B3_D;
    const bool colok = !DPAD || 8 * (int)(threadIdx.x & 7) < dreal;
    float gam[8], bet[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        gam[k] = colok ? a.ln_g[8 * (threadIdx.x & 7) + k] : 0.f;
        bet[k] = colok ? a.ln_b[8 * (threadIdx.x & 7) + k] : 0.f;
    }
    {
        int nsq0, rows0;
        map_rows_b3<RAT_MAP_FWD>(a, blockIdx.x, rowtok0, nsq0, rows0);
    }
    __syncthreads();
    RAT_PROF_DECL
    int parity = 0;
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x, parity ^= 1) {
        const int64_t* rowtok = rowtok0 + parity * ATT_ROWS;
        int nsq, rows;
        {
            const int64_t q0 = chunk * a.nsq_chunk;
            const int64_t left = a.nseq - q0;
            nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
            rows = nsq * a.L;
        }
        float4 x0, x1;                                           // kept: the residual of the plain PreNorm(Attention)(x) + x layer
        b3_load_piece(a.x, rowtok, x0, x1, dreal, colok);
        b3_layer_norm_to_planes<DPAD>(rowtok[threadIdx.x >> 3] >= 0, x0, x1, a.eps, xp, gam, bet, nullptr, nullptr, dreal, colok);
        if (chunk + gridDim.x < a.nchunks) {
            int nsq1, rows1;
            map_rows_b3<RAT_MAP_FWD>(a, chunk + gridDim.x, (rowtok0 + (parity ^ 1) * ATT_ROWS), nsq1, rows1);
        }
        __syncthreads();
        RAT_PROF_MARK(0);
        float pf = 0.f;
        f32x4 yacc[2] = {rat_zero4(), rat_zero4()};              // GRP: this wave's two output-projection tiles, summed over the groups
        const int ngroups = GRP ? a.groups : 1;
        for (int grp = 0; grp < ngroups; ++grp) {                // (one trip unless GRP; the body keeps its indentation)
        const RatWPlanes wq = GRP ? RatWPlanes{W.qkv.base + (size_t)grp * (B3_GRP_PLANES / 16), W.qkv.steps} : W.qkv;
        const RatWPlanes wo = GRP ? RatWPlanes{W.out.base + (size_t)grp * (B3_GRP_PLANES / 16), W.out.steps} : W.out;
        float* const o_save = (GRP && a.o_save != nullptr) ? a.o_save + (int64_t)grp * a.group_tok * B3_I : a.o_save;
        float* const lse_save = (GRP && a.lse_save != nullptr) ? a.lse_save + (int64_t)grp * a.group_tok * B3_H : a.lse_save;
        // Q|K|V = LN(x) W_qkv^T
        b3_gemm_rows<2>(xp, wq, B3_Q3 / 16, [&](int mt, int nt, const f32x4& acc) {
            const int col = rat_acc_col(nt);
#pragma unroll
            for (int r = 0; r < 4; ++r) qkv[(size_t)rat_acc_row(mt, r) * B3_LDQ + col] = acc[r];
        });
        __syncthreads();
        RAT_PROF_MARK(1);
        // softmax(Q K^T * scale) V on the VALU — identical to attn_fwd_kernel<64, 10>.  (A two-stage form for L <= 24 — the row of scores
        //  kept in registers, max first, then ONE exponential and a plain packed axpy per key instead of the online rescaling: 110
        //  instead of 180 VALU cycles per pair — measured 4-10 % SLOWER, one key or three keys per trip alike; 5 / 6 / 7 keys per trip
        //  instead of 3: no change; three queries per lane on a third of the keys (a third of the LDS bytes per pair, partial softmax
        //  states merged by lane shuffles): 9-17 % slower.  tools/ab_attn.sh.  Round 3: two queries per lane over ALL keys (half the LDS
        //  bytes per pair, bit-identical): +9.5 % / +4 % at L = 21 / 11; softmax against the Cauchy-Schwarz bound |q| max|k| (no running
        //  maximum, no rescaling, independent keys): +-0 / +3 % — tools/experiments/attn_fwd3_core_variants.hip.txt.)
        if ((int)threadIdx.x < ATT_ROWS * 2 && chunk + gridDim.x < a.nchunks && (!GRP || grp == ngroups - 1))   // (no prefetch: +2-3 %, same-box A/B)
            pf = prefetch_lines_map((rowtok0 + (parity ^ 1) * ATT_ROWS), threadIdx.x, 2, a.x, dreal);
        typedef HeadVec<B3_DH> HV;
        const int nq = QSUB ? a.nq : L;                          // queries that matter per sequence (RatSeqMap.queries; normally L)
        const int ntasks = MCF ? 0 : nsq * B3_H * nq;
        const float sl2 = a.scale * RAT_LOG2E;
        if (MCF) b3_fwd_core_mfma<(MCF > 0 ? MCF : 1)>(qkv, lse_s, L, nsq, a.scale);
        for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
            const int i = task % nq;
            const int h = (task / nq) % B3_H;
            const int sq = task / (nq * B3_H);
            const int row_i = sq * L + i;
            float* qp = qkv + (size_t)row_i * B3_LDQ + h * B3_DH;
            HV q, o, kv;
            q.load(qp, B3_DH);
            o.zero();
            float m = -INFINITY, l = 0.f;
            const float* kbase = qkv + (size_t)(sq * L) * B3_LDQ + B3_I + h * B3_DH;
            int j = 0;
            for (; j + CORE_UNROLL <= L; j += CORE_UNROLL) {
                HV kk[CORE_UNROLL], vv[CORE_UNROLL];
#pragma unroll
                for (int u = 0; u < CORE_UNROLL; ++u) {
                    const float* kp = kbase + (size_t)(j + u) * B3_LDQ;
                    kk[u].load(kp, B3_DH);
                    vv[u].load(kp + B3_I, B3_DH);
                }
                float sc[CORE_UNROLL];
#pragma unroll
                for (int u = 0; u < CORE_UNROLL; ++u) sc[u] = q.dot(kk[u]) * sl2;
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
                const float* kp = kbase + (size_t)j * B3_LDQ;
                kv.load(kp, B3_DH);
                const float sv = q.dot(kv) * sl2;
                const float mn = fmaxf(m, sv);
                const float corr = rat_exp2(m - mn);
                const float p = rat_exp2(sv - mn);
                l = l * corr + p;
                kv.load(kp + B3_I, B3_DH);
                o.scale_axpy(corr, p, kv);
                m = mn;
            }
            const float inv = 1.0f / l;
            o.store(qp, B3_DH, inv);
            lse_s[row_i * B3_H + h] = m + rat_log2(l);
        }
        if (QSUB && nq < L) {                                    // the positions nobody asked for: O = 0, lse = 0 (defined, never used)
            for (int e = threadIdx.x; e < rows * B3_H; e += ATT_THREADS) {
                const int r = e / B3_H, h = e - r * B3_H;
                if (r % L < nq) continue;
                HV z;
                z.zero();
                z.store(qkv + (size_t)r * B3_LDQ + h * B3_DH, B3_DH, 1.0f);
                lse_s[e] = 0.f;
            }
        }
        __syncthreads();
        RAT_PROF_MARK(2);
        // O (the Q columns of the valid rows; padding rows are exact zeros) -> planes, and -> o_save for the backward: whole 320-byte
        // rows in 16-byte pieces with the non-temporal hint (round 4; before, every core lane stored its head's 40 bytes in five
        // scattered 8-byte stores at the end of its key loop).  lse_save leaves the same way, from the LDS copy.
#ifndef RAT_O_HALF                                                // 640 whole pieces on 512 threads (2 trips, the second a quarter full)
        for (int e = threadIdx.x; e < ATT_ROWS * (B3_I / 8); e += ATT_THREADS) {
            const int r = e / (B3_I / 8), o8 = e - r * (B3_I / 8);
            const float* src = qkv + (size_t)r * B3_LDQ + 8 * o8;
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            rat_u4 h, m, l;
            rat_split8(v0, v1, h, m, l);
            op.store(r, o8, h, m, l);
            const int64_t tok = rowtok[r];
            if (o_save != nullptr && tok >= 0) {
                rat_st4_stream(o_save + tok * B3_I + 8 * o8, v0);
                rat_st4_stream(o_save + tok * B3_I + 8 * o8 + 4, v1);
            }
        }
#else
        // A/B only (-DRAT_O_HALF): HALF pieces (4 columns), 1280 on 512 threads = 3 trips of half the work — measured no faster (L 21 equal,
        // L 11 +0.7 %, profiles/round5/r5_ohalf_ab.txt): the pass is not bound by the busiest thread's work
        for (int e = threadIdx.x; e < ATT_ROWS * (B3_I / 4); e += ATT_THREADS) {
            const int r = e / (B3_I / 4), q4 = e - r * (B3_I / 4);
            const float4 v = *reinterpret_cast<const float4*>(qkv + (size_t)r * B3_LDQ + 4 * q4);
            unsigned h0, h1, m0, m1, l0, l1;
            rat_split2(v.x, v.y, h0, m0, l0);
            rat_split2(v.z, v.w, h1, m1, l1);
            op.store_half(r, q4, h0, h1, m0, m1, l0, l1);
            const int64_t tok = rowtok[r];
            if (o_save != nullptr && tok >= 0) rat_st4_stream(o_save + tok * B3_I + 4 * q4, v);
        }
#endif
        if (lse_save != nullptr && (int)threadIdx.x < 2 * ATT_ROWS) {
            const int r = threadIdx.x >> 1, part = threadIdx.x & 1;
            const int64_t tok = rowtok[r];
            if (tok >= 0) rat_st4_stream(lse_save + tok * B3_H + 4 * part, *reinterpret_cast<const float4*>(lse_s + r * B3_H + 4 * part));
        }
        __syncthreads();
        RAT_PROF_MARK(3);
        // y = O W_out^T + b_out (+ residual), staged through LDS for whole-row stores
        if (GRP) {                                               // this group's partial projection stays in the accumulators
            b3_gemm_rows<3>(op, wo, B3_D / 16, [&](int mt, int nt, const f32x4& acc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) yacc[mt & 1][r] += acc[r];       // (mt = 2 (wave >> 2) + {0, 1})
            });
            continue;                                            // (O -> planes of the next group waits behind two barriers: no third one here)
        }
#ifdef RAT_FWD_WOUT_L2                                           // (A/B knob: round 3's form, fragments from L2)
        b3_gemm_rows<3>(op, W.out, B3_D / 16, [&](int mt, int nt, const f32x4& acc) {
#else
        b3_gemm_rows<3>(op, wout_lds, B3_D / 16, [&](int mt, int nt, const f32x4& acc) {
#endif
            const int col = rat_acc_col(nt);
            const float bias = (!DPAD || col < dreal) ? a.b_out[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) ys[(size_t)rat_acc_row(mt, r) * LDY + col] = acc[r] + bias;
        });
        }                                                        // (group loop)
        if (GRP) {                                               // the x planes under `ys` were last read two barriers ago (last group's Q|K|V)
            const int w = rat_wave(), col = rat_acc_col(w & 3);
            const float bias = a.b_out[col];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) ys[(size_t)rat_acc_row(2 * (w >> 2) + i, r) * LDY + col] = yacc[i][r] + bias;
        }
        __syncthreads();
        RAT_PROF_MARK(4);
        if (EX) {
            store_rows_residual(a.y, ys, LDY, a.res, rowtok, rows, dreal, true, a.out_scale, &a.drop);
        } else {                                                 // y = tile + x, the x piece still in registers: no global re-read
            const int r = threadIdx.x >> 3, sb = threadIdx.x & 7;
            const int64_t tok = rowtok[r];
            if (tok >= 0 && colok) {
                const float4 t0 = *reinterpret_cast<const float4*>(ys + (size_t)r * LDY + 8 * sb);
                const float4 t1 = *reinterpret_cast<const float4*>(ys + (size_t)r * LDY + 8 * sb + 4);
                *reinterpret_cast<float4*>(a.y + tok * dreal + 8 * sb) = make_float4(t0.x + x0.x, t0.y + x0.y, t0.z + x0.z, t0.w + x0.w);
                *reinterpret_cast<float4*>(a.y + tok * dreal + 8 * sb + 4) = make_float4(t1.x + x1.x, t1.y + x1.y, t1.z + x1.z, t1.w + x1.w);
            }
        }
        __syncthreads();
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));
#endif
        RAT_PROF_MARK(5);
    }
    RAT_PROF_FLUSH(a.prof, 48);
}

// ---- forward, bf16x3, attention core on the MATRIX pipe as well (round 3) ------------------------------------------------------
// softmax(Q K^T * scale) V per (sequence, head) as 16 x 16 x 32 bf16 MFMAs with 3-way split operands — the same fp32-class
// arithmetic as the projections (every bf16 x bf16 product exact, fp32 accumulation) instead of one VALU lane per (sequence,
// head, query) walking all keys (41.6 % of attn_fwd3_kernel's time, r2_phase_shares.txt).  What makes the tiny per-head
// products (L x 10 x L, L = 11 / 21) fit the 16 x 16 x 32 instruction without wasting its K dimension:
//   * S^T = K Q^T: the contraction is over dim_head = 10.  The 32 k-slots of ONE instruction hold all three bf16 chunks of a
//     (row, head) vector: lane group g < 3 carries chunk g of elements 0..7 (a 16-byte PIECE), lane group 3 the piece
//     [h8 h9 m8 m9 l8 l9 0 0].  A = K rows, B = Q rows gives the "diagonal" products kh qh + km qm + kl ql; reading A with its
//     chunks ROTATED by one and by two (lane group g takes piece (g + r) mod 3; the fourth piece is stored in its three
//     rotations) gives the six cross products — 3 instructions for all nine products of the split (the bf16x3 GEMMs keep six),
//     94 % of the K dimension used, every operand fetch a 16-byte LDS read.
//   * the accumulator of S^T (lane = query column, registers = 4 consecutive keys) IS the B operand layout of the next product
//     O^T = V^T P^T (contraction over keys: k-slot 8 g + t <-> key 4 g + (t & 3), chunk pair t >> 2), so the probabilities never
//     leave their registers: scale, mask, softmax, split into three bf16 chunks, three MFMAs against A = V^T read from
//     per-(sequence, head) TRANSPOSED planes [c][key] (8-byte reads).  Six products: ph vh, ph vm, pm vh, pm vm, ph vl, pl vh.
//   * the Q|K|V projection runs TRANSPOSED (weights as the A operand, tokens on the lanes: the fragment registers are the same,
//     only the roles in the instruction swap), so that a lane's accumulator quad is 4 consecutive Q|K|V columns of ONE token: Q and K
//     leave as 32-bit stores of element pairs straight into the piece layout, V as 2-byte stores into the transposed planes
//     (consecutive lanes = consecutive keys: conflict-free).  There is no fp32 Q|K|V tile any more.
// A wave works on (sequence, head, query tile) units: 40 pairs (L = 11) / 24 pairs x 2 query tiles (L = 21) per 64-row chunk.
// O goes to an fp32 LDS tile (over the dead LayerNorm planes), from which the unchanged tail takes over (O -> planes, output
// projection); o_save leaves as whole 320-byte rows from that tile.
// LDS map (bytes): [0, 24576) LN planes -> fp32 O tile [64][84] -> fp32 y tile [64][68] | Q pieces 33792 -> O planes |
// K pieces 50176 | V^T planes <= 51840 | row maps.  Q rows are 528 bytes (8 heads x 64 + 16), K rows 784 (8 x 96 + 16) apart:
// the 16 rows of a fragment read fall on distinct banks; V^T rows are 2 KP + 8 bytes apart.
constexpr int B3M_QROW = B3_H * 64 + 16;                // bytes per token row of the Q region
constexpr int B3M_KROW = B3_H * 96 + 16;                // ... of the K region (the fourth piece in its three rotations)
constexpr int B3M_Q = 64 * B3M_QROW;                    // 33792
constexpr int B3M_K = 64 * B3M_KROW;                    // 50176
constexpr int B3M_V = 51840;                            // V^T planes: nsq_chunk * 8 heads * 3 planes * 10 rows of (2 KP + 8) bytes, KP = 16 ceil(L / 16)
constexpr int B3M_LDO = B3_I + 4;                       // fp32 O tile row (floats)
constexpr size_t B3M_OFF_Q = (size_t)3 * B3_XP, B3M_OFF_K = B3M_OFF_Q + B3M_Q, B3M_OFF_V = B3M_OFF_K + B3M_K,
                 B3M_OFF_MAP = B3M_OFF_V + B3M_V;
constexpr size_t b3m_fwd_smem() { return B3M_OFF_MAP + 2 * 64 * 8 + 64 * 4; }
static_assert((size_t)64 * B3M_LDO * 4 <= (size_t)3 * B3_XP, "the fp32 O tile overlays the LayerNorm planes");
static_assert((size_t)3 * B3_OP <= (size_t)B3M_Q, "the O planes overlay the Q pieces");
static_assert(b3m_fwd_smem() <= 160 * 1024, "LDS budget");
// does the V^T region hold a chunk's sequences at this length?
static bool b3m_fits(int L, int nsq_chunk) {
    return L >= 1 && L <= 64 && (size_t)nsq_chunk * B3_H * 30 * (2 * 16 * ((L + 15) / 16) + 8) <= (size_t)B3M_V;
}

// upper 16 bits of the three chunks of x (x = h + m + l exactly, rat_split2's truncation split)
__device__ __forceinline__ void b3m_split1(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned xb = rat_fbits(x);
    const float r1 = x - rat_bitsf(xb & 0xffff0000u);
    const unsigned rb = rat_fbits(r1);
    const float r2 = r1 - rat_bitsf(rb & 0xffff0000u);
    h = (unsigned short)(xb >> 16);
    m = (unsigned short)(rb >> 16);
    l = (unsigned short)(rat_fbits(r2) >> 16);
}
__device__ __forceinline__ unsigned b3m_pack(unsigned short lo, unsigned short hi) { return (unsigned)lo | ((unsigned)hi << 16); }

// max / sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the lane groups of one accumulator column).  gfx950: two row-swap
// instructions (v_permlane16_swap: row 1 <-> row 0 and row 3 <-> row 2 of the two operands; v_permlane32_swap: upper half <->
// lower half) instead of two trips through the LDS crossbar (ds_bpermute); same pairing order as the shuffle form.
__device__ __forceinline__ float b3m_rows_max(float v) {
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
__device__ __forceinline__ float b3m_rows_sum(float v) {
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

struct B3mCore {
    const char* qv;
    const char* kv;
    const char* vt;
    float* os;
    float* lse_save;
    const int64_t* rowtok;
    int L, VROW, VPL, VB;        // V^T: bytes per c row / per plane / per (sequence, head) block
    float sl2;
    int nsq;
};

// NU independent units = (sequence, head, query tile) at once, all KB key blocks of a unit in registers.  Phases, each over all units:
// every operand fetch of S^T (16-byte reads) -> every S^T block (3 MFMAs each, independent chains) -> ONE softmax over a unit's keys (no
// online rescaling) with the V^T fetches in flight -> the 3 KB MFMAs of O^T -> stores.  The units share nothing, so within a phase the
// hardware has NU * KB independent chains to interleave (a single unit is one long dependent chain: LDS read -> 3 MFMAs -> cross-lane
// max -> exp2 -> split -> 3 MFMAs).
template <int KB, int NU>
__device__ __forceinline__ void b3m_units(const B3mCore& c, int u0) {
    const int lane = rat_lane(), g = lane >> 4, n16 = lane & 15, L = c.L;
    int sq[NU], hd[NU], qi[NU];
    rat_u4 qf[NU], kf[NU][KB][3];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        const int u = u0 + n, task = rat_wave() + ATT_WAVES * (u / KB), qt = u % KB;
        sq[n] = task >> 3;
        hd[n] = task & 7;
        qi[n] = 16 * qt + n16;
        const int qrow = sq[n] * L + (qi[n] < L ? qi[n] : L - 1);
        qf[n] = *reinterpret_cast<const rat_u4*>(c.qv + qrow * B3M_QROW + hd[n] * 64 + 16 * g);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int kj = 16 * kb + n16;
            const char* kr = c.kv + (sq[n] * L + (kj < L ? kj : L - 1)) * B3M_KROW + hd[n] * 96;
            // rotation r: lane group g < 3 reads piece (g + r) mod 3, lane group 3 the r-th rotation of the fourth piece
#pragma unroll
            for (int r = 0; r < 3; ++r)
                kf[n][kb][r] = *reinterpret_cast<const rat_u4*>(kr + (g == 3 ? 48 + 16 * r : 16 * ((g + r) % 3)));
        }
    }
    RAT_SCHED_FENCE();
    f32x4 st[NU][KB];
#pragma unroll
    for (int n = 0; n < NU; ++n)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 t = RAT_MFMA_BF16(rat_as_bf16x8(kf[n][kb][2]), rat_as_bf16x8(qf[n]), rat_zero4());
            t = RAT_MFMA_BF16(rat_as_bf16x8(kf[n][kb][1]), rat_as_bf16x8(qf[n]), t);
            st[n][kb] = RAT_MFMA_BF16(rat_as_bf16x8(kf[n][kb][0]), rat_as_bf16x8(qf[n]), t);
        }
    // V^T fragments (keys 16 kb + 4 g .. + 3 of row c in each plane): requested now, consumed after the softmax
    uint2 vf[NU][KB][3];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        const char* vblk = c.vt + (sq[n] * B3_H + hd[n]) * c.VB + (n16 < 10 ? n16 : 9) * c.VROW + 8 * g;     // this lane's c row of V^T
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int p = 0; p < 3; ++p) vf[n][kb][p] = *reinterpret_cast<const uint2*>(vblk + 32 * kb + p * c.VPL);
    }
    float pr[NU][KB][4], lt[NU], mx[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[n][kb][r] = (16 * kb + 4 * g + r) < L ? st[n][kb][r] * c.sl2 : -INFINITY;
                m = fmaxf(m, pr[n][kb][r]);
            }
        mx[n] = b3m_rows_max(m);
    }
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[n][kb][r] = rat_exp2(pr[n][kb][r] - mx[n]);
                sum += pr[n][kb][r];
            }
        lt[n] = b3m_rows_sum(sum);
    }
    f32x4 ot[NU];
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        ot[n] = rat_zero4();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            unsigned short ph[4], pm[4], pl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b3m_split1(pr[n][kb][r], ph[r], pm[r], pl[r]);
            const unsigned h01 = b3m_pack(ph[0], ph[1]), h23 = b3m_pack(ph[2], ph[3]);
            const unsigned m01 = b3m_pack(pm[0], pm[1]), m23 = b3m_pack(pm[2], pm[3]);
            const unsigned l01 = b3m_pack(pl[0], pl[1]), l23 = b3m_pack(pl[2], pl[3]);
            const uint2 vh = vf[n][kb][0], vm = vf[n][kb][1], vl = vf[n][kb][2];
            rat_u4 a_hm, a_lh, b_hh, b_mm, b_hl;
            a_hm.x = vh.x; a_hm.y = vh.y; a_hm.z = vm.x; a_hm.w = vm.y;
            a_lh.x = vl.x; a_lh.y = vl.y; a_lh.z = vh.x; a_lh.w = vh.y;
            b_hh.x = h01; b_hh.y = h23; b_hh.z = h01; b_hh.w = h23;
            b_mm.x = m01; b_mm.y = m23; b_mm.z = m01; b_mm.w = m23;
            b_hl.x = h01; b_hl.y = h23; b_hl.z = l01; b_hl.w = l23;
            ot[n] = RAT_MFMA_BF16(rat_as_bf16x8(a_lh), rat_as_bf16x8(b_hl), ot[n]);      // vl ph + vh pl
            ot[n] = RAT_MFMA_BF16(rat_as_bf16x8(a_hm), rat_as_bf16x8(b_mm), ot[n]);      // vh pm + vm pm
            ot[n] = RAT_MFMA_BF16(rat_as_bf16x8(a_hm), rat_as_bf16x8(b_hh), ot[n]);      // vh ph + vm ph
        }
    }
#pragma unroll
    for (int n = 0; n < NU; ++n) {
        const float inv = 1.0f / lt[n];
        if (qi[n] < L) {                                         // O^T rows 4 g + r = dim_head index c (10 of 16 used), column = query
            const int R = sq[n] * L + qi[n];
            float* dst = c.os + (size_t)R * B3M_LDO + hd[n] * B3_DH + 4 * g;
            if (g < 2) {
                *reinterpret_cast<float2*>(dst) = make_float2(ot[n][0] * inv, ot[n][1] * inv);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(ot[n][2] * inv, ot[n][3] * inv);
            } else if (g == 2) {
                *reinterpret_cast<float2*>(dst) = make_float2(ot[n][0] * inv, ot[n][1] * inv);
            } else if (c.lse_save != nullptr) {                  // (the otherwise idle lane group stores the log-sum-exp)
                c.lse_save[c.rowtok[R] * B3_H + hd[n]] = mx[n] + rat_log2(lt[n]);
            }
        }
    }
}

template <int KB>
__device__ __forceinline__ void b3m_core(const B3mCore& c) {
    constexpr int NU = KB == 1 ? 3 : 2;
    const int w = rat_wave();
    const int ntask = c.nsq * B3_H > w ? (c.nsq * B3_H - w + ATT_WAVES - 1) / ATT_WAVES : 0;       // this wave's (sequence, head) pairs
    const int nu = ntask * KB;
    int u = 0;
    for (; u + NU <= nu; u += NU) b3m_units<KB, NU>(c, u);
    if (NU > 2 && u + 2 <= nu) {
        b3m_units<KB, 2>(c, u);
        u += 2;
    }
    for (; u < nu; ++u) b3m_units<KB, 1>(c, u);
}

// b3_gemm_rows with the operand roles swapped: C^T[16 NT columns][64 rows] — the weight fragment is the A operand, the activation
// fragment the B operand (the registers are the same: lane l holds k-slots 8 (l >> 4).. of row / column l & 15 either way).  The
// epilogue gets acc[r] = C[row 16 mt + (lane & 15)][column 16 nt + 4 (lane >> 4) + r]: four consecutive columns of one token per lane.
template <int KS, class PA, class Epi>
__device__ __forceinline__ void b3_gemm_rows_t(const PA& A, const RatWPlanes& Bw, int n_tiles, const Epi& epi) {
    const int w = rat_wave(), mt0 = 2 * (w >> 2);
    int nt = w & 3;
    if (nt >= n_tiles) return;
    RatB3 b = Bw(nt, 0);
    RatB3 a[2][KS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < KS; ++s) a[i][s] = A.row_frag(mt0 + i, s);
    for (; nt < n_tiles; nt += 4) {
        f32x4 acc[2] = {rat_zero4(), rat_zero4()};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool last = s == KS - 1;
            const RatB3 bn = Bw(last ? (nt + 4 < n_tiles ? nt + 4 : nt) : nt, last ? 0 : s + 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = RAT_MFMA_BF16(b.l, a[i][s].h, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = RAT_MFMA_BF16(b.h, a[i][s].l, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = RAT_MFMA_BF16(b.m, a[i][s].m, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = RAT_MFMA_BF16(b.m, a[i][s].h, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = RAT_MFMA_BF16(b.h, a[i][s].m, acc[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = RAT_MFMA_BF16(b.h, a[i][s].h, acc[i]);
            b = bn;
        }
        epi(mt0, nt, acc[0]);
        epi(mt0 + 1, nt, acc[1]);
    }
}

template <bool EX>
__global__ void __launch_bounds__(ATT_THREADS) attn_fwd3m_kernel(AttnArgs a, Attn3W W) {
    RAT_DYN_SMEM(smem);
    const PlanesX xp{smem};                                                 // LayerNorm(x) planes
    float* const os = reinterpret_cast<float*>(smem);                       // fp32 O tile [64][84] (after the projection)
    float* const ys = reinterpret_cast<float*>(smem);                       // fp32 y tile [64][68] (after O -> planes)
    char* const qv = smem + B3M_OFF_Q;
    char* const kv = smem + B3M_OFF_K;
    char* const vt = smem + B3M_OFF_V;
    const PlanesO op{smem + B3M_OFF_Q};                                     // O planes over the (then dead) Q pieces
    int64_t* const rowtok0 = reinterpret_cast<int64_t*>(smem + B3M_OFF_MAP);
    int* const rowmap = reinterpret_cast<int*>(smem + B3M_OFF_MAP + 2 * 64 * 8);    // row -> (sequence slot << 8) | position
    constexpr int LDY = B3_D + 4;
    const int L = a.L;
    const int KB = (L + 15) >> 4;                                          // key blocks per sequence
    const int VROW = 2 * 16 * KB + 8, VPL = 10 * VROW, VB = 3 * VPL;       // V^T: bytes per c row / plane / (sequence, head) block

    // every byte the MFMAs may read must hold a finite bf16 (pads and not-yet-written rows included): zero the operand regions once
    for (int e = threadIdx.x; e < (int)((B3M_OFF_MAP - B3M_OFF_Q) / 16); e += ATT_THREADS)
        reinterpret_cast<float4*>(smem + B3M_OFF_Q)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < 64) rowmap[threadIdx.x] = ((threadIdx.x / L) << 8) | (threadIdx.x % L);
    float gam[8], bet[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        gam[k] = a.ln_g[8 * (threadIdx.x & 7) + k];
        bet[k] = a.ln_b[8 * (threadIdx.x & 7) + k];
    }
    {
        int nsq0, rows0;
        map_rows_b3<RAT_MAP_FWD>(a, blockIdx.x, rowtok0, nsq0, rows0);
    }
    __syncthreads();
    RAT_PROF_DECL
    const int lane = rat_lane(), g = lane >> 4, n16 = lane & 15;
    const float sl2 = a.scale * RAT_LOG2E;
    int parity = 0;
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x, parity ^= 1) {
        const int64_t* rowtok = rowtok0 + parity * ATT_ROWS;
        int nsq, rows;
        {
            const int64_t q0 = chunk * a.nsq_chunk;
            const int64_t left = a.nseq - q0;
            nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
            rows = nsq * a.L;
        }
        float4 x0, x1;                                           // kept: the residual of the plain PreNorm(Attention)(x) + x layer
        b3_load_piece(a.x, rowtok, x0, x1);
        b3_layer_norm_to_planes(rowtok[threadIdx.x >> 3] >= 0, x0, x1, a.eps, xp, gam, bet, nullptr, nullptr);
        if (chunk + gridDim.x < a.nchunks) {
            int nsq1, rows1;
            map_rows_b3<RAT_MAP_FWD>(a, chunk + gridDim.x, (rowtok0 + (parity ^ 1) * ATT_ROWS), nsq1, rows1);
        }
        __syncthreads();
        RAT_PROF_MARK(0);
        // (Q|K|V)^T = W_qkv LN(x)^T: tokens on the lanes, written straight into the core's operand layouts
        b3_gemm_rows_t<2>(xp, W.qkv, B3_Q3 / 16, [&](int mt, int nt, const f32x4& acc) {
            const int which = nt / 5;                            // 0 Q, 1 K, 2 V: uniform per column tile (80 = 5 x 16)
            const int R = 16 * mt + n16;                         // this lane's token row
            const int c0 = 16 * nt + 4 * g - 80 * which;         // first of its four columns inside Q / K / V (even)
            unsigned short h[4], m[4], l[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b3m_split1(acc[r], h[r], m[r], l[r]);
            if (which < 2) {
#pragma unroll
                for (int pr2 = 0; pr2 < 2; ++pr2) {              // element pairs (c, c + 1), c even: never across a head
                    const int cc = c0 + 2 * pr2, head = cc / 10, c = cc - 10 * head;
                    const unsigned dh = b3m_pack(h[2 * pr2], h[2 * pr2 + 1]), dm = b3m_pack(m[2 * pr2], m[2 * pr2 + 1]),
                                   dl = b3m_pack(l[2 * pr2], l[2 * pr2 + 1]);
                    const bool tail = c == 8;                    // the fourth piece [h8 h9 m8 m9 l8 l9 0 0]
                    char* p = (which ? kv + R * B3M_KROW + head * 96 : qv + R * B3M_QROW + head * 64);
                    const int step = tail ? 4 : 16;
                    char* q = p + (tail ? 48 : 2 * c);
                    *reinterpret_cast<unsigned*>(q) = dh;
                    *reinterpret_cast<unsigned*>(q + step) = dm;
                    *reinterpret_cast<unsigned*>(q + 2 * step) = dl;
                    if (which == 1 && tail) {                    // K: the fourth piece rotated by one ([m l h]) and by two ([l h m])
                        *reinterpret_cast<unsigned*>(p + 64) = dm;
                        *reinterpret_cast<unsigned*>(p + 68) = dl;
                        *reinterpret_cast<unsigned*>(p + 72) = dh;
                        *reinterpret_cast<unsigned*>(p + 80) = dl;
                        *reinterpret_cast<unsigned*>(p + 84) = dh;
                        *reinterpret_cast<unsigned*>(p + 88) = dm;
                    }
                }
            } else if (R < rows) {
                const int rm = rowmap[R];
                char* vb = vt + (rm >> 8) * B3_H * VB + (rm & 255) * 2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cc = c0 + r, head = cc / 10, c = cc - 10 * head;
                    char* p = vb + head * VB + c * VROW;
                    *reinterpret_cast<unsigned short*>(p) = h[r];
                    *reinterpret_cast<unsigned short*>(p + VPL) = m[r];
                    *reinterpret_cast<unsigned short*>(p + 2 * VPL) = l[r];
                }
            }
        });
        __syncthreads();
        RAT_PROF_MARK(1);
        float pf = 0.f;
        if ((int)threadIdx.x < ATT_ROWS * 2 && chunk + gridDim.x < a.nchunks)
            pf = prefetch_lines_map((rowtok0 + (parity ^ 1) * ATT_ROWS), threadIdx.x, 2, a.x, B3_D);
        // ---- the core: (sequence, head) pairs dealt to the waves; a wave works on NU independent (pair, query tile) units at a time
        {
            const B3mCore cc{qv, kv, vt, os, a.lse_save, rowtok, L, VROW, VPL, VB, sl2, nsq};
            switch (KB) {
                case 1: b3m_core<1>(cc); break;
                case 2: b3m_core<2>(cc); break;
                case 3: b3m_core<3>(cc); break;
                default: b3m_core<4>(cc); break;
            }
        }
        __syncthreads();
        RAT_PROF_MARK(2);
        // O tile -> planes (row operand of the output projection; padding rows are exact zeros) and, as whole rows, -> o_save
        for (int e = threadIdx.x; e < ATT_ROWS * (B3_I / 8); e += ATT_THREADS) {
            const int r = e / (B3_I / 8), o8 = e - r * (B3_I / 8);
            const float* src = os + (size_t)r * B3M_LDO + 8 * o8;
            float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            const int64_t tok = rowtok[r];
            b3_zero_unless(tok >= 0, v0);
            b3_zero_unless(tok >= 0, v1);
            if (tok >= 0 && a.o_save != nullptr) {
                *reinterpret_cast<float4*>(a.o_save + tok * B3_I + 8 * o8) = v0;
                *reinterpret_cast<float4*>(a.o_save + tok * B3_I + 8 * o8 + 4) = v1;
            }
            rat_u4 h, m, l;
            rat_split8(v0, v1, h, m, l);
            op.store(r, o8, h, m, l);
        }
        __syncthreads();
        RAT_PROF_MARK(3);
        // y = O W_out^T + b_out (+ residual), staged through LDS for whole-row stores
        b3_gemm_rows<3>(op, W.out, B3_D / 16, [&](int mt, int nt, const f32x4& acc) {
            const int col = rat_acc_col(nt);
            const float bias = a.b_out[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) ys[(size_t)rat_acc_row(mt, r) * LDY + col] = acc[r] + bias;
        });
        __syncthreads();
        RAT_PROF_MARK(4);
        if (EX) {
            store_rows_residual(a.y, ys, LDY, a.res, rowtok, rows, B3_D, true, a.out_scale, &a.drop);
        } else {                                                 // y = tile + x, the x piece still in registers: no global re-read
            const int r = threadIdx.x >> 3, sb = threadIdx.x & 7;
            const int64_t tok = rowtok[r];
            if (tok >= 0) {
                const float4 t0 = *reinterpret_cast<const float4*>(ys + (size_t)r * LDY + 8 * sb);
                const float4 t1 = *reinterpret_cast<const float4*>(ys + (size_t)r * LDY + 8 * sb + 4);
                *reinterpret_cast<float4*>(a.y + tok * B3_D + 8 * sb) = make_float4(t0.x + x0.x, t0.y + x0.y, t0.z + x0.z, t0.w + x0.w);
                *reinterpret_cast<float4*>(a.y + tok * B3_D + 8 * sb + 4) = make_float4(t1.x + x1.x, t1.y + x1.y, t1.z + x1.z, t1.w + x1.w);
            }
        }
        __syncthreads();
#ifndef RAT_EMU
        asm volatile("" ::"v"(pf));
#endif
        RAT_PROF_MARK(5);
    }
    RAT_PROF_FLUSH(a.prof, 48);
}

// ---- backward, bf16x3.  LDS map (bytes): [x planes 24576][dy planes 24576][Q|K|V fp32 62464][O fp32 21504][dO fp32 21504][misc];
// the three fp32 tiles are contiguous: once the attention core is done, d(Q|K|V) is re-written over them as planes (3 x 30720),
// and the dy planes (dead after dO / dW_out) become the fp32 tile of d(LayerNorm out).
constexpr int B3_LDT = B3_I + 4;                        // fp32 O / dO tile row (floats)
constexpr int B3_LDN = B3_D + 4;                        // fp32 d(LN out) tile row
constexpr size_t B3_OFF_DYP = (size_t)3 * B3_XP, B3_OFF_QKV = 2 * B3_OFF_DYP, B3_OFF_OB = B3_OFF_QKV + (size_t)64 * B3_LDQ * 4,
                 B3_OFF_DOB = B3_OFF_OB + (size_t)64 * B3_LDT * 4, B3_OFF_MISC = B3_OFF_DOB + (size_t)64 * B3_LDT * 4;
constexpr size_t b3_bwd_smem() { return B3_OFF_MISC + (size_t)64 * (2 + 2 * B3_H) * 4 + 2 * 64 * 8 + 2 * B3_D * 4; }
static_assert(B3_OFF_MISC - B3_OFF_QKV >= (size_t)3 * B3_QP + 64, "d(Q|K|V) planes overlay the three fp32 tiles");
static_assert((size_t)64 * B3_LDN * 4 <= (size_t)3 * B3_XP, "d(LN out) overlays the dy planes");

// C[64][64] = A (planes, row operand, KS K-steps) x B: wave w owns row tiles {2 (w >> 2), +1} x column tile (w & 3); the operands
// of step s + 1 (weight fragment from L2, A fragments from LDS) are requested before the MFMAs of step s
template <int KS, class PA, class Epi>
__device__ __forceinline__ void b3_gemm_rows_longk(const PA& A, const RatWPlanes& Bw, const Epi& epi) {
    const int w = rat_wave(), mt0 = 2 * (w >> 2), nt = w & 3;
    f32x4 acc[2] = {rat_zero4(), rat_zero4()};
    RatB3 b = Bw(nt, 0);
    RatB3 a[2] = {A.row_frag(mt0, 0), A.row_frag(mt0 + 1, 0)};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int sn = s + 1 < KS ? s + 1 : s;
        const RatB3 bn = Bw(nt, sn);
        const RatB3 an[2] = {A.row_frag(mt0, sn), A.row_frag(mt0 + 1, sn)};
        rat_mfma3_block<2>(acc, a, b);
        a[0] = an[0];
        a[1] = an[1];
        b = bn;
    }
    epi(mt0, nt, acc[0]);
    epi(mt0 + 1, nt, acc[1]);
}

// ---- the attention-backward core on the MATRIX pipe (round 5; tools/probes/attn_bwd_core_probe.hip is its stand-alone twin) -----------
// Every (sequence, head) pair of the chunk is ONE wave's job, exact fp32 on v_mfma_f32_16x16x4_f32 (no operand splitting):
//   S = Q K^T and dP = dO V^T as 16 x 16 tiles over k = dim_head (10 -> 12, three steps); p = exp2(S scale log2e - lse) and
//   dS = p (dP - delta) on the accumulators; P, then dS, through a wave-private [16][33] LDS tile into the A operands of
//   dV += P^T dO, dQ = dS K, dK += dS^T Q.  Nothing is recomputed (5 products per pair; the two VALU passes do 7) and there is no
//   work-group barrier inside the core.  NIT = 16-row tiles per sequence (1: L <= 16, 2: L <= 32).
// Measured per chunk on the MI355X (profiles/round5/r5_attn_bwd_core_probe.txt, cycles, every CU busy):
//   L 31: 20.8 k against 27.4 k for the VALU passes (x 0.76), L 16: 13.2 k against 16.1 k (x 0.82) — but L 21: 30.2 k against 19.9 k and
//   L 11: 16.2 k against 11.3 k: a pair costs ~10.4 k (NIT 2) / ~3.2 k (NIT 1) cycles whatever L is, the VALU passes ~14 L^2.  Inside
//   the kernel (profiles/round5/r5_attn_bwd_core_ab.txt): L 31 1.745 against 1.830 ms per launch alone, 1.70 against 1.88 ms in the
//   Tmall-like step (0.39 -> 0.435 of the fp32 MFMA roofline); L 16 no difference (+-3 %).  So the host selects it for L >= 28 only
//   (b3_matrix_core): BASELINE configs[4]'s cross-sample sequences (K = 30 -> L = 31).
template <int NIT>
__device__ __forceinline__ void b3_bwd_core_mfma(float* qkv, float* ob, const float* dob, const float* lses, float* scratch, int L, int nsq,
                                                 float scale) {
    constexpr int SCR = 16 * 33 + 16;
    const int w = rat_wave(), l = rat_lane(), g = l >> 4, m = l & 15;
    float* scr = scratch + w * SCR;
    float* dl = scr + 16 * 33;
    const float sl2 = scale * RAT_LOG2E;
    const int npairs = nsq * B3_H;
    const bool cm = m < B3_DH;                                // this lane's column of a [.][dim_head] operand exists
    const int mc = m;                                         // (loads are unconditional at their natural address — rows / columns past the
                                                              //  operand stay inside the kernel's LDS — and masked by a select: base + immediate)
    for (int pair = w; pair < npairs; pair += ATT_WAVES) {
        const int h = pair % B3_H, sq = pair / B3_H;
        const int r0 = sq * L, cq = h * B3_DH, ck = B3_I + h * B3_DH, cv = 2 * B3_I + h * B3_DH;
        f32x4 adK[NIT], adV[NIT];
#pragma unroll
        for (int jt = 0; jt < NIT; ++jt) adK[jt] = adV[jt] = rat_zero4();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i0 = 16 * it;
            const int irows = L - i0 < 16 ? L - i0 : 16;
            {                                                 // delta_i = dO_i . O_i for the tile's rows
                const int rr = r0 + i0 + m;
                const float* a_ = dob + (size_t)rr * B3_LDT + cq;
                const float* b_ = ob + (size_t)rr * B3_LDT + cq;
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < B3_DH; c += 2) {
                    const float2 x = *reinterpret_cast<const float2*>(a_ + c), y = *reinterpret_cast<const float2*>(b_ + c);
                    d = fmaf(x.x, y.x, d);
                    d = fmaf(x.y, y.y, d);
                }
                if (l < 16) dl[l] = m < irows ? d : 0.f;
            }
            // stage 1 operands (A: lane holds [row m][k g]; B: [k g][col m]), all requested before the first MFMA
            float aq[3], ao[3], bk[3][NIT], bv[3][NIT];
            const int ri = r0 + i0 + m;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int cc = 4 * ks + g;
                aq[ks] = qkv[(size_t)ri * B3_LDQ + cq + cc];
                ao[ks] = dob[(size_t)ri * B3_LDT + cq + cc];
#pragma unroll
                for (int jt = 0; jt < NIT; ++jt) {
                    const int rj = r0 + 16 * jt + m;
                    bk[ks][jt] = qkv[(size_t)rj * B3_LDQ + ck + cc];
                    bv[ks][jt] = qkv[(size_t)rj * B3_LDQ + cv + cc];
                }
            }
            f32x4 aS[NIT], aP[NIT];
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt) aS[jt] = aP[jt] = rat_zero4();
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const bool okc = 4 * ks + g < B3_DH, oki = okc && m < irows;
                const float xq = oki ? aq[ks] : 0.f, xo = oki ? ao[ks] : 0.f;
#pragma unroll
                for (int jt = 0; jt < NIT; ++jt) {
                    const bool okj = okc && 16 * jt + m < L;
                    aS[jt] = RAT_MFMA16(xq, okj ? bk[ks][jt] : 0.f, aS[jt]);
                    aP[jt] = RAT_MFMA16(xo, okj ? bv[ks][jt] : 0.f, aP[jt]);
                }
            }
            RAT_WAVE_FENCE();
            // p and dS on the accumulators (C layout: column m = key, rows 4 g + r = query)
            f32x4 dS[NIT];
            {
                float lse4[4], d4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ii = 4 * g + r;
                    lse4[r] = lses[(r0 + i0 + ii) * B3_H + h];
                    d4[r] = dl[ii];
                }
#pragma unroll
                for (int jt = 0; jt < NIT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = 4 * g + r < irows && 16 * jt + m < L;
                        const float p = ok ? rat_exp2(aS[jt][r] * sl2 - lse4[r]) : 0.f;
                        scr[(4 * g + r) * 33 + 16 * jt + m] = p;
                        dS[jt][r] = p * (aP[jt][r] - d4[r]);
                    }
            }
            RAT_WAVE_FENCE();
            // dV[j][c] += sum_i P[i][j] dO[i][c]   (A = P^T from the tile, B = dO; rows beyond the tile carry P = 0)
            {
                float bdo[4], ap[4][NIT];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ii = 4 * ks + g;
                    bdo[ks] = dob[(size_t)(r0 + i0 + ii) * B3_LDT + cq + mc];
#pragma unroll
                    for (int jt = 0; jt < NIT; ++jt) ap[ks][jt] = scr[ii * 33 + 16 * jt + m];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float b = (cm && 4 * ks + g < irows) ? bdo[ks] : 0.f;
#pragma unroll
                    for (int jt = 0; jt < NIT; ++jt) adV[jt] = RAT_MFMA16(ap[ks][jt], b, adV[jt]);
                }
            }
            RAT_WAVE_FENCE();
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(4 * g + r) * 33 + 16 * jt + m] = dS[jt][r];
            RAT_WAVE_FENCE();
            // dK[j][c] += sum_i dS[i][j] Q[i][c]   (A = dS^T, B = Q);   dQ[i][c] = sum_j dS[i][j] K[j][c]   (A = dS, B = K)
            {
                float bq[4], at[4][NIT];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ii = 4 * ks + g;
                    bq[ks] = qkv[(size_t)(r0 + i0 + ii) * B3_LDQ + cq + mc];
#pragma unroll
                    for (int jt = 0; jt < NIT; ++jt) at[ks][jt] = scr[ii * 33 + 16 * jt + m];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float b = (cm && 4 * ks + g < irows) ? bq[ks] : 0.f;
#pragma unroll
                    for (int jt = 0; jt < NIT; ++jt) adK[jt] = RAT_MFMA16(at[ks][jt], b, adK[jt]);
                }
            }
            f32x4 adQ = rat_zero4();
#pragma unroll
            for (int half = 0; half < NIT; ++half) {
                float bkk[4], as[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int jj = 16 * half + 4 * ks + g;
                    bkk[ks] = qkv[(size_t)(r0 + jj) * B3_LDQ + ck + mc];
                    as[ks] = scr[m * 33 + jj];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) adQ = RAT_MFMA16(as[ks], (cm && 16 * half + 4 * ks + g < L) ? bkk[ks] : 0.f, adQ);
            }
            if (cm)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < irows) ob[(size_t)(r0 + i0 + 4 * g + r) * B3_LDT + cq + m] = adQ[r] * scale;
            RAT_WAVE_FENCE();
        }
        if (cm)
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * jt + 4 * g + r;
                    if (j < L) {
                        qkv[(size_t)(r0 + j) * B3_LDQ + ck + m] = adK[jt][r] * scale;
                        qkv[(size_t)(r0 + j) * B3_LDQ + cv + m] = adV[jt][r];
                    }
                }
    }
}
static_assert((size_t)ATT_WAVES * (16 * 33 + 16) * 4 <= (size_t)3 * B3_XP, "the matrix core's wave-private tiles live in the dead dy planes");

// PH: sequences of at most 12 tokens — the probabilities P of pass 1 fit the (then dead) dy planes (64 rows x L x 8 heads x 4 B <= 24 KB)
// and are handed to pass 2, which then needs neither the q . k product nor the exponential again
// MC (1 / 2 = 16-row tiles per sequence): the attention core on the matrix pipe (b3_bwd_core_mfma) instead of the two VALU passes;
// sequences of at most 32 tokens, every position a query
template <bool EX, bool QSUB = false, bool DPAD = false, bool PH = false, int MC = 0>
__global__ void __launch_bounds__(ATT_THREADS) attn_bwd3_kernel(AttnArgs a, Attn3W W) {
    static_assert(MC == 0 || (!QSUB && !PH), "the matrix core computes every query and hands nothing over");   // MC = 16-row tiles per sequence (1 / 2)
    RAT_DYN_SMEM(smem);
    const PlanesX xp{smem};                                                  // LayerNorm(x)
    const PlanesX dyp{smem + B3_OFF_DYP};                                    // dy (x out_scale)
    float* dxn = reinterpret_cast<float*>(smem + B3_OFF_DYP);                // [64][68] d(LN out), over the dead dy planes
    float* qkv = reinterpret_cast<float*>(smem + B3_OFF_QKV);                // [64][244] Q|K|V, then dK|dV in place
    const PlanesQ dqp{smem + B3_OFF_QKV};                                    // d(Q|K|V) planes, over qkv / ob / dob
    float* ob = reinterpret_cast<float*>(smem + B3_OFF_OB);                  // [64][84] O, then dQ
    float* dob = reinterpret_cast<float*>(smem + B3_OFF_DOB);                // [64][84] dO
    float* mu = reinterpret_cast<float*>(smem + B3_OFF_MISC);
    float* rs = mu + ATT_ROWS;
    float* lses = rs + ATT_ROWS;                                             // [64][8]
    float* dlt = lses + ATT_ROWS * B3_H;                                     // [64][8]
    int64_t* const rowtok0 = reinterpret_cast<int64_t*>(dlt + ATT_ROWS * B3_H);
    const int L = a.L;
    const int r_own = threadIdx.x >> 3, sub = threadIdx.x & 7;               // this thread's (row slot, 8-column piece)
    const int dreal = DPAD ? a.d : B3_D;                                     // DPAD: embedding_dim 40 / 48 / 56 inside the 64-wide tiles
    const bool colok = !DPAD || 8 * sub < dreal;

    f32x4 accq[QSLOTS], acco[OSLOTS];                                        // persistent dW_qkv / dW_out^T tiles
#pragma unroll
    for (int i = 0; i < QSLOTS; ++i) accq[i] = rat_zero4();
#pragma unroll
    for (int i = 0; i < OSLOTS; ++i) acco[i] = rat_zero4();
    float dgam[8], dbet[8], dbo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dgam[k] = dbet[k] = dbo[k] = 0.f;
    float* const lnw = reinterpret_cast<float*>(rowtok0 + 2 * ATT_ROWS);     // [2][64] LayerNorm gamma | beta (kept out of the registers)
    if (threadIdx.x < 2 * B3_D) {
        const int c = threadIdx.x < B3_D ? threadIdx.x : threadIdx.x - B3_D;
        lnw[threadIdx.x] = (DPAD && c >= dreal) ? 0.f : (threadIdx.x < B3_D ? a.ln_g[c] : a.ln_b[c]);
    }
    for (int e = threadIdx.x; e < (int)((B3_OFF_MISC - B3_OFF_QKV) / 4); e += ATT_THREADS) qkv[e] = 0.f;   // pad columns, slack
    {
        int nsq0, rows0;
        map_rows_b3<RAT_MAP_BWD>(a, blockIdx.x, rowtok0, nsq0, rows0);
    }
    __syncthreads();
    RAT_PROF_DECL
    int parity = 0;
    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x, parity ^= 1) {
        const int64_t* rowtok = rowtok0 + parity * ATT_ROWS;
        int nsq;
        {
            const int64_t q0 = chunk * a.nsq_chunk;
            const int64_t left = a.nseq - q0;
            nsq = left < a.nsq_chunk ? (int)left : a.nsq_chunk;
        }
        const int64_t tok_own = rowtok[r_own];
        // ---- P0: x -> LayerNorm -> planes; dy -> planes (+ db_out partials); O, lse -> fp32 tiles
        {   // Same-box A/B of the alternatives (tools/ab_attn.sh): touching the next chunk's lines into L2 behind the VALU passes +4-5 %
            // (also with the touched value waited for right after pass 1 instead of at the end of the iteration);
            // requesting the next chunk's rows a phase or two early (P4, P5, P6) +8-10 % — the registers that carry them across the
            // GEMM phases come back as spills, and a spill reload is a scratch load that waits for vmcnt(0).
            const bool valid = tok_own >= 0 && colok;
            const uint32_t po = DPAD ? b3_piece_off_d(tok_own, dreal, colok) : b3_piece_off(tok_own);
            B3RowFetchO fo;
            float4 x0 = b3_ld4(a.x, po), x1 = b3_ld4(a.x, po + 16u);
#ifdef RAT_ATTN_BWD_DY_NT
            float4 d0 = rat_ld4_stream(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.dy) + po)),
                   d1 = rat_ld4_stream(reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.dy) + po + 16u));
#else
            float4 d0 = b3_ld4(a.dy, po), d1 = b3_ld4(a.dy, po + 16u);
#endif
            fo.issue(a.o_save, rowtok);
            float lsen = b3_ld1(a.lse_save, (uint32_t)(tok_own >= 0 ? tok_own : 0) * (uint32_t)(B3_H * 4) + 4u * sub);
            RAT_SCHED_FENCE();                                               // every request is out before anything is consumed
            b3_zero_unless(valid, x0);
            b3_zero_unless(valid, x1);
            b3_zero_unless(valid, d0);
            b3_zero_unless(valid, d1);
            lsen = tok_own >= 0 ? lsen : 0.f;
            {
                float gam[8], bet[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    gam[k] = lnw[8 * sub + k];
                    bet[k] = lnw[B3_D + 8 * sub + k];
                }
                b3_layer_norm_to_planes<DPAD>(tok_own >= 0, x0, x1, a.eps, xp, gam, bet, mu, rs, dreal, colok);
            }
            if (EX && a.drop.threshold != 0 && valid) {                      // dy through the projection's Dropout
                const int64_t i0 = tok_own * dreal + 8 * sub;
                d0.x = a.drop.apply(d0.x, i0); d0.y = a.drop.apply(d0.y, i0 + 1); d0.z = a.drop.apply(d0.z, i0 + 2); d0.w = a.drop.apply(d0.w, i0 + 3);
                d1.x = a.drop.apply(d1.x, i0 + 4); d1.y = a.drop.apply(d1.y, i0 + 5); d1.z = a.drop.apply(d1.z, i0 + 6); d1.w = a.drop.apply(d1.w, i0 + 7);
            }
            if (EX && a.out_scale != 1.0f) {
                const float m_ = a.out_scale;
                d0.x *= m_; d0.y *= m_; d0.z *= m_; d0.w *= m_; d1.x *= m_; d1.y *= m_; d1.z *= m_; d1.w *= m_;
            }
            dbo[0] += d0.x; dbo[1] += d0.y; dbo[2] += d0.z; dbo[3] += d0.w; dbo[4] += d1.x; dbo[5] += d1.y; dbo[6] += d1.z; dbo[7] += d1.w;
            rat_u4 h, m, l;
            rat_split8(d0, d1, h, m, l);
            dyp.store(r_own, sub, h, m, l);
            fo.stash(ob, B3_LDT);
            lses[threadIdx.x] = lsen;                                        // 512 threads = 64 rows x 8 heads
        }
        if (chunk + gridDim.x < a.nchunks) {
            int nsq1, rows1;
            map_rows_b3<RAT_MAP_BWD>(a, chunk + gridDim.x, (rowtok0 + (parity ^ 1) * ATT_ROWS), nsq1, rows1);
        }
        __syncthreads();
        RAT_PROF_MARK(0);
        // ---- P1: Q|K|V = LN(x) W_qkv^T   P2: dO = dy W_out   P2b: dW_out^T += O^T dy
        {
            auto epi_q = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
#pragma unroll
                for (int r = 0; r < 4; ++r) qkv[(size_t)rat_acc_row(mt, r) * B3_LDQ + col] = acc[r];
            };
            auto epi_o = [&](int mt, int nt, const f32x4& acc) {
                const int col = rat_acc_col(nt);
#pragma unroll
                for (int r = 0; r < 4; ++r) dob[(size_t)rat_acc_row(mt, r) * B3_LDT + col] = acc[r];
            };
            b3_gemm_rows<2>(xp, W.qkv, B3_Q3 / 16, epi_q);                          // 15 column tiles: 4, 4, 4, 3 per column-tile group
            RAT_SCHED_FENCE();
            RAT_PROF_MARK(1);
            b3_gemm_rows<2, true>(dyp, W.outT, B3_I / 16, epi_o);                   //  5 column tiles dealt from the other end: 1, 1, 1, 2
            RAT_SCHED_FENCE();
        }
        RAT_PROF_MARK(2);
#ifdef RAT_DWOUT_R4                                                           // A/B only: round 4's assignment (wave = inner tile, waves 5-7 idle)
        if (rat_wave() < B3_I / 16) {
            const int mt = rat_wave(), l = rat_lane(), g = l >> 4, col = 16 * mt + (l & 15);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ob[(size_t)rat_col_slot_row(s, g, j) * B3_LDT + col];
                const RatB3 af = rat_split8_frag(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
#pragma unroll
                for (int nt = 0; nt < OSLOTS; ++nt) {
                    acco[nt] = rat_mfma3(af, dyp.col_frag(nt, s), acco[nt]);
                    RAT_SCHED_FENCE();
                }
            }
        }
#else
        {   // dW_out^T is 5 inner tiles x 4 column tiles.  Waves 0-3 own inner tile w (4 column tiles each, as before); inner tile 4 —
            // round 4 gave all of it to wave 4, which shares a SIMD with wave 0: 96 MFMAs on that SIMD against 48 on the others, three
            // waves idle — is dealt one column tile each to waves 4-7: 60 MFMAs per SIMD.
            static_assert(B3_I / 16 == 5 && OSLOTS == 4 && ATT_WAVES == 8, "the dW_out assignment is written for 5 x 4 tiles on 8 waves");
            const int w = rat_wave(), mt = w < 4 ? w : 4, l = rat_lane(), g = l >> 4, col = 16 * mt + (l & 15);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ob[(size_t)rat_col_slot_row(s, g, j) * B3_LDT + col];
                const RatB3 af = rat_split8_frag(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
                if (w < 4) {
#pragma unroll
                    for (int nt = 0; nt < OSLOTS; ++nt) {
                        acco[nt] = rat_mfma3(af, dyp.col_frag(nt, s), acco[nt]);
                        RAT_SCHED_FENCE();
                    }
                } else {
                    acco[0] = rat_mfma3(af, dyp.col_frag(w - 4, s), acco[0]);
                }
            }
        }
#endif
        __syncthreads();
        RAT_PROF_MARK(3);
        // ---- P3: attention backward on the VALU — the two passes of attn_bwd_kernel<64, 10>.  Same-box A/B of the loop shapes
        // (tools/ab_attn.sh): 2 / 3 / 4 keys per trip with all their rows requested up front +7 / +14 / +20 % (spills), the
        // software-pipelined form (rows of key j + 1 requested before the arithmetic of key j, ping-pong registers) +4 %.
        // A ONE-pass form was built and measured too (every (sequence, head) group inside one wave; lane = query owner AND key owner of
        // row pos; step t: key (pos + t) mod L, (p, dS) handed to the key's owner by a lane shuffle, so nothing is recomputed: 25
        // instead of 35 packed FMAs per pair): correct, but +29 % at L = 21 and +16 % at L = 11 — its K / V / q / dO reads are a
        // different row per lane, while in both passes below all lanes of a group read the SAME row (an LDS broadcast).
        // RatSeqMap.queries < L: the dy rows of the other positions are zero by contract, so their dQ is zero and they add nothing to
        // dK / dV — pass 1 runs for nq queries per sequence, pass 2 sums over them.
        typedef HeadVec<B3_DH> HV;
        const int nq = QSUB ? a.nq : L;
        const int ntasks = nsq * B3_H * L, nqtasks = QSUB ? nsq * B3_H * nq : ntasks;
        const float sl2 = a.scale * RAT_LOG2E;
        if (MC) b3_bwd_core_mfma<(MC > 0 ? MC : 1)>(qkv, ob, dob, lses, dxn, L, nsq, a.scale);   // (the dy planes are dead since P2b: the wave-private tiles go there)
        for (int task = threadIdx.x; !MC && task < nqtasks; task += ATT_THREADS) {
            const int i = task % nq;
            const int h = (task / nq) % B3_H;
            const int sq = task / (nq * B3_H);
            const int row_i = sq * L + i;
            const int ho = h * B3_DH;
            float* opp = ob + (size_t)row_i * B3_LDT + ho;
            HV q, go, dq, kv;
            q.load(qkv + (size_t)row_i * B3_LDQ + ho, B3_DH);
            go.load(dob + (size_t)row_i * B3_LDT + ho, B3_DH);
            kv.load(opp, B3_DH);
            const float delta = go.dot(kv);
            dq.zero();
            dlt[row_i * B3_H + h] = delta;
            const float lse = lses[row_i * B3_H + h];
            const float* kbase = qkv + (size_t)(sq * L) * B3_LDQ + B3_I + ho;
            float* const prow = PH ? dxn + ((sq * B3_H + h) * L + i) * L : nullptr;     // P[(sequence, head)][query i][key j]
            for (int j = 0; j < L; ++j) {
                const float* kp = kbase + (size_t)j * B3_LDQ;
                kv.load(kp + B3_I, B3_DH);
                const float dp = go.dot(kv);
                kv.load(kp, B3_DH);
                const float p = rat_exp2(q.dot(kv) * sl2 - lse);
                if (PH) prow[j] = p;
                dq.axpy(p * (dp - delta), kv);
            }
            dq.store(opp, B3_DH, a.scale);
        }
        if (QSUB && nq < L) {
            for (int e = threadIdx.x; e < nsq * L * B3_H; e += ATT_THREADS) {
                const int r = e / B3_H, h = e - r * B3_H;
                if (r % L < nq) continue;
                HV z;
                z.zero();
                z.store(ob + (size_t)r * B3_LDT + h * B3_DH, B3_DH, 1.0f);    // dQ of a position that is no query
            }
        }
        __syncthreads();
        RAT_PROF_MARK(4);
        for (int task = threadIdx.x; !MC && task < ntasks; task += ATT_THREADS) {
            const int j = task % L;
            const int h = (task / L) % B3_H;
            const int sq = task / (L * B3_H);
            const int ho = h * B3_DH;
            float* kp = qkv + (size_t)(sq * L + j) * B3_LDQ + B3_I + ho;
            HV kk, vv, dk, dv, t;
            kk.load(kp, B3_DH);
            vv.load(kp + B3_I, B3_DH);
            dk.zero();
            dv.zero();
            for (int i = 0; i < nq; ++i) {
                const int row_i = sq * L + i;
                t.load(dob + (size_t)row_i * B3_LDT + ho, B3_DH);
                const float dp = t.dot(vv);
                const float delta = dlt[row_i * B3_H + h];
                HV qv;
                qv.load(qkv + (size_t)row_i * B3_LDQ + ho, B3_DH);
                float p;
                if (PH) p = dxn[((sq * B3_H + h) * L + i) * L + j];           // consecutive lanes = consecutive keys: conflict-free
                else p = rat_exp2(qv.dot(kk) * sl2 - lses[row_i * B3_H + h]);
                dv.axpy(p, t);
                dk.axpy(p * (dp - delta), qv);
            }
            dk.store(kp, B3_DH, a.scale);
            dv.store(kp + B3_I, B3_DH, 1.0f);
        }
        __syncthreads();
        RAT_PROF_MARK(5);
        // ---- P3c: d(Q|K|V) = [dQ (in ob) | dK | dV (in qkv)] -> planes over the three fp32 tiles: all reads, barrier, all writes
        {
            constexpr int NP = B3_Q3 / 8;                                    // 30 pieces per row
            constexpr int NIT = (ATT_ROWS * NP + ATT_THREADS - 1) / ATT_THREADS;
            float4 lo[NIT], hi[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                lo[it] = hi[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < ATT_ROWS * NP) {
                    const int r = e / NP, o = e - r * NP;
                    const float* src = o < B3_I / 8 ? ob + (size_t)r * B3_LDT + 8 * o : qkv + (size_t)r * B3_LDQ + 8 * o;
                    lo[it] = *reinterpret_cast<const float4*>(src);
                    hi[it] = *reinterpret_cast<const float4*>(src + 4);
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                if (e < ATT_ROWS * NP) {
                    const int r = e / NP, o = e - r * NP;
                    rat_u4 h, m, l;
                    rat_split8(lo[it], hi[it], h, m, l);
                    dqp.store(r, o, h, m, l);
                }
            }
            // the padded last K step of P4 reads 32 bytes past each plane's last row: for the first two planes that is the next
            // plane's first row (finite), behind the third it is stale fp32 data whose halves may look like bf16 NaNs — clear it
            if (threadIdx.x < 8) reinterpret_cast<float*>(smem + B3_OFF_QKV + (size_t)3 * B3_QP)[threadIdx.x] = 0.f;
        }
        __syncthreads();
        RAT_PROF_MARK(6);
        // ---- P4: d(LN out) = dQKV W_qkv   P5: dW_qkv += dQKV^T LN(x)
        b3_gemm_rows_longk<8>(dqp, W.qkvT, [&](int mt, int nt, const f32x4& acc) {
            const int col = rat_acc_col(nt);
#pragma unroll
            for (int r = 0; r < 4; ++r) dxn[(size_t)rat_acc_row(mt, r) * B3_LDN + col] = acc[r];
        });
        RAT_SCHED_FENCE();
        RAT_PROF_MARK(7);
        {
            const int w = rat_wave(), nt = w & 3;
            const RatB3 b0 = xp.col_frag(nt, 0), b1 = xp.col_frag(nt, 1);
            RatB3 a0 = dqp.col_frag(w >> 2, 0), a1 = dqp.col_frag(w >> 2, 1);
#pragma unroll
            for (int i = 0; i < QSLOTS; ++i) {
                const int mt = (w >> 2) + 2 * i;
                const int mn = mt + 2 < B3_Q3 / 16 ? mt + 2 : mt;
                const RatB3 n0 = dqp.col_frag(mn, 0), n1 = dqp.col_frag(mn, 1);
                if (mt < B3_Q3 / 16) {
                    accq[i] = rat_mfma3(a0, b0, accq[i]);
                    accq[i] = rat_mfma3(a1, b1, accq[i]);
                }
                a0 = n0;
                a1 = n1;
            }
        }
        __syncthreads();
        RAT_PROF_MARK(8);
        // ---- P6: LayerNorm backward + the added gradient: dx = add + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxn * gamma
        {
            const bool valid = tok_own >= 0 && colok;
            const float mean = mu[r_own], rstd = rs[r_own];
            const float* addp = EX ? a.add : a.dy;
            float xh[8], gg[8], ad[8], out[8], gam[8];
            float4 xv2[2], av2[2];
            const uint32_t po = DPAD ? b3_piece_off_d(tok_own, dreal, colok) : b3_piece_off(tok_own);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                xv2[k] = b3_ld4(a.x, po + 16u * k);
                av2[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (addp != nullptr) av2[k] = b3_ld4(addp, po + 16u * k);    // uniform branch
            }
            RAT_SCHED_FENCE();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                b3_zero_unless(valid, xv2[k]);
                b3_zero_unless(valid, av2[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) gam[k] = lnw[8 * sub + k];
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                const float4 xv = xv2[k >> 2], av = av2[k >> 2];
                const float4 gv = *reinterpret_cast<const float4*>(dxn + (size_t)r_own * B3_LDN + 8 * sub + k);
                xh[k] = xv.x; xh[k + 1] = xv.y; xh[k + 2] = xv.z; xh[k + 3] = xv.w;
                ad[k] = av.x; ad[k + 1] = av.y; ad[k + 2] = av.z; ad[k + 3] = av.w;
                gg[k] = gv.x; gg[k + 1] = gv.y; gg[k + 2] = gv.z; gg[k + 3] = gv.w;
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xh[k] = valid ? (xh[k] - mean) * rstd : 0.f;
                gg[k] = valid ? gg[k] : 0.f;
                const float gw = valid ? gg[k] * gam[k] : 0.f;
                s1 += gw;
                s2 += gw * xh[k];
            }
            s1 = rat_group_sum<8>(s1) / (DPAD ? (float)dreal : (float)B3_D);
            s2 = rat_group_sum<8>(s2) / (DPAD ? (float)dreal : (float)B3_D);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gw = valid ? gg[k] * gam[k] : 0.f;
                out[k] = valid ? ad[k] + rstd * (gw - s1 - xh[k] * s2) : 0.f;
                dgam[k] += gg[k] * xh[k];
                dbet[k] += gg[k];
            }
            if (valid) {
                b3_st4(a.y, po, make_float4(out[0], out[1], out[2], out[3]));
                b3_st4(a.y, po + 16u, make_float4(out[4], out[5], out[6], out[7]));
            }
        }
        __syncthreads();
        RAT_PROF_MARK(9);
    }
    RAT_PROF_FLUSH(a.prof, 60);

    // ---- this work-group's parameter-gradient slab: [dW_qkv | dW_out | db_out | dgamma | dbeta]
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
    float* s_wqkv = slab;                                                    // (the host's layout: [3 I][d], [d][I], [d], [d], [d])
    float* s_wout = s_wqkv + (int64_t)B3_Q3 * dreal;
    float* s_bout = s_wout + (int64_t)dreal * B3_I;
    float* s_gam = s_bout + dreal;
    float* s_bet = s_gam + dreal;
    {
        const int w = rat_wave(), col = rat_acc_col(w & 3);
#pragma unroll
        for (int i = 0; i < QSLOTS; ++i) {
            const int mt = (w >> 2) + 2 * i;
            if (mt < B3_Q3 / 16 && (!DPAD || col < dreal))
#pragma unroll
                for (int r = 0; r < 4; ++r) s_wqkv[(int64_t)rat_acc_row(mt, r) * dreal + col] = accq[i][r];
        }
#ifdef RAT_DWOUT_R4
        if (w < B3_I / 16)
#pragma unroll
            for (int nt = 0; nt < OSLOTS; ++nt)
                if (!DPAD || rat_acc_col(nt) < dreal)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_wout[(int64_t)rat_acc_col(nt) * B3_I + rat_acc_row(w, r)] = acco[nt][r];
#else
        if (w < 4) {
#pragma unroll
            for (int nt = 0; nt < OSLOTS; ++nt)
                if (!DPAD || rat_acc_col(nt) < dreal)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_wout[(int64_t)rat_acc_col(nt) * B3_I + rat_acc_row(w, r)] = acco[nt][r];
        } else if (!DPAD || rat_acc_col(w - 4) < dreal) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_wout[(int64_t)rat_acc_col(w - 4) * B3_I + rat_acc_row(4, r)] = acco[0][r];
        }
#endif
    }
    // db_out / dgamma / dbeta: 64 row-slot partials per column -> LDS -> fixed-order column sums
    float* red = reinterpret_cast<float*>(smem);                             // [64][68]
    float* const outs[3] = {s_bout, s_gam, s_bet};
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) red[(size_t)r_own * B3_LDN + 8 * sub + k] = which == 0 ? dbo[k] : (which == 1 ? dgam[k] : dbet[k]);
        __syncthreads();
        if ((int)threadIdx.x < dreal) {
            float sacc = 0.f;
            for (int rr = 0; rr < ATT_ROWS; ++rr) sacc += red[(size_t)rr * B3_LDN + threadIdx.x];
            outs[which][threadIdx.x] = sacc;
        }
    }
}

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
static bool b3_geom(int d, int heads, int dim_head) { return b3_dim(d) && heads == B3_H && dim_head == B3_DH; }
static bool b3_shape(int d, int heads, int dim_head, const RatAttnParams* w) {
    return b3_geom(d, heads, dim_head) && w->w_out != nullptr;
}
// PH instantiation of attn_bwd3_kernel: P of a chunk inside the dy planes' 24 KB
static bool b3_ph_fits(int L, int nsq_chunk) { return (size_t)nsq_chunk * B3_H * L * L * 4 <= (size_t)3 * B3_XP; }
// the matrix-pipe backward core (b3_bwd_core_mfma) by sequence length — see its comment for the measurements behind the rule; the
// attn_bwd_core_mfma knob forces it on (1, any L <= 32) or off (0)
static bool b3_matrix_core(int L) {
    const int k = rat_knob(RAT_KNOB_ATTN_BWD_CORE_MFMA);
    if (L > 32 || k == 0) return false;
    return k == 1 || L >= 28;
}
// the matrix-pipe FORWARD core (b3_fwd_core_mfma): attn_fwd_core_mfma knob 0 = by length (28 ... 32 tokens), 2 = forced on (L <= 32), 3 = off
// (1 selects attn_fwd3m_kernel, round 3's bf16x3 core)
static int b3_fwd_matrix_core(int L, int nq) {
    const int k = rat_knob(RAT_KNOB_ATTN_FWD_CORE_MFMA);
    if (L > 32 || nq < L || k == 3 || k == 1) return 0;
    if (k != 2 && L < 28) return 0;
    return L > 16 ? 2 : 1;
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
        // the attn_fwd_core_mfma knob (RAT_ATTN_FWD_CORE=mfma at load) selects attn_fwd3m_kernel (the attention core on the matrix pipe as well).  It is correct (same
        // parity gates) but MEASURED SLOWER than the VALU core at this geometry — 0.80 / 0.69 ms against 0.64 / 0.52 ms per launch at
        // L = 21 / 11 (profiles/round3/r3_attn_fwd_core_ab.txt): 16 x 16 score tiles are 43-47 % full at L = 21 / 11, and what the
        // matrix pipe saves is spent on the VALU again, splitting Q|K|V and P into bf16 chunks and laying them out — so it is opt-in.
        const bool mfma_core = rat_knob(RAT_KNOB_ATTN_FWD_CORE_MFMA) == 1;
        if (dpad) {
            if (plain && a.nq < a.L) RAT_LAUNCH((attn_fwd3_kernel<false, true, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
            else if (plain) RAT_LAUNCH((attn_fwd3_kernel<false, false, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
            else RAT_LAUNCH((attn_fwd3_kernel<true, false, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        } else if (mfma_core && b3m_fits(a.L, a.nsq_chunk)) {
            if (plain) RAT_LAUNCH((attn_fwd3m_kernel<false>), b3_blocks, ATT_THREADS, b3m_fwd_smem(), stream, a, W);
            else RAT_LAUNCH((attn_fwd3m_kernel<true>), b3_blocks, ATT_THREADS, b3m_fwd_smem(), stream, a, W);
        } else if (plain && a.nq < a.L) RAT_LAUNCH((attn_fwd3_kernel<false, true>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (plain && b3_fwd_matrix_core(a.L, a.nq) == 2)
            RAT_LAUNCH((attn_fwd3_kernel<false, false, false, false, 2>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (plain && b3_fwd_matrix_core(a.L, a.nq) == 1)      // (sequences of at most 16 tokens: only when the knob forces it)
            RAT_LAUNCH((attn_fwd3_kernel<false, false, false, false, 1>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (plain) RAT_LAUNCH((attn_fwd3_kernel<false>), b3_blocks, ATT_THREADS, b3_fwd_smem(), stream, a, W);
        else if (b3_fwd_matrix_core(a.L, a.L) == 2)                // (EX computes every position whatever `queries` says)
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
static int wide_groups(int d, int heads, int dim_head) {
    return (d >= 1 && d <= 16 && dim_head == WG_DH && heads > WG_H && heads % WG_H == 0 && heads / WG_H <= 8) ? heads / WG_H : 0;
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
        if (check_dims(map_host, d, WG_H, dim_head, false)) return -1;
        RAT_REQUIRE(x && y && w_host && w_host->ln_g && w_host->ln_b && w_host->w_qkv && w_host->w_out && w_host->b_out, "null pointer");
        RAT_REQUIRE((o_save == nullptr) == (lse_save == nullptr) && ntok > 0, "o_save and lse_save come together, [G][ntok][.]");
        AttnArgs a{};
        fill_common(a, w_host, map_host, d, WG_H, dim_head, ln_eps);
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
        const size_t smem = AttnGeom(d, WG_H, dim_head).fwd_smem() + (size_t)WG_WO_FLOATS * 4;
        const int per_cu = (int)((160 * 1024) / smem) >= 2 ? 2 : 1;
        const unsigned blocks = (unsigned)(a.nchunks < rat_max_blocks() * per_cu ? a.nchunks : rat_max_blocks() * per_cu);
        if (d == 10) RAT_LAUNCH((attn_fwd_wide_kernel<10>), blocks, ATT_THREADS, smem, stream, a);
        else RAT_LAUNCH((attn_fwd_wide_kernel<0>), blocks, ATT_THREADS, smem, stream, a);
        return rat_check_launch("rat_attn_fwd_groups (small d)");
    }
    const int G = b3_groups(d, heads, dim_head);
    RAT_REQUIRE(G > 0, "rat_attn_fwd_groups serves dim_head 10 and 16 ... 64 heads in groups of 8 at embedding_dim 64 or <= 16");
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
    RAT_REQUIRE(G > 0 && G <= WG_MAXG, "rat_attn_bwd_groups serves embedding_dim <= 16, dim_head 10 and 16 ... 32 heads in groups of 8");
    if (check_dims(map_host, d, WG_H, dim_head, true)) return -1;
    RAT_REQUIRE(x && dy && o_save && lse_save && dx && w_host && grads_host && workspace && ntok > 0, "null pointer");
    RAT_REQUIRE(w_host->ln_g && w_host->ln_b && w_host->w_qkv && w_host->w_out && w_host->b_out, "null parameter");
    RAT_REQUIRE(grads_host->ln_g && grads_host->ln_b && grads_host->w_qkv && grads_host->w_out && grads_host->b_out, "null gradient");
    RAT_REQUIRE(workspace_bytes >= rat_attn_bwd_groups_workspace(d, heads, dim_head), "workspace too small");
    AttnArgs a{};
    fill_common(a, w_host, map_host, d, WG_H, dim_head, ln_eps);
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
    const size_t smem = AttnGeom(d, WG_H, dim_head).bwd_smem(WG_H) + (size_t)(WG_WQ_FLOATS + WG_WO_FLOATS) * 4;
    if (d == 10) RAT_LAUNCH((attn_bwd_wide_kernel<10>), blocks, ATT_THREADS, smem, stream, a);
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
        if (dpad) {
            if (a.add_lds && a.nq < a.L) RAT_LAUNCH((attn_bwd3_kernel<false, true, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
            else if (a.add_lds) RAT_LAUNCH((attn_bwd3_kernel<false, false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
            else RAT_LAUNCH((attn_bwd3_kernel<true, false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        } else if (a.add_lds && a.nq < a.L) RAT_LAUNCH((attn_bwd3_kernel<false, true>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else if (b3_matrix_core(a.L) && a.nq >= a.L && a.add_lds && a.L > 16)
            RAT_LAUNCH((attn_bwd3_kernel<false, false, false, false, 2>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else if (b3_matrix_core(a.L) && a.nq >= a.L && a.L > 16)
            RAT_LAUNCH((attn_bwd3_kernel<true, false, false, false, 2>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else if (b3_matrix_core(a.L) && a.nq >= a.L && a.add_lds)          // (sequences of at most 16 tokens: only when the knob forces it)
            RAT_LAUNCH((attn_bwd3_kernel<false, false, false, false, 1>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
        else if (b3_matrix_core(a.L) && a.nq >= a.L)
            RAT_LAUNCH((attn_bwd3_kernel<true, false, false, false, 1>), blocks, ATT_THREADS, b3_bwd_smem(), stream, a, W);
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
