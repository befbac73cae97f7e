// attn_b3.h — part of attn.hip's translation unit (included inside its anonymous namespace, after the shared helpers and the exact-fp32
// kernels): the bf16x3 kernels of the north-star geometry (embedding_dim 64 / 40 / 48 / 56, 8 heads x 10) — attn_fwd3_kernel,
// attn_bwd3_kernel, their matrix-pipe attention cores and the GEMM / plane helpers they share.
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
        // (no selects on these operands — round 6: K columns 10, 11 of the k dimension meet Q's, which ARE masked to zero below; key rows past
        //  the sequence only reach scores that are masked to -inf before the softmax, and V rows past it meet p = 0; V "columns" m >= 10 only
        //  produce output rows c >= 10 of O^T, which are never stored.  Everything read is finite: rows of a neighbouring sequence or the
        //  zero rows the padded projection writes)
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
// NH (round 6): heads of the layer, 8 (x 10) or 4 (x 20) — RAT_m3 at the north-star config runs heads / 2 heads of width 2 dim_head
// (RAT_m3.py:181): the same projections and planes (inner = 80), a different slicing of Q | K | V in the VALU core.  NH = 4 has no matrix
// core, group loop or narrower-embedding instantiation (nothing asks for them).
template <bool EX, bool QSUB = false, bool DPAD = false, bool GRP = false, int MCF = 0, int NH = B3_H>
__global__ void __launch_bounds__(ATT_THREADS) attn_fwd3_kernel(AttnArgs a, Attn3W W) {
    constexpr int NDH = B3_I / NH;
    static_assert(NH == B3_H || (NH == 4 && MCF == 0 && !GRP && !DPAD), "the matrix cores, the group loop and DPAD are written for 8 heads x 10");
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
        map_rows(a, blockIdx.x, rowtok0, nsq0, rows0);
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
            map_rows(a, chunk + gridDim.x, (rowtok0 + (parity ^ 1) * ATT_ROWS), nsq1, rows1);
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
        float* const lse_save = (GRP && a.lse_save != nullptr) ? a.lse_save + (int64_t)grp * a.group_tok * NH : a.lse_save;
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
        typedef HeadVec<NDH> HV;
        const int nq = QSUB ? a.nq : L;                          // queries that matter per sequence (RatSeqMap.queries; normally L)
        const int ntasks = MCF ? 0 : nsq * NH * nq;
        const float sl2 = a.scale * RAT_LOG2E;
        if (MCF) b3_fwd_core_mfma<(MCF > 0 ? MCF : 1)>(qkv, lse_s, L, nsq, a.scale);
        for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
            const int i = task % nq;
            const int h = (task / nq) % NH;
            const int sq = task / (nq * NH);
            const int row_i = sq * L + i;
            float* qp = qkv + (size_t)row_i * B3_LDQ + h * NDH;
            HV q, o, kv;
            q.load(qp, NDH);
            o.zero();
            float m = -INFINITY, l = 0.f;
            const float* kbase = qkv + (size_t)(sq * L) * B3_LDQ + B3_I + h * NDH;
            int j = 0;
            for (; j + CORE_UNROLL <= L; j += CORE_UNROLL) {
                HV kk[CORE_UNROLL], vv[CORE_UNROLL];
#pragma unroll
                for (int u = 0; u < CORE_UNROLL; ++u) {
                    const float* kp = kbase + (size_t)(j + u) * B3_LDQ;
                    kk[u].load(kp, NDH);
                    vv[u].load(kp + B3_I, NDH);
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
                kv.load(kp, NDH);
                const float sv = q.dot(kv) * sl2;
                const float mn = fmaxf(m, sv);
                const float corr = rat_exp2(m - mn);
                const float p = rat_exp2(sv - mn);
                l = l * corr + p;
                kv.load(kp + B3_I, NDH);
                o.scale_axpy(corr, p, kv);
                m = mn;
            }
            const float inv = 1.0f / l;
            o.store(qp, NDH, inv);
            lse_s[row_i * NH + h] = m + rat_log2(l);
        }
        if (QSUB && nq < L) {                                    // the positions nobody asked for: O = 0, lse = 0 (defined, never used)
            for (int e = threadIdx.x; e < rows * NH; e += ATT_THREADS) {
                const int r = e / NH, h = e - r * NH;
                if (r % L < nq) continue;
                HV z;
                z.zero();
                z.store(qkv + (size_t)r * B3_LDQ + h * NDH, NDH, 1.0f);
                lse_s[e] = 0.f;
            }
        }
        __syncthreads();
        RAT_PROF_MARK(2);
        // O (the Q columns of the valid rows; padding rows are exact zeros) -> planes, and -> o_save for the backward: whole 320-byte
        // rows in 16-byte pieces with the non-temporal hint (round 4; before, every core lane stored its head's 40 bytes in five
        // scattered 8-byte stores at the end of its key loop).  lse_save leaves the same way, from the LDS copy.
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
        if (lse_save != nullptr && (int)threadIdx.x < (NH / 4) * ATT_ROWS) {       // a row's NH values as NH / 4 16-byte pieces
            const int r = threadIdx.x / (NH / 4), part = threadIdx.x % (NH / 4);
            const int64_t tok = rowtok[r];
            if (tok >= 0) rat_st4_stream(lse_save + tok * NH + 4 * part, *reinterpret_cast<const float4*>(lse_s + r * NH + 4 * part));
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
        b3_gemm_rows<3>(op, wout_lds, B3_D / 16, [&](int mt, int nt, const f32x4& acc) {
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
    constexpr int SLD = 16 * NIT + 1, SCR = 16 * SLD + 16;          // wave-private [16 queries][16 NIT keys (+ 1)] tile + 16 deltas
    const int w = rat_wave(), l = rat_lane(), g = l >> 4, m = l & 15;
    // (three tiles — 33 ... 48 tokens — in this all-at-once form were built in round 6 and lost to the VALU passes: 64-66 spilled VGPRs;
    //  b3_bwd_core_mfma_kt below serves them.  tools/experiments/attn_bwd_core_three_tiles_at_once.txt)
    float* scr = scratch + w * SCR;
    float* dl = scr + 16 * SLD;
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
                // (round 6: the selects on the K / V / Q operands of the B side went — the k columns 10, 11 are masked on the A side; key rows
                //  past the sequence only reach (i, j) pairs whose p is forced to 0 below; operand "columns" m >= 10 only reach output
                //  columns c >= 10, which are never stored; and what those loads can read past the Q|K|V tile is the O tile: finite.  The
                //  dO-side selects STAY: rows past the dO tile are the never-written tail of the kernel's LDS)
                const bool okc = 4 * ks + g < B3_DH, oki = okc && m < irows;
                const float xq = oki ? aq[ks] : 0.f, xo = oki ? ao[ks] : 0.f;
#pragma unroll
                for (int jt = 0; jt < NIT; ++jt) {
                    aS[jt] = RAT_MFMA16(xq, bk[ks][jt], aS[jt]);
                    aP[jt] = RAT_MFMA16(xo, bv[ks][jt], aP[jt]);
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
                        scr[(4 * g + r) * SLD + 16 * jt + m] = p;
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
                    for (int jt = 0; jt < NIT; ++jt) ap[ks][jt] = scr[ii * SLD + 16 * jt + m];
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
                for (int r = 0; r < 4; ++r) scr[(4 * g + r) * SLD + 16 * jt + m] = dS[jt][r];
            RAT_WAVE_FENCE();
            // dK[j][c] += sum_i dS[i][j] Q[i][c]   (A = dS^T, B = Q);   dQ[i][c] = sum_j dS[i][j] K[j][c]   (A = dS, B = K)
            {
                float bq[4], at[4][NIT];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int ii = 4 * ks + g;
                    bq[ks] = qkv[(size_t)(r0 + i0 + ii) * B3_LDQ + cq + mc];
#pragma unroll
                    for (int jt = 0; jt < NIT; ++jt) at[ks][jt] = scr[ii * SLD + 16 * jt + m];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float b = bq[ks];
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
                    as[ks] = scr[m * SLD + jj];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) adQ = RAT_MFMA16(as[ks], bkk[ks], adQ);
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
// ---- the same core with the KEY tiles as the inner loop (round 6) ------------------------------------------------------------------------------
// b3_bwd_core_mfma keeps S, dP and dS of ALL key tiles of a query tile in registers at once (3 NIT accumulator quads + 6 NIT operands on
// top of the 2 NIT dK / dV accumulators): beside attn_bwd3_kernel's 48 persistent weight-gradient VGPRs that spills 18 registers at NIT = 2
// and 64-66 at NIT = 3.  Here one (query tile, key tile) pair is finished — S, dP, p, dS, dV += P^T dO, dK += dS^T Q, dQ += dS K — before the
// next key tile starts: one accumulator quad each for S / dP / dS, a [16][17] wave-private tile whatever NIT is (9 KB for eight waves), the
// query tile's operands loaded once per query tile.  Three wave fences per pair of tiles instead of four per query tile.
template <int NIT>
__device__ __forceinline__ void b3_bwd_core_mfma_kt(float* qkv, float* ob, const float* dob, const float* lses, float* scratch, int L, int nsq,
                                                    float scale) {
    constexpr int SLD = 17, SCR = 16 * SLD + 16;
    const int w = rat_wave(), l = rat_lane(), g = l >> 4, m = l & 15;
    float* scr = scratch + w * SCR;
    float* dl = scr + 16 * SLD;
    const float sl2 = scale * RAT_LOG2E;
    const int npairs = nsq * B3_H;
    const bool cm = m < B3_DH;                                // this lane's column of a [.][dim_head] operand exists
    for (int pair = w; pair < npairs; pair += ATT_WAVES) {
        const int h = pair % B3_H, sq = pair / B3_H;
        const int r0 = sq * L, cq = h * B3_DH, ck = B3_I + h * B3_DH, cv = 2 * B3_I + h * B3_DH;
        f32x4 adK[NIT], adV[NIT];
#pragma unroll
        for (int jt = 0; jt < NIT; ++jt) adK[jt] = adV[jt] = rat_zero4();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i0 = 16 * it;
            if (i0 >= L) break;                               // (wave-uniform)
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
            // the query tile's operands, once: A of S / dP (lane: [row m][k g]) and B of dK / dV (lane: [k = query 4 ks + g][col m])
            float aq[3], ao[3], bq[4], bdo[4];
            const int ri = r0 + i0 + m;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int cc = 4 * ks + g;
                const bool oki = cc < B3_DH && m < irows;     // (which selects stay and which went: see b3_bwd_core_mfma)
                const float q_ = qkv[(size_t)ri * B3_LDQ + cq + cc], o_ = dob[(size_t)ri * B3_LDT + cq + cc];
                aq[ks] = oki ? q_ : 0.f;
                ao[ks] = oki ? o_ : 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ii = 4 * ks + g;
                bq[ks] = qkv[(size_t)(r0 + i0 + ii) * B3_LDQ + cq + m];
                const float o_ = dob[(size_t)(r0 + i0 + ii) * B3_LDT + cq + m];
                bdo[ks] = (cm && ii < irows) ? o_ : 0.f;
            }
            RAT_WAVE_FENCE();                                 // (dl is written above, read below)
            float lse4[4], d4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lse4[r] = lses[(r0 + i0 + 4 * g + r) * B3_H + h];
                d4[r] = dl[4 * g + r];
            }
            f32x4 adQ = rat_zero4();
#pragma unroll
            for (int jt = 0; jt < NIT; ++jt) {
                const int j0 = 16 * jt;
                if (j0 >= L) break;                           // (wave-uniform)
                float bk[3], bv[3], bkk[4];
                const int rj = r0 + j0 + m;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const int cc = 4 * ks + g;
                    bk[ks] = qkv[(size_t)rj * B3_LDQ + ck + cc];
                    bv[ks] = qkv[(size_t)rj * B3_LDQ + cv + cc];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int jj = j0 + 4 * ks + g;
                    bkk[ks] = qkv[(size_t)(r0 + jj) * B3_LDQ + ck + m];
                }
                f32x4 aS = rat_zero4(), aP = rat_zero4();
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    aS = RAT_MFMA16(aq[ks], bk[ks], aS);
                    aP = RAT_MFMA16(ao[ks], bv[ks], aP);
                }
                // p and dS on the accumulators (C layout: column m = key, rows 4 g + r = query)
                f32x4 dS;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = 4 * g + r < irows && j0 + m < L;
                    const float p = ok ? rat_exp2(aS[r] * sl2 - lse4[r]) : 0.f;
                    scr[(4 * g + r) * SLD + m] = p;
                    dS[r] = p * (aP[r] - d4[r]);
                }
                RAT_WAVE_FENCE();
                {                                             // dV[j][c] += sum_i P[i][j] dO[i][c]   (A = P^T from the tile, B = dO)
                    float ap[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) ap[ks] = scr[(4 * ks + g) * SLD + m];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) adV[jt] = RAT_MFMA16(ap[ks], bdo[ks], adV[jt]);
                }
                RAT_WAVE_FENCE();
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(4 * g + r) * SLD + m] = dS[r];
                RAT_WAVE_FENCE();
                {                                             // dK[j][c] += sum_i dS[i][j] Q[i][c] (A = dS^T, B = Q);  dQ[i][c] += sum_j dS[i][j] K[j][c] (A = dS, B = K)
                    float at[4], as[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        at[ks] = scr[(4 * ks + g) * SLD + m];
                        as[ks] = scr[m * SLD + 4 * ks + g];
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        adK[jt] = RAT_MFMA16(at[ks], bq[ks], adK[jt]);
                        adQ = RAT_MFMA16(as[ks], bkk[ks], adQ);
                    }
                }
                RAT_WAVE_FENCE();                             // (the next key tile overwrites the private tile)
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
template <bool EX, bool QSUB = false, bool DPAD = false, bool PH = false, int MC = 0, int NH = B3_H>
__global__ void __launch_bounds__(ATT_THREADS) attn_bwd3_kernel(AttnArgs a, Attn3W W) {
    constexpr int NDH = B3_I / NH;
    static_assert(NH == B3_H || (NH == 4 && MC == 0 && !DPAD), "the matrix core and DPAD are written for 8 heads x 10");
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
    float* dlt = lses + ATT_ROWS * NH;                                     // [64][8]
    int64_t* const rowtok0 = reinterpret_cast<int64_t*>(dlt + ATT_ROWS * NH);
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
        map_rows(a, blockIdx.x, rowtok0, nsq0, rows0);
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
            float4 d0 = b3_ld4(a.dy, po), d1 = b3_ld4(a.dy, po + 16u);
            fo.issue(a.o_save, rowtok);
            float lsen = b3_ld1(a.lse_save, (uint32_t)(tok_own >= 0 ? tok_own : 0) * (uint32_t)(NH * 4) + 4u * (sub < NH ? sub : 0));
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
            if (sub < NH) lses[r_own * NH + sub] = lsen;                     // (NH = 8: 512 threads = 64 rows x 8 heads)
        }
        if (chunk + gridDim.x < a.nchunks) {
            int nsq1, rows1;
            map_rows(a, chunk + gridDim.x, (rowtok0 + (parity ^ 1) * ATT_ROWS), nsq1, rows1);
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
        typedef HeadVec<NDH> HV;
        const int nq = QSUB ? a.nq : L;
        const int ntasks = nsq * NH * L, nqtasks = QSUB ? nsq * NH * nq : ntasks;
        const float sl2 = a.scale * RAT_LOG2E;
        if (MC >= 10) b3_bwd_core_mfma_kt<(MC >= 10 ? MC % 10 : 1)>(qkv, ob, dob, lses, dxn, L, nsq, a.scale);   // MC = 10 + NIT: key tiles as the inner loop
        else if (MC) b3_bwd_core_mfma<(MC > 0 && MC < 10 ? MC : 1)>(qkv, ob, dob, lses, dxn, L, nsq, a.scale);   // (the dy planes are dead since P2b: the wave-private tiles go there)
        for (int task = threadIdx.x; !MC && task < nqtasks; task += ATT_THREADS) {
            const int i = task % nq;
            const int h = (task / nq) % NH;
            const int sq = task / (nq * NH);
            const int row_i = sq * L + i;
            const int ho = h * NDH;
            float* opp = ob + (size_t)row_i * B3_LDT + ho;
            HV q, go, dq, kv;
            q.load(qkv + (size_t)row_i * B3_LDQ + ho, NDH);
            go.load(dob + (size_t)row_i * B3_LDT + ho, NDH);
            kv.load(opp, NDH);
            const float delta = go.dot(kv);
            dq.zero();
            dlt[row_i * NH + h] = delta;
            const float lse = lses[row_i * NH + h];
            const float* kbase = qkv + (size_t)(sq * L) * B3_LDQ + B3_I + ho;
            float* const prow = PH ? dxn + ((sq * NH + h) * L + i) * L : nullptr;     // P[(sequence, head)][query i][key j]
            for (int j = 0; j < L; ++j) {
                const float* kp = kbase + (size_t)j * B3_LDQ;
                kv.load(kp + B3_I, NDH);
                const float dp = go.dot(kv);
                kv.load(kp, NDH);
                const float p = rat_exp2(q.dot(kv) * sl2 - lse);
                if (PH) prow[j] = p;
                dq.axpy(p * (dp - delta), kv);
            }
            dq.store(opp, NDH, a.scale);
        }
        if (QSUB && nq < L) {
            for (int e = threadIdx.x; e < nsq * L * NH; e += ATT_THREADS) {
                const int r = e / NH, h = e - r * NH;
                if (r % L < nq) continue;
                HV z;
                z.zero();
                z.store(ob + (size_t)r * B3_LDT + h * NDH, NDH, 1.0f);    // dQ of a position that is no query
            }
        }
        __syncthreads();
        RAT_PROF_MARK(4);
        for (int task = threadIdx.x; !MC && task < ntasks; task += ATT_THREADS) {
            const int j = task % L;
            const int h = (task / L) % NH;
            const int sq = task / (L * NH);
            const int ho = h * NDH;
            float* kp = qkv + (size_t)(sq * L + j) * B3_LDQ + B3_I + ho;
            HV kk, vv, dk, dv, t;
            kk.load(kp, NDH);
            vv.load(kp + B3_I, NDH);
            dk.zero();
            dv.zero();
            for (int i = 0; i < nq; ++i) {
                const int row_i = sq * L + i;
                t.load(dob + (size_t)row_i * B3_LDT + ho, NDH);
                const float dp = t.dot(vv);
                const float delta = dlt[row_i * NH + h];
                HV qv;
                qv.load(qkv + (size_t)row_i * B3_LDQ + ho, NDH);
                float p;
                if (PH) p = dxn[((sq * NH + h) * L + i) * L + j];           // consecutive lanes = consecutive keys: conflict-free
                else p = rat_exp2(qv.dot(kk) * sl2 - lses[row_i * NH + h]);
                dv.axpy(p, t);
                dk.axpy(p * (dp - delta), qv);
            }
            dk.store(kp, NDH, a.scale);
            dv.store(kp + B3_I, NDH, 1.0f);
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
