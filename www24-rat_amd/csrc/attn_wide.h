// attn_wide.h — part of attn.hip's translation unit (included inside its anonymous namespace, after the shared helpers): the
// exact-fp32 kernels that run a WIDE-HEAD layer at a small embedding dimension (the shipped Tmall geometry) in one launch per direction.
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
template <int WH>
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
            const bool lt = (int)threadIdx.x < ATT_ROWS * WH;        // (WH = 8: every thread owns one lse element; 4: the first 256)
            const int64_t tl = lt ? rowtok[threadIdx.x / WH] : -1;
#pragma unroll
            for (int it = 0; it < NO; ++it) {
                const int e = threadIdx.x + ATT_THREADS * it;
                o[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tok[it] >= 0) o[it] = *reinterpret_cast<const float4*>(o_g + tok[it] * WG_I + 4 * (e % W4));
            }
            lse = tl >= 0 ? lse_g[tl * WH + threadIdx.x % WH] : 0.f;
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
            if ((int)threadIdx.x < ATT_ROWS * WH) lses[threadIdx.x] = lse;
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
static_assert(ATT_ROWS * WG_H == ATT_THREADS, "at most one lse element per thread");
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
// WH (round 6): heads per group, 8 (x 10) or 4 (x 20) — RAT_m3 runs heads / 2 heads of width 2 dim_head (RAT_m3.py:181); a group's inner width is 80
// either way, so the weight slices, the GEMM phases and the LDS map are the same; the VALU core slices Q | K | V differently and lse is WH wide
template <int GD, int WH = WG_H>
__global__ void __launch_bounds__(ATT_THREADS, 4) attn_fwd_wide_kernel(AttnArgs a) {   // (4 waves per SIMD: two work-groups per CU)
    constexpr int WDH = WG_I / WH;
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
                WideGroupFetch<WH> gf;
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
            typedef HeadVec<WDH> HV;
            float* const o_save = a.o_save != nullptr ? a.o_save + (int64_t)grp * a.group_tok * WG_I : nullptr;
            float* const lse_save = a.lse_save != nullptr ? a.lse_save + (int64_t)grp * a.group_tok * WH : nullptr;
            const int ntasks = nsq * WH * L;
            const float sl2 = a.scale * RAT_LOG2E;
            for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
                const int i = task % L;
                const int h = (task / L) % WH;
                const int sq = task / (L * WH);
                const int row_i = sq * L + i;
                float* qp = qkv + (size_t)row_i * WG_LDQ + h * WDH;
                HV q, o, kv;
                q.load(qp, WDH);
                o.zero();
                float m = -INFINITY, l = 0.f;
                const float* kbase = qkv + (size_t)(sq * L) * WG_LDQ + WG_I + h * WDH;
                int j = 0;
                for (; j + CORE_UNROLL <= L; j += CORE_UNROLL) {
                    HV kk[CORE_UNROLL], vv[CORE_UNROLL];
#pragma unroll
                    for (int u = 0; u < CORE_UNROLL; ++u) {
                        const float* kp = kbase + (size_t)(j + u) * WG_LDQ;
                        kk[u].load(kp, WDH);
                        vv[u].load(kp + WG_I, WDH);
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
                    kv.load(kp, WDH);
                    const float s = q.dot(kv) * sl2;
                    const float mn = fmaxf(m, s);
                    const float corr = rat_exp2(m - mn);
                    const float p = rat_exp2(s - mn);
                    l = l * corr + p;
                    kv.load(kp + WG_I, WDH);
                    o.scale_axpy(corr, p, kv);
                    m = mn;
                }
                const float inv = 1.0f / l;
                o.store(qp, WDH, inv);
                const int64_t tok = rowtok[row_i];
                if (o_save != nullptr) o.store(o_save + tok * WG_I + h * WDH, WDH, inv);
                if (lse_save != nullptr) lse_save[tok * WH + h] = m + rat_log2(l);
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

template <int GD, int WH = WG_H>
__global__ void __launch_bounds__(ATT_THREADS) attn_bwd_wide_kernel(AttnArgs a) {
    constexpr int WDH = WG_I / WH;
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
    float* dlt = lses + (size_t)ATT_ROWS * WH;                 // [64][8]
    int64_t* const rowtok0 = reinterpret_cast<int64_t*>(dlt + (size_t)ATT_ROWS * WH);
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
            const float* const lse_g = a.lse_save + (int64_t)grp * a.group_tok * WH;
            if (fast) {                                          // O, lse and the group's weight slices: one round trip
                WideGroupFetch<WH> gf;
                gf.issue(o_g, lse_g, rowtok, a.w_qkv, a.w_out, itot, grp, true);
                gf.stash(ob, lses, wq_s, wo_s, true);
            } else {
                load_rows(ob, WG_LDT, o_g, rowtok, WG_I, false);
                if (lds_w) wide_stage_wq(wq_s, a.w_qkv, itot, grp, D);
                wide_stage_wo(wo_s, a.w_out, itot, grp, D);
                for (int e = threadIdx.x; e < ATT_ROWS * WH; e += ATT_THREADS) {
                    const int64_t tok = rowtok[e / WH];
                    lses[e] = tok >= 0 ? lse_g[tok * WH + e % WH] : 0.f;
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
            typedef HeadVec<WDH> HV;
            const int ntasks = nsq * WH * L;
            const float sl2 = a.scale * RAT_LOG2E;
            {   // both passes touch LDS only: the lines this block loads next travel HBM -> L2 meanwhile
                int t = threadIdx.x;
                if (grp + 1 < G) {
                    pf += wide_touch(rowtok, t, o_g + a.group_tok * WG_I, WG_I);
                    pf += wide_touch(rowtok, t, lse_g + a.group_tok * WH, WH);
                } else if (more) {
                    pf += wide_touch(rowtok_next, t, a.x, D);
                    pf += wide_touch(rowtok_next, t, a.dy, D);
                    pf += wide_touch(rowtok_next, t, a.o_save, WG_I);
                    pf += wide_touch(rowtok_next, t, a.lse_save, WH);
                }
            }
            for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
                const int i = task % L;
                const int h = (task / L) % WH;
                const int sq = task / (L * WH);
                const int row_i = sq * L + i;
                const int ho = h * WDH;
                float* op = ob + (size_t)row_i * WG_LDT + ho;
                HV q, go, dq, kv;
                q.load(qkv + (size_t)row_i * WG_LDQ + ho, WDH);
                go.load(dob + (size_t)row_i * WG_LDT + ho, WDH);
                kv.load(op, WDH);
                const float delta = go.dot(kv);
                dq.zero();
                dlt[row_i * WH + h] = delta;
                const float lse = lses[row_i * WH + h];
                const float* kbase = qkv + (size_t)(sq * L) * WG_LDQ + WG_I + ho;
                for (int j = 0; j < L; ++j) {
                    const float* kp = kbase + (size_t)j * WG_LDQ;
                    kv.load(kp + WG_I, WDH);
                    const float dp = go.dot(kv);
                    kv.load(kp, WDH);
                    const float p = rat_exp2(q.dot(kv) * sl2 - lse);
                    dq.axpy(p * (dp - delta), kv);
                }
                dq.store(op, WDH, a.scale);
            }
            __syncthreads();
            RAT_PROF_MARK(3);
            for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
                const int j = task % L;
                const int h = (task / L) % WH;
                const int sq = task / (L * WH);
                const int ho = h * WDH;
                float* kp = qkv + (size_t)(sq * L + j) * WG_LDQ + WG_I + ho;
                HV kk, vv, dk, dv, t;
                kk.load(kp, WDH);
                vv.load(kp + WG_I, WDH);
                dk.zero();
                dv.zero();
                for (int i = 0; i < L; ++i) {
                    const int row_i = sq * L + i;
                    t.load(dob + (size_t)row_i * WG_LDT + ho, WDH);
                    const float dp = t.dot(vv);
                    const float lse = lses[row_i * WH + h], delta = dlt[row_i * WH + h];
                    HV qv;
                    qv.load(qkv + (size_t)row_i * WG_LDQ + ho, WDH);
                    const float p = rat_exp2(qv.dot(kk) * sl2 - lse);
                    dv.axpy(p, t);
                    dk.axpy(p * (dp - delta), qv);
                }
                dk.store(kp, WDH, a.scale);
                dv.store(kp + WG_I, WDH, 1.0f);
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
