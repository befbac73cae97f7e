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
// The projections run on v_mfma_f32_16x16x4_f32 with the weights streamed from L2 as B operands; the
// (L x L x dh) attention core of a sequence-head is far too small and ragged for a 16x16 MFMA tile at fp32
// (fp32 MFMA peak == fp32 VALU peak on gfx950, and an 11x11x10 problem fills 39 % of a padded tile), so it
// runs on the VALU, one lane per (sequence, head, query) with an online softmax.
// Backward recomputes LayerNorm and QKV from x, re-derives P from the saved log-sum-exp, and keeps every
// weight gradient in MFMA accumulators across the work-group's whole chunk loop (written once per launch
// to a per-work-group slab, then summed in fixed order => deterministic).
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int ATT_THREADS = 512;
constexpr int ATT_WAVES = ATT_THREADS / 64;
constexpr int ATT_ROWS = 64;
constexpr int DH_MAX = 16;        // dim_head <= 16 (every shipped config uses 10)
constexpr int QSLOTS = 8;         // persistent dW_qkv tiles per wave  (3I16/16 * D16/16 <= 64)
constexpr int OSLOTS = 4;         // persistent dW_out tiles per wave  (D16/16 * I16/16 <= 32)
constexpr int LN_COLS = 16;       // columns per lane in the 8-lanes-per-row LayerNorm passes (d <= 128)

struct AttnArgs {
    const float* x;
    const float* dy;
    float* y;            // forward output / backward dx
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
    int d, heads, dh;
    float eps, scale;
    int vec_x, vec_wqkv, vec_wout;
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
    size_t fwd_smem() const { return (size_t)ATT_ROWS * (ldx + ldq) * 4 + ATT_ROWS * 8; }
    size_t bwd_smem(int heads) const {
        return (size_t)ATT_ROWS * (2 * ldx + ldq + 2 * ldt) * 4 + (size_t)ATT_ROWS * (2 + 2 * heads) * 4 + ATT_ROWS * 8;
    }
    int64_t slab_floats() const { return (int64_t)Q3 * D + (int64_t)D * I + 3 * (int64_t)D; }
};

__device__ __forceinline__ int64_t seq_token(const AttnArgs& a, int64_t q, int p) {
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

// load D floats per row from a token-indexed global array into an LDS tile (padding rows -> 0)
__device__ __forceinline__ void load_rows(float* tile, int ld, const float* src, const int64_t* rowtok, int width,
                                          bool vec) {
    if (vec) {
        const int w4 = width >> 2;
        for (int e = threadIdx.x; e < ATT_ROWS * w4; e += ATT_THREADS) {
            const int r = e / w4, c4 = e - r * w4;
            const int64_t tok = rowtok[r];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tok >= 0) v = *reinterpret_cast<const float4*>(src + tok * width + 4 * c4);
            *reinterpret_cast<float4*>(tile + (size_t)r * ld + 4 * c4) = v;
        }
    } else {
        for (int e = threadIdx.x; e < ATT_ROWS * width; e += ATT_THREADS) {
            const int r = e / width, c = e - r * width;
            const int64_t tok = rowtok[r];
            tile[(size_t)r * ld + c] = tok >= 0 ? src[tok * width + c] : 0.f;
        }
    }
}

__device__ __forceinline__ void zero_cols(float* tile, int ld, int c0) {   // tile[:, c0:ld] = 0
    const int w = ld - c0;
    for (int e = threadIdx.x; e < ATT_ROWS * w; e += ATT_THREADS) tile[(size_t)(e / w) * ld + c0 + e % w] = 0.f;
}

// in-place LayerNorm of the valid rows of xs (8 lanes per row); optionally keeps mean / rstd
__device__ __forceinline__ void layer_norm_rows(float* xs, int ld, int D, int rows, const float* g, const float* b,
                                                float eps, float* mu_out, float* rs_out) {
    const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
    float* xr = xs + (size_t)r * ld;
    float s = 0.f;
    for (int c = sub; c < D; c += 8) s += xr[c];
    const float mean = rat_group_sum<8>(s) / (float)D;
    float v = 0.f;
    for (int c = sub; c < D; c += 8) {
        const float t = xr[c] - mean;
        v += t * t;
    }
    const float rstd = 1.0f / sqrtf(rat_group_sum<8>(v) / (float)D + eps);
    if (r < rows)
        for (int c = sub; c < D; c += 8) xr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
    if (mu_out != nullptr && sub == 0) {
        mu_out[r] = mean;
        rs_out[r] = rstd;
    }
}

// qkv[rows][0:3I] = xs W_qkv^T   (each wave: all M tiles x 2 N tiles per task)
__device__ __forceinline__ void qkv_projection(const AttnArgs& a, const AttnGeom& g, const float* xs, float* qkv,
                                               int mt_valid) {
    const int ntn = g.Q16 / 16, ntasks = (ntn + 1) / 2;
    const RatLdsRows A{xs, g.ldx};
    const RatGlobalWnk Bw{a.w_qkv, g.Q3, g.D, g.D, a.vec_wqkv != 0};
    for (int task = rat_wave(); task < ntasks; task += ATT_WAVES) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = rat_zero4();
        const int nt0 = task * 2;
        const int nbv = ntn - nt0 < 2 ? ntn - nt0 : 2;
        rat_wave_gemm<4, 2>(acc, A, Bw, 0, nt0, mt_valid, nbv, g.D16 / 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (i < mt_valid && j < nbv) {
                    const int col = rat_acc_col(nt0 + j);
                    if (col < g.Q3)
#pragma unroll
                        for (int r = 0; r < 4; ++r) qkv[(size_t)rat_acc_row(i, r) * g.ldq + col] = acc[i][j][r];
                }
    }
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(ATT_THREADS) attn_fwd_kernel(AttnArgs a) {
    RAT_DYN_SMEM(smem);
    const AttnGeom g(a.d, a.heads, a.dh);
    float* xs = reinterpret_cast<float*>(smem);
    float* qkv = xs + (size_t)ATT_ROWS * g.ldx;
    int64_t* rowtok = reinterpret_cast<int64_t*>(qkv + (size_t)ATT_ROWS * g.ldq);
    const int L = a.L, D = g.D, I = g.I, dh = a.dh;

    zero_cols(xs, g.ldx, D);
    zero_cols(qkv, g.ldq, g.Q3);
    __syncthreads();

    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x) {
        int nsq, rows;
        map_rows(a, chunk, rowtok, nsq, rows);
        __syncthreads();
        load_rows(xs, g.ldx, a.x, rowtok, D, a.vec_x != 0);
        __syncthreads();
        layer_norm_rows(xs, g.ldx, D, rows, a.ln_g, a.ln_b, a.eps, nullptr, nullptr);
        __syncthreads();
        const int mt_valid = (rows + 15) / 16;
        qkv_projection(a, g, xs, qkv, mt_valid);
        __syncthreads();

        // softmax(Q K^T * scale) V, one lane per (sequence, head, query); result replaces Q in place
        const int ntasks = nsq * a.heads * L;
        for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
            const int i = task % L;
            const int h = (task / L) % a.heads;
            const int sq = task / (L * a.heads);
            const int row_i = sq * L + i;
            float* qp = qkv + (size_t)row_i * g.ldq + h * dh;
            float q[DH_MAX], o[DH_MAX];
#pragma unroll
            for (int c = 0; c < DH_MAX; ++c) {
                q[c] = c < dh ? qp[c] : 0.f;
                o[c] = 0.f;
            }
            float m = -INFINITY, l = 0.f;
            for (int j = 0; j < L; ++j) {
                const float* kp = qkv + (size_t)(sq * L + j) * g.ldq + I + h * dh;
                const float* vp = kp + I;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < DH_MAX; ++c)
                    if (c < dh) s = fmaf(q[c], kp[c], s);
                s *= a.scale;
                const float mn = fmaxf(m, s);
                const float corr = expf(m - mn);
                const float p = expf(s - mn);
                l = l * corr + p;
#pragma unroll
                for (int c = 0; c < DH_MAX; ++c)
                    if (c < dh) o[c] = fmaf(p, vp[c], o[c] * corr);
                m = mn;
            }
            const float inv = 1.0f / l;
            const int64_t tok = rowtok[row_i];
#pragma unroll
            for (int c = 0; c < DH_MAX; ++c)
                if (c < dh) {
                    const float ov = o[c] * inv;
                    qp[c] = ov;
                    if (a.o_save != nullptr) a.o_save[tok * I + h * dh + c] = ov;
                }
            if (a.lse_save != nullptr) a.lse_save[tok * a.heads + h] = m + logf(l);
        }
        __syncthreads();

        // y = O W_out^T + b_out + x   (or y = O + x when Attention has no output projection)
        if (a.w_out != nullptr) {
            const int ntn = g.D16 / 16, mblocks = (mt_valid + 1) / 2, ntasks2 = mblocks * ntn;
            const RatLdsRows A{qkv, g.ldq};
            const RatGlobalWnk Bw{a.w_out, D, I, I, a.vec_wout != 0};
            for (int task = rat_wave(); task < ntasks2; task += ATT_WAVES) {
                const int mt0 = (task / ntn) * 2, nt = task % ntn;
                const int mtv = mt_valid - mt0 < 2 ? mt_valid - mt0 : 2;
                f32x4 acc[2][1];
                acc[0][0] = acc[1][0] = rat_zero4();
                rat_wave_gemm<2, 1>(acc, A, Bw, mt0, nt, mtv, 1, g.I16 / 16);
                const int col = rat_acc_col(nt);
                if (col < D) {
                    const float bias = a.b_out[col];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (i < mtv)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = rat_acc_row(mt0 + i, r);
                                if (row < rows) {
                                    const int64_t tok = rowtok[row];
                                    a.y[tok * D + col] = acc[i][0][r] + bias + a.x[tok * D + col];
                                }
                            }
                }
            }
        } else {
            for (int e = threadIdx.x; e < rows * D; e += ATT_THREADS) {
                const int r = e / D, c = e - r * D;
                const int64_t tok = rowtok[r];
                a.y[tok * D + c] = qkv[(size_t)r * g.ldq + c] + a.x[tok * D + c];
            }
        }
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------------------------- backward
__global__ void __launch_bounds__(ATT_THREADS) attn_bwd_kernel(AttnArgs a) {
    RAT_DYN_SMEM(smem);
    const AttnGeom g(a.d, a.heads, a.dh);
    const int L = a.L, D = g.D, I = g.I, dh = a.dh, H = a.heads;
    float* xs = reinterpret_cast<float*>(smem);                 // [64][ldx]  LayerNorm(x)
    float* dys = xs + (size_t)ATT_ROWS * g.ldx;                 // [64][ldx]  dL/dy
    float* qkv = dys + (size_t)ATT_ROWS * g.ldx;                // [64][ldq]  Q|K|V, later dQ|dK|dV
    float* ob = qkv + (size_t)ATT_ROWS * g.ldq;                 // [64][ldt]  O, later dQ
    float* dob = ob + (size_t)ATT_ROWS * g.ldt;                 // [64][ldt]  dO, later d(LayerNorm out)
    float* mu = dob + (size_t)ATT_ROWS * g.ldt;                 // [64]
    float* rs = mu + ATT_ROWS;                                  // [64]
    float* lses = rs + ATT_ROWS;                                // [64][H]
    float* dlt = lses + (size_t)ATT_ROWS * H;                   // [64][H]   rowsum(dO * O)
    int64_t* rowtok = reinterpret_cast<int64_t*>(dlt + (size_t)ATT_ROWS * H);
    const bool has_out = a.w_out != nullptr;

    // persistent parameter-gradient accumulators
    f32x4 accq[QSLOTS], acco[OSLOTS];
#pragma unroll
    for (int s = 0; s < QSLOTS; ++s) accq[s] = rat_zero4();
#pragma unroll
    for (int s = 0; s < OSLOTS; ++s) acco[s] = rat_zero4();
    float dgam[LN_COLS], dbet[LN_COLS];
#pragma unroll
    for (int k = 0; k < LN_COLS; ++k) dgam[k] = dbet[k] = 0.f;
    float dbo = 0.f;
    const int q_tn = g.D16 / 16, q_tiles = (g.Q16 / 16) * q_tn;       // dW_qkv tiles: (3I16/16) x (D16/16)
    const int o_tn = g.I16 / 16, o_tiles = (g.D16 / 16) * o_tn;       // dW_out tiles: (D16/16) x (I16/16)

    zero_cols(xs, g.ldx, D);
    zero_cols(dys, g.ldx, D);
    zero_cols(qkv, g.ldq, g.Q3);
    zero_cols(ob, g.ldt, 0);
    zero_cols(dob, g.ldt, 0);
    __syncthreads();

    for (int64_t chunk = blockIdx.x; chunk < a.nchunks; chunk += gridDim.x) {
        int nsq, rows;
        map_rows(a, chunk, rowtok, nsq, rows);
        __syncthreads();
        load_rows(xs, g.ldx, a.x, rowtok, D, a.vec_x != 0);
        load_rows(dys, g.ldx, a.dy, rowtok, D, a.vec_x != 0);
        load_rows(ob, g.ldt, a.o_save, rowtok, I, (I % 4) == 0 && a.vec_x != 0);
        for (int e = threadIdx.x; e < ATT_ROWS * H; e += ATT_THREADS) {
            const int64_t tok = rowtok[e / H];
            lses[e] = tok >= 0 ? a.lse_save[tok * H + e % H] : 0.f;
        }
        __syncthreads();
        layer_norm_rows(xs, g.ldx, D, rows, a.ln_g, a.ln_b, a.eps, mu, rs);
        __syncthreads();
        const int mt_valid = (rows + 15) / 16;

        // (1) recompute Q|K|V
        qkv_projection(a, g, xs, qkv, mt_valid);
        // (2) dO = dy W_out  (dob[rows][0:I])
        if (has_out) {
            const int ntn = g.I16 / 16, mblocks = (mt_valid + 1) / 2, ntasks = mblocks * ntn;
            const RatLdsRows A{dys, g.ldx};
            const RatGlobalWkn Bw{a.w_out, D, I, I};
            for (int task = rat_wave(); task < ntasks; task += ATT_WAVES) {
                const int mt0 = (task / ntn) * 2, nt = task % ntn;
                const int mtv = mt_valid - mt0 < 2 ? mt_valid - mt0 : 2;
                f32x4 acc[2][1];
                acc[0][0] = acc[1][0] = rat_zero4();
                rat_wave_gemm<2, 1>(acc, A, Bw, mt0, nt, mtv, 1, g.D16 / 16);
                const int col = rat_acc_col(nt);
                if (col < I)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (i < mtv)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dob[(size_t)rat_acc_row(mt0 + i, r) * g.ldt + col] = acc[i][0][r];
            }
            // (3) dW_out += dy^T O ; db_out += colsum(dy)
            const RatLdsCols At{dys, g.ldx};
            const RatLdsCols Bt{ob, g.ldt};
#pragma unroll
            for (int s = 0; s < OSLOTS; ++s) {
                const int id = rat_wave() + ATT_WAVES * s;
                if (id < o_tiles) acco[s] = rat_wave_gemm1(acco[s], At, Bt, id / o_tn, id % o_tn, mt_valid);
            }
            if (threadIdx.x < D) {
                float sacc = 0.f;
                for (int r = 0; r < rows; ++r) sacc += dys[(size_t)r * g.ldx + threadIdx.x];
                dbo += sacc;
            }
        } else {
            for (int e = threadIdx.x; e < ATT_ROWS * D; e += ATT_THREADS) {
                const int r = e / D, c = e - r * D;
                dob[(size_t)r * g.ldt + c] = dys[(size_t)r * g.ldx + c];
            }
        }
        __syncthreads();

        // (4) attention backward, pass 1: one lane per query row -> delta, dQ (written over O)
        const int ntasks = nsq * H * L;
        for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
            const int i = task % L;
            const int h = (task / L) % H;
            const int sq = task / (L * H);
            const int row_i = sq * L + i;
            const float* qp = qkv + (size_t)row_i * g.ldq + h * dh;
            const float* dop = dob + (size_t)row_i * g.ldt + h * dh;
            float* op = ob + (size_t)row_i * g.ldt + h * dh;
            float q[DH_MAX], go[DH_MAX], dq[DH_MAX];
            float delta = 0.f;
#pragma unroll
            for (int c = 0; c < DH_MAX; ++c) {
                q[c] = c < dh ? qp[c] : 0.f;
                go[c] = c < dh ? dop[c] : 0.f;
                dq[c] = 0.f;
                if (c < dh) delta = fmaf(go[c], op[c], delta);
            }
            dlt[row_i * H + h] = delta;
            const float lse = lses[row_i * H + h];
            for (int j = 0; j < L; ++j) {
                const float* kp = qkv + (size_t)(sq * L + j) * g.ldq + I + h * dh;
                const float* vp = kp + I;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < DH_MAX; ++c)
                    if (c < dh) {
                        s = fmaf(q[c], kp[c], s);
                        dp = fmaf(go[c], vp[c], dp);
                    }
                const float p = expf(s * a.scale - lse);
                const float ds = p * (dp - delta);
#pragma unroll
                for (int c = 0; c < DH_MAX; ++c)
                    if (c < dh) dq[c] = fmaf(ds, kp[c], dq[c]);
            }
#pragma unroll
            for (int c = 0; c < DH_MAX; ++c)
                if (c < dh) op[c] = dq[c] * a.scale;
        }
        __syncthreads();
        // pass 2: one lane per key row -> dK, dV (written over K, V)
        for (int task = threadIdx.x; task < ntasks; task += ATT_THREADS) {
            const int j = task % L;
            const int h = (task / L) % H;
            const int sq = task / (L * H);
            float* kp = qkv + (size_t)(sq * L + j) * g.ldq + I + h * dh;
            float* vp = kp + I;
            float kk[DH_MAX], vv[DH_MAX], dk[DH_MAX], dv[DH_MAX];
#pragma unroll
            for (int c = 0; c < DH_MAX; ++c) {
                kk[c] = c < dh ? kp[c] : 0.f;
                vv[c] = c < dh ? vp[c] : 0.f;
                dk[c] = dv[c] = 0.f;
            }
            for (int i = 0; i < L; ++i) {
                const int row_i = sq * L + i;
                const float* qp = qkv + (size_t)row_i * g.ldq + h * dh;
                const float* dop = dob + (size_t)row_i * g.ldt + h * dh;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int c = 0; c < DH_MAX; ++c)
                    if (c < dh) {
                        s = fmaf(qp[c], kk[c], s);
                        dp = fmaf(dop[c], vv[c], dp);
                    }
                const float p = expf(s * a.scale - lses[row_i * H + h]);
                const float ds = p * (dp - dlt[row_i * H + h]);
#pragma unroll
                for (int c = 0; c < DH_MAX; ++c)
                    if (c < dh) {
                        dk[c] = fmaf(ds, qp[c], dk[c]);
                        dv[c] = fmaf(p, dop[c], dv[c]);
                    }
            }
#pragma unroll
            for (int c = 0; c < DH_MAX; ++c)
                if (c < dh) {
                    kp[c] = dk[c] * a.scale;
                    vp[c] = dv[c];
                }
        }
        __syncthreads();
        // dQ (in ob) -> Q columns of qkv: qkv now holds d[Q|K|V]
        for (int e = threadIdx.x; e < rows * I; e += ATT_THREADS) {
            const int r = e / I, c = e - r * I;
            qkv[(size_t)r * g.ldq + c] = ob[(size_t)r * g.ldt + c];
        }
        __syncthreads();

        // (5) d(LN out) = dQKV W_qkv -> dob[rows][0:D] ; (6) dW_qkv += dQKV^T LN(x)
        {
            const int ntn = g.D16 / 16, mblocks = (mt_valid + 1) / 2, ntasks2 = mblocks * ntn;
            const RatLdsRows A{qkv, g.ldq};
            const RatGlobalWkn Bw{a.w_qkv, g.Q3, D, D};
            for (int task = rat_wave(); task < ntasks2; task += ATT_WAVES) {
                const int mt0 = (task / ntn) * 2, nt = task % ntn;
                const int mtv = mt_valid - mt0 < 2 ? mt_valid - mt0 : 2;
                f32x4 acc[2][1];
                acc[0][0] = acc[1][0] = rat_zero4();
                rat_wave_gemm<2, 1>(acc, A, Bw, mt0, nt, mtv, 1, g.Q16 / 16);
                const int col = rat_acc_col(nt);
                if (col < D)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (i < mtv)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dob[(size_t)rat_acc_row(mt0 + i, r) * g.ldt + col] = acc[i][0][r];
            }
            const RatLdsCols At{qkv, g.ldq};
            const RatLdsCols Bt{xs, g.ldx};
#pragma unroll
            for (int s = 0; s < QSLOTS; ++s) {
                const int id = rat_wave() + ATT_WAVES * s;
                if (id < q_tiles) accq[s] = rat_wave_gemm1(accq[s], At, Bt, id / q_tn, id % q_tn, mt_valid);
            }
        }
        __syncthreads();

        // (7) LayerNorm backward + residual: dx = dy + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxn * gamma
        {
            const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
            const bool valid = r < rows;
            const int64_t tok = rowtok[r];
            const float mean = mu[r], rstd = rs[r];
            float xh[LN_COLS], gg[LN_COLS];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < LN_COLS; ++k) {
                const int c = sub + 8 * k;
                xh[k] = 0.f;
                gg[k] = 0.f;
                if (c < D && valid) {
                    xh[k] = (a.x[tok * D + c] - mean) * rstd;
                    gg[k] = dob[(size_t)r * g.ldt + c];
                    const float gw = gg[k] * a.ln_g[c];
                    s1 += gw;
                    s2 += gw * xh[k];
                }
            }
            s1 = rat_group_sum<8>(s1) / (float)D;
            s2 = rat_group_sum<8>(s2) / (float)D;
#pragma unroll
            for (int k = 0; k < LN_COLS; ++k) {
                const int c = sub + 8 * k;
                if (c < D && valid) {
                    const float gw = gg[k] * a.ln_g[c];
                    a.y[tok * D + c] = dys[(size_t)r * g.ldx + c] + rstd * (gw - s1 - xh[k] * s2);
                    dgam[k] += gg[k] * xh[k];
                    dbet[k] += gg[k];
                }
            }
        }
        __syncthreads();
    }

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
                const int col = rat_acc_col(id % o_tn);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rat_acc_row(id / o_tn, r);
                    if (row < D && col < I) s_wout[(int64_t)row * I + col] = acco[s][r];
                }
            }
        }
        if (threadIdx.x < D) s_bout[threadIdx.x] = dbo;
    }
    // dgamma / dbeta: 64 row-slots x D partials -> LDS -> column sums
    float* red = xs;                                             // [64][ldx] is free now
    {
        const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
#pragma unroll
        for (int k = 0; k < LN_COLS; ++k) {
            const int c = sub + 8 * k;
            if (c < D) red[(size_t)r * g.ldx + c] = dgam[k];
        }
        __syncthreads();
        if (threadIdx.x < D) {
            float sacc = 0.f;
            for (int rr = 0; rr < ATT_ROWS; ++rr) sacc += red[(size_t)rr * g.ldx + threadIdx.x];
            s_gam[threadIdx.x] = sacc;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LN_COLS; ++k) {
            const int c = sub + 8 * k;
            if (c < D) red[(size_t)r * g.ldx + c] = dbet[k];
        }
        __syncthreads();
        if (threadIdx.x < D) {
            float sacc = 0.f;
            for (int rr = 0; rr < ATT_ROWS; ++rr) sacc += red[(size_t)rr * g.ldx + threadIdx.x];
            s_bet[threadIdx.x] = sacc;
        }
    }
}

int check_dims(const RatSeqMap* map, int d, int heads, int dim_head, bool backward) {
    RAT_REQUIRE(map != nullptr, "null seq map");
    RAT_REQUIRE(d > 0 && heads > 0 && dim_head > 0, "bad dims");
    RAT_REQUIRE(dim_head <= DH_MAX, "dim_head > 16 is not supported by this kernel");
    RAT_REQUIRE(d <= 8 * LN_COLS, "embedding_dim > 128 is not supported by this kernel");
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
    a.nsq_chunk = ATT_ROWS / map->L;
    a.nchunks = (map->nseq + a.nsq_chunk - 1) / a.nsq_chunk;
    a.d = d;
    a.heads = heads;
    a.dh = dim_head;
    a.eps = ln_eps;
    a.scale = 1.0f / sqrtf((float)dim_head);
    const int I = heads * dim_head;
    a.vec_wqkv = (d % 4 == 0) && aligned16(w->w_qkv);
    a.vec_wout = (I % 4 == 0) && aligned16(w->w_out);
}

}  // namespace

extern "C" int rat_attn_fwd(const float* x, float* y, float* o_save, float* lse_save, const RatAttnParams* w_host,
                            const RatSeqMap* map_host, int d, int heads, int dim_head, float ln_eps, void* stream) {
    if (check_dims(map_host, d, heads, dim_head, false)) return -1;
    RAT_REQUIRE(x && y && w_host && w_host->ln_g && w_host->ln_b && w_host->w_qkv, "null pointer");
    RAT_REQUIRE(w_host->w_out != nullptr || heads * dim_head == d, "missing w_out");
    AttnArgs a{};
    fill_common(a, w_host, map_host, d, heads, dim_head, ln_eps);
    a.x = x;
    a.y = y;
    a.o_save = o_save;
    a.lse_save = lse_save;
    a.vec_x = (d % 4 == 0) && aligned16(x);
    const AttnGeom g(d, heads, dim_head);
    const size_t smem = g.fwd_smem();
    const int per_cu = (int)((160 * 1024) / smem) >= 2 ? 2 : 1;
    int64_t blocks = a.nchunks < 256 * per_cu ? a.nchunks : 256 * per_cu;
    RAT_LAUNCH(attn_fwd_kernel, (unsigned)blocks, ATT_THREADS, smem, stream, a);
    return rat_check_launch("rat_attn_fwd");
}

static int attn_bwd_blocks(int64_t nchunks) { return (int)(nchunks < 256 ? nchunks : 256); }

extern "C" size_t rat_attn_bwd_workspace(int d, int heads, int dim_head) {
    const AttnGeom g(d, heads, dim_head);
    return (size_t)256 * (size_t)g.slab_floats() * sizeof(float);
}

extern "C" int rat_attn_bwd(const float* x, const float* dy, const float* o_save, const float* lse_save, float* dx,
                            const RatAttnParams* w_host, const RatAttnParams* grads_host, float* workspace,
                            size_t workspace_bytes, const RatSeqMap* map_host, int d, int heads, int dim_head,
                            float ln_eps, void* stream) {
    if (check_dims(map_host, d, heads, dim_head, true)) return -1;
    RAT_REQUIRE(x && dy && o_save && lse_save && dx && w_host && grads_host && workspace, "null pointer");
    RAT_REQUIRE(w_host->w_out != nullptr || heads * dim_head == d, "missing w_out");
    RAT_REQUIRE(workspace_bytes >= rat_attn_bwd_workspace(d, heads, dim_head), "workspace too small");
    AttnArgs a{};
    fill_common(a, w_host, map_host, d, heads, dim_head, ln_eps);
    a.x = x;
    a.dy = dy;
    a.y = dx;
    a.o_save = const_cast<float*>(o_save);
    a.lse_save = const_cast<float*>(lse_save);
    a.vec_x = (d % 4 == 0) && aligned16(x) && aligned16(dy) && aligned16(o_save);
    const AttnGeom g(d, heads, dim_head);
    a.slabs = workspace;
    a.slab_stride = g.slab_floats();
    const int blocks = attn_bwd_blocks(a.nchunks);
    RAT_LAUNCH(attn_bwd_kernel, blocks, ATT_THREADS, g.bwd_smem(heads), stream, a);
    if (rat_check_launch("rat_attn_bwd")) return -1;
    const int D = d, I = heads * dim_head;
    float* outs[5] = {grads_host->w_qkv, grads_host->w_out, grads_host->b_out, grads_host->ln_g, grads_host->ln_b};
    const int64_t sizes[5] = {(int64_t)3 * I * D, w_host->w_out ? (int64_t)D * I : 0, w_host->w_out ? D : 0, D, D};
    const int64_t offs[5] = {0, (int64_t)3 * I * D, (int64_t)3 * I * D + (int64_t)D * I,
                             (int64_t)3 * I * D + (int64_t)D * I + D, (int64_t)3 * I * D + (int64_t)D * I + 2 * D};
    return rat_launch_reduce_slabs(workspace, blocks, a.slab_stride, outs, offs, sizes, 5, stream);
}
