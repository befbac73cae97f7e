// head.hip — K3: BatchNorm1d+ReLU, bias gradients, and the logit / sigmoid / BCE stage of the prediction head.
//   MLP_Layer (fuxictr/pytorch/layers/deep.py:126-141), LR_Layer (shallow.py:36-45), RAT_m2.forward lines 138-150,
//   BCE of get_loss_fn (torch_utils.py:51-63).
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int HD_THREADS = 256;
constexpr int HD_COLS = 32;                 // columns per block
constexpr int HD_RG = HD_THREADS / HD_COLS; // row groups per block

// sum over the HD_RG row groups of one column; result valid for every thread of that column
__device__ __forceinline__ float col_reduce(float v, float* scratch) {
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    __syncthreads();
    scratch[rg * HD_COLS + cg] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < HD_RG; ++r) s += scratch[r * HD_COLS + cg];
    return s;
}

constexpr int LB_SAMPLES = 16;              // samples per block of logit_bwd
constexpr int BN_SPLITS = 32;               // row splits: grid = (N/32 column blocks) x BN_SPLITS

// phase 1 of BatchNorm statistics: per (row split, column) partial sums about a pivot (the column's first row),
// which keeps the one-pass variance free of cancellation: var = (S2 - S1^2/M)/M with S = sums of (z - pivot).
__global__ void __launch_bounds__(HD_THREADS)
bn_stats_kernel(const float* __restrict__ z, float* __restrict__ ws, int M, int N) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    const int col = (blockIdx.x % nblk) * HD_COLS + cg, split = blockIdx.x / nblk;
    const int per = (M + BN_SPLITS - 1) / BN_SPLITS;
    const int r0 = split * per, r1 = r0 + per < M ? r0 + per : M;
    float s1 = 0.f, s2 = 0.f;
    if (col < N) {
        const float pivot = z[col];
        for (int m = r0 + rg; m < r1; m += HD_RG) {
            const float t = z[(size_t)m * N + col] - pivot;
            s1 += t;
            s2 = fmaf(t, t, s2);
        }
    }
    s1 = col_reduce(s1, scratch);
    s2 = col_reduce(s2, scratch);
    if (col < N && rg == 0) {
        ws[(size_t)split * N + col] = s1;
        ws[(size_t)(BN_SPLITS + split) * N + col] = s2;
    }
}

// Hidden-layer activation of the DNN head (MLP_Layer, deep.py:108-141; torch_utils.get_activation, torch_utils.py:83-94).  `act` is one
// of RAT_ACT_* (include/rat_hip.h): 0 ReLU, 1 none, 2 Sigmoid, 3 Tanh, 4 LeakyReLU(0.01), 5 ELU(1.0).  The derivative is written in
// terms of the activation's OUTPUT a (saved by the forward), so the backward needs nothing but a and the incoming gradient.
__device__ __forceinline__ float head_act(float y, int act) {
    switch (act) {
        case 1: return y;
        case 2: return 1.0f / (1.0f + expf(-y));
        case 3: return tanhf(y);
        case 4: return y > 0.f ? y : 0.01f * y;
        case 5: return y > 0.f ? y : expm1f(y);
        default: return y > 0.f ? y : 0.f;
    }
}
__device__ __forceinline__ float head_act_bwd(float a, float da, int act) {
    switch (act) {
        case 1: return da;
        case 2: return da * a * (1.0f - a);
        case 3: return da * (1.0f - a * a);
        case 4: return a > 0.f ? da : 0.01f * da;
        case 5: return a > 0.f ? da : da * (a + 1.0f);
        default: return a > 0.f ? da : 0.f;
    }
}

__global__ void __launch_bounds__(HD_THREADS)
bn_relu_apply_kernel(const float* __restrict__ z, float* __restrict__ a, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float* save_mean, float* save_rstd, const float* ws, int M,
                     int N, int training, int use_bn, float eps, float momentum, int act) {
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    const int col = (blockIdx.x % nblk) * HD_COLS + cg, split = blockIdx.x / nblk;
    if (col >= N) return;
    float mean = 0.f, rstd = 1.f, gam = 1.f, bet = 0.f;
    if (use_bn) {
        if (training) {
            float s1 = 0.f, s2 = 0.f;
            for (int w = 0; w < BN_SPLITS; ++w) {                      // fixed order -> every block gets the same bits
                s1 += ws[(size_t)w * N + col];
                s2 += ws[(size_t)(BN_SPLITS + w) * N + col];
            }
            const float d1 = s1 / (float)M;
            mean = z[col] + d1;
            float var = s2 / (float)M - d1 * d1;
            var = var > 0.f ? var : 0.f;
            rstd = 1.0f / sqrtf(var + eps);
            if (split == 0 && rg == 0) {
                save_mean[col] = mean;
                save_rstd[col] = rstd;
                const float unbiased = M > 1 ? var * (float)M / (float)(M - 1) : var;
                running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
                running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
            }
        } else {
            mean = running_mean[col];
            rstd = 1.0f / sqrtf(running_var[col] + eps);
        }
        gam = gamma[col];
        bet = beta[col];
    }
    const int per = (M + BN_SPLITS - 1) / BN_SPLITS;
    const int r0 = split * per, r1 = r0 + per < M ? r0 + per : M;
    for (int m = r0 + rg; m < r1; m += HD_RG) {
        const float y = (z[(size_t)m * N + col] - mean) * rstd * gam + bet;
        a[(size_t)m * N + col] = head_act(y, act);
    }
}

__global__ void __launch_bounds__(HD_THREADS)
bn_bwd_stats_kernel(const float* __restrict__ z, const float* __restrict__ a, const float* __restrict__ da,
                    const float* save_mean, const float* save_rstd, float* __restrict__ ws, int M, int N, int act) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    const int col = (blockIdx.x % nblk) * HD_COLS + cg, split = blockIdx.x / nblk;
    const int per = (M + BN_SPLITS - 1) / BN_SPLITS;
    const int r0 = split * per, r1 = r0 + per < M ? r0 + per : M;
    float s1 = 0.f, s2 = 0.f;
    if (col < N) {
        const float mean = save_mean[col], rstd = save_rstd[col];
        for (int m = r0 + rg; m < r1; m += HD_RG) {
            const size_t o = (size_t)m * N + col;
            const float g = head_act_bwd(a[o], da[o], act);
            s1 += g;
            s2 = fmaf(g, (z[o] - mean) * rstd, s2);
        }
    }
    s1 = col_reduce(s1, scratch);
    s2 = col_reduce(s2, scratch);
    if (col < N && rg == 0) {
        ws[(size_t)split * N + col] = s1;
        ws[(size_t)(BN_SPLITS + split) * N + col] = s2;
    }
}

__global__ void __launch_bounds__(HD_THREADS)
bn_relu_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ a, const float* __restrict__ da,
                         float* __restrict__ dz, const float* gamma, const float* save_mean, const float* save_rstd,
                         float* dgamma, float* dbeta, const float* ws, int M, int N, int use_bn, int act) {
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    const int col = (blockIdx.x % nblk) * HD_COLS + cg, split = blockIdx.x / nblk;
    if (col >= N) return;
    const int per = (M + BN_SPLITS - 1) / BN_SPLITS;
    const int r0 = split * per, r1 = r0 + per < M ? r0 + per : M;
    if (!use_bn) {
        for (int m = r0 + rg; m < r1; m += HD_RG) {
            const size_t o = (size_t)m * N + col;
            dz[o] = head_act_bwd(a[o], da[o], act);
        }
        return;
    }
    float s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < BN_SPLITS; ++w) {
        s1 += ws[(size_t)w * N + col];
        s2 += ws[(size_t)(BN_SPLITS + w) * N + col];
    }
    if (split == 0 && rg == 0) {
        dgamma[col] = s2;
        dbeta[col] = s1;
    }
    const float mean = save_mean[col], rstd = save_rstd[col], gam = gamma[col];
    const float m1 = s1 / (float)M, m2 = s2 / (float)M;
    for (int m = r0 + rg; m < r1; m += HD_RG) {
        const size_t o = (size_t)m * N + col;
        const float xh = (z[o] - mean) * rstd;
        const float g = head_act_bwd(a[o], da[o], act);
        dz[o] = gam * rstd * (g - m1 - xh * m2);
    }
}

// ---- SyncBN (data parallelism): BatchNorm1d statistics over the GLOBAL batch.  Each rank reduces its own
// rows to (mean_r, M2_r = sum (z - mean_r)^2, count_r) per column, the host all-gathers the 2N+1 floats, and every rank combines
// them in rank order with Chan's pairwise update (stable, and bit-identical on every rank).  Backward: local (sum g, sum g*xhat)
// per column, all-reduced (sum) by the host; dgamma / dbeta stay LOCAL sums (the gradient bucket's all-reduce adds the ranks).
__global__ void __launch_bounds__(HD_THREADS)
bn_local_finalize_kernel(const float* __restrict__ z, const float* __restrict__ ws, float* __restrict__ stats, int M, int N) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col == 0) stats[2 * (size_t)N] = (float)M;
    if (col >= N) return;
    float s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < BN_SPLITS; ++w) {
        s1 += ws[(size_t)w * N + col];
        s2 += ws[(size_t)(BN_SPLITS + w) * N + col];
    }
    const float d1 = s1 / (float)M;
    const float m2 = s2 - s1 * d1;
    stats[col] = z[col] + d1;
    stats[(size_t)N + col] = m2 > 0.f ? m2 : 0.f;
}

__global__ void __launch_bounds__(HD_THREADS)
bn_relu_apply_sync_kernel(const float* __restrict__ z, float* __restrict__ a, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                          const float* __restrict__ all_stats, int world, int M, int N, float eps, float momentum, int act) {
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    const int col = (blockIdx.x % nblk) * HD_COLS + cg, split = blockIdx.x / nblk;
    if (col >= N) return;
    const size_t rec = 2 * (size_t)N + 1;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int r = 0; r < world; ++r) {                                  // fixed rank order -> the same bits on every rank
        const float* st = all_stats + (size_t)r * rec;
        const float nr = st[2 * (size_t)N];
        if (nr <= 0.f) continue;
        const float delta = st[col] - mean, nt = n + nr;
        mean += delta * (nr / nt);
        m2 += st[(size_t)N + col] + delta * delta * (n * nr / nt);
        n = nt;
    }
    float var = m2 / n;
    var = var > 0.f ? var : 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (split == 0 && rg == 0) {
        save_mean[col] = mean;
        save_rstd[col] = rstd;
        const float unbiased = n > 1.f ? var * n / (n - 1.f) : var;
        running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
        running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
    }
    const float gam = gamma[col], bet = beta[col];
    const int per = (M + BN_SPLITS - 1) / BN_SPLITS;
    const int r0 = split * per, r1 = r0 + per < M ? r0 + per : M;
    for (int m = r0 + rg; m < r1; m += HD_RG) {
        const float y = (z[(size_t)m * N + col] - mean) * rstd * gam + bet;
        a[(size_t)m * N + col] = head_act(y, act);
    }
}

__global__ void __launch_bounds__(HD_THREADS)
bn_bwd_sums_final_kernel(const float* __restrict__ ws, float* __restrict__ sums, int N) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= N) return;
    float s1 = 0.f, s2 = 0.f;
    for (int w = 0; w < BN_SPLITS; ++w) {
        s1 += ws[(size_t)w * N + col];
        s2 += ws[(size_t)(BN_SPLITS + w) * N + col];
    }
    sums[col] = s1;
    sums[(size_t)N + col] = s2;
}

__global__ void __launch_bounds__(HD_THREADS)
bn_relu_bwd_apply_sync_kernel(const float* __restrict__ z, const float* __restrict__ a, const float* __restrict__ da,
                              float* __restrict__ dz, const float* gamma, const float* save_mean, const float* save_rstd,
                              const float* local_sums, const float* global_sums, float* dgamma, float* dbeta,
                              const float* __restrict__ all_stats, int world, int M, int N, int act) {
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    const int col = (blockIdx.x % nblk) * HD_COLS + cg, split = blockIdx.x / nblk;
    if (col >= N) return;
    if (split == 0 && rg == 0) {
        dbeta[col] = local_sums[col];
        dgamma[col] = local_sums[(size_t)N + col];
    }
    const float mean = save_mean[col], rstd = save_rstd[col], gam = gamma[col];
    float total_count = 0.f;                                           // global row count, from the forward's gathered records
    for (int r = 0; r < world; ++r) total_count += all_stats[(size_t)r * (2 * (size_t)N + 1) + 2 * (size_t)N];
    const float m1 = global_sums[col] / total_count, m2 = global_sums[(size_t)N + col] / total_count;
    const int per = (M + BN_SPLITS - 1) / BN_SPLITS;
    const int r0 = split * per, r1 = r0 + per < M ? r0 + per : M;
    for (int m = r0 + rg; m < r1; m += HD_RG) {
        const size_t o = (size_t)m * N + col;
        const float xh = (z[o] - mean) * rstd;
        const float g = head_act_bwd(a[o], da[o], act);
        dz[o] = gam * rstd * (g - m1 - xh * m2);
    }
}

// ---- column-strip forms (round 4): ONE launch per layer and direction.  A work-group owns 8 columns (32 bytes of every row) for ALL rows,
// so the batch statistics, their use, and — backward — the Linear bias gradient (the column sums of dz) need no second launch, no
// partial-sum workspace and no atomics; sums are formed in a fixed order (lane tree, then waves 0..7).  512 threads: thread =
// (row slot t / 2, 16-byte half t % 2); rows slot, slot + 256, ...  Up to 4096 rows the strip stays in registers between the passes
// (RPT rows per thread), longer matrices are re-read (they come from L2).  Work-groups that share 128-byte lines are placed on the same
// XCD (block b runs on XCD b % 8: XCD x owns a contiguous range of column groups), so every line is fetched by one L2 only.
constexpr int ST_THREADS = 512;
constexpr int ST_SLOTS = ST_THREADS / 2;

__device__ __forceinline__ float4 st_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st_st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }

// sum over the 256 threads of one half; every thread of that half gets it
__device__ __forceinline__ float4 st_block_sum(float4 v, float* scratch) {
#pragma unroll
    for (int m = 32; m >= 2; m >>= 1) {
        v.x += __shfl_xor(v.x, m, 64);
        v.y += __shfl_xor(v.y, m, 64);
        v.z += __shfl_xor(v.z, m, 64);
        v.w += __shfl_xor(v.w, m, 64);
    }
    const int w = threadIdx.x >> 6, half = threadIdx.x & 1;
    __syncthreads();
    if ((threadIdx.x & 63) < 2) st_st4(scratch + (w * 2 + half) * 4, v);
    __syncthreads();
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < ST_THREADS / 64; ++k) {
        const float4 t = st_ld4(scratch + (k * 2 + half) * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    return s;
}

__device__ __forceinline__ int st_column_group(int ngroups, int per_xcd) {
    const int g = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    return g < ngroups ? g : -1;
}

// RPT > 0: rows held in registers (M <= 256 RPT); RPT == 0: streaming
template <int RPT>
__global__ void __launch_bounds__(ST_THREADS)
bn_act_fwd_strip_kernel(const float* __restrict__ z, float* __restrict__ a, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float* save_mean, float* save_rstd, int M, int N, int ngroups, int per_xcd, int training,
                        int use_bn, float eps, float momentum, int act) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int grp = st_column_group(ngroups, per_xcd);
    if (grp < 0) return;
    const int half = threadIdx.x & 1, slot = threadIdx.x >> 1, col = 8 * grp + 4 * half;
    const bool ok = col < N;
    constexpr int NR = RPT > 0 ? RPT : 1;
    float4 zr[NR];
    if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = slot + ST_SLOTS * i;
            zr[i] = (ok && r < M) ? st_ld4(z + (size_t)r * N + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f), rstd = make_float4(1.f, 1.f, 1.f, 1.f), gam = rstd, bet = mean;
    if (use_bn) {
        if (training) {
            const float4 pivot = ok ? st_ld4(z + col) : mean;
            float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
            auto add = [&](const float4& v) {
                const float tx = v.x - pivot.x, ty = v.y - pivot.y, tz = v.z - pivot.z, tw = v.w - pivot.w;
                s1.x += tx; s1.y += ty; s1.z += tz; s1.w += tw;
                s2.x = fmaf(tx, tx, s2.x); s2.y = fmaf(ty, ty, s2.y); s2.z = fmaf(tz, tz, s2.z); s2.w = fmaf(tw, tw, s2.w);
            };
            if (RPT > 0) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    if (ok && slot + ST_SLOTS * i < M) add(zr[i]);
            } else if (ok) {
                for (int r = slot; r < M; r += ST_SLOTS) add(st_ld4(z + (size_t)r * N + col));
            }
            s1 = st_block_sum(s1, scratch);
            s2 = st_block_sum(s2, scratch);
            const float fm = (float)M;
            float d1[4] = {s1.x / fm, s1.y / fm, s1.z / fm, s1.w / fm};
            const float q2[4] = {s2.x / fm, s2.y / fm, s2.z / fm, s2.w / fm}, pv[4] = {pivot.x, pivot.y, pivot.z, pivot.w};
            float mn[4], rs[4], vr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mn[j] = pv[j] + d1[j];
                float var = q2[j] - d1[j] * d1[j];
                var = var > 0.f ? var : 0.f;
                vr[j] = var;
                rs[j] = 1.0f / sqrtf(var + eps);
            }
            mean = make_float4(mn[0], mn[1], mn[2], mn[3]);
            rstd = make_float4(rs[0], rs[1], rs[2], rs[3]);
            if (slot == 0 && ok) {
                st_st4(save_mean + col, mean);
                st_st4(save_rstd + col, rstd);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float unbiased = M > 1 ? vr[j] * fm / (float)(M - 1) : vr[j];
                    running_mean[col + j] = (1.f - momentum) * running_mean[col + j] + momentum * mn[j];
                    running_var[col + j] = (1.f - momentum) * running_var[col + j] + momentum * unbiased;
                }
            }
        } else if (ok) {
            mean = st_ld4(running_mean + col);
            const float4 rv = st_ld4(running_var + col);
            rstd = make_float4(1.0f / sqrtf(rv.x + eps), 1.0f / sqrtf(rv.y + eps), 1.0f / sqrtf(rv.z + eps), 1.0f / sqrtf(rv.w + eps));
        }
        if (ok) {
            gam = st_ld4(gamma + col);
            bet = st_ld4(beta + col);
        }
    }
    if (!ok) return;
    auto apply = [&](const float4& v, int r) {
        float4 y;
        y.x = head_act((v.x - mean.x) * rstd.x * gam.x + bet.x, act);
        y.y = head_act((v.y - mean.y) * rstd.y * gam.y + bet.y, act);
        y.z = head_act((v.z - mean.z) * rstd.z * gam.z + bet.z, act);
        y.w = head_act((v.w - mean.w) * rstd.w * gam.w + bet.w, act);
        st_st4(a + (size_t)r * N + col, y);
    };
    if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            if (slot + ST_SLOTS * i < M) apply(zr[i], slot + ST_SLOTS * i);
    } else {
        for (int r = slot; r < M; r += ST_SLOTS) apply(st_ld4(z + (size_t)r * N + col), r);
    }
}

// backward of the above + the bias gradient of the Linear in front of it: dbias_lin[c] = sum_rows dz[r][c] (nullptr = not wanted)
template <int RPT>
__global__ void __launch_bounds__(ST_THREADS)
bn_act_bwd_strip_kernel(const float* __restrict__ z, const float* __restrict__ a, const float* __restrict__ da, float* __restrict__ dz,
                        const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                        float* dbias_lin, const float* __restrict__ outer_dl, const float* outer_w, float* outer_dw, int M, int N,
                        int ngroups, int per_xcd, int use_bn, int act) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int grp = st_column_group(ngroups, per_xcd);
    if (grp < 0) return;
    const int half = threadIdx.x & 1, slot = threadIdx.x >> 1, col = 8 * grp + 4 * half;
    const bool ok = col < N;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mean = zero, rstd = make_float4(1.f, 1.f, 1.f, 1.f), gam = rstd;
    if (use_bn && ok) {
        mean = st_ld4(save_mean + col);
        rstd = st_ld4(save_rstd + col);
        gam = st_ld4(gamma + col);
    }
    // outer_dl != nullptr: this is the LAST hidden layer and the Linear behind it has ONE output (the DNN's logit, deep.py:135-137):
    // its input gradient is the outer product da[r][c] = dl[r] * w[c] — formed here instead of being read — and its weight gradient
    // dw[c] = sum_r dl[r] * a[r][c] is one more column sum of this strip.
    float4 ow = zero, sw = zero;
    if (outer_dl != nullptr && ok) ow = st_ld4(outer_w + col);
    auto grad_of = [&](int r, bool count) {
        const size_t o = (size_t)r * N + col;
        const float4 av = st_ld4(a + o);
        float4 dv;
        if (outer_dl != nullptr) {
            const float dl = outer_dl[r];
            dv = make_float4(dl * ow.x, dl * ow.y, dl * ow.z, dl * ow.w);
            if (count) {
                sw.x = fmaf(dl, av.x, sw.x); sw.y = fmaf(dl, av.y, sw.y); sw.z = fmaf(dl, av.z, sw.z); sw.w = fmaf(dl, av.w, sw.w);
            }
        } else {
            dv = st_ld4(da + o);
        }
        return make_float4(head_act_bwd(av.x, dv.x, act), head_act_bwd(av.y, dv.y, act), head_act_bwd(av.z, dv.z, act),
                           head_act_bwd(av.w, dv.w, act));
    };
    auto xhat_of = [&](size_t o) {
        const float4 zv = st_ld4(z + o);
        return make_float4((zv.x - mean.x) * rstd.x, (zv.y - mean.y) * rstd.y, (zv.z - mean.z) * rstd.z, (zv.w - mean.w) * rstd.w);
    };
    constexpr int NR = RPT > 0 ? RPT : 1;
    float4 g[NR], xh[NR];
    if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = slot + ST_SLOTS * i;
            const bool live = ok && r < M;
            g[i] = live ? grad_of(r, true) : zero;
            xh[i] = (live && use_bn) ? xhat_of((size_t)r * N + col) : zero;
        }
    }
    float4 m1 = zero, m2 = zero;
    if (use_bn) {
        float4 s1 = zero, s2 = zero;
        auto add = [&](const float4& gv, const float4& xv) {
            s1.x += gv.x; s1.y += gv.y; s1.z += gv.z; s1.w += gv.w;
            s2.x = fmaf(gv.x, xv.x, s2.x); s2.y = fmaf(gv.y, xv.y, s2.y); s2.z = fmaf(gv.z, xv.z, s2.z); s2.w = fmaf(gv.w, xv.w, s2.w);
        };
        if (RPT > 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) add(g[i], xh[i]);                    // dead rows hold zeros
        } else if (ok) {
            for (int r = slot; r < M; r += ST_SLOTS) add(grad_of(r, true), xhat_of((size_t)r * N + col));
        }
        s1 = st_block_sum(s1, scratch);
        s2 = st_block_sum(s2, scratch);
        if (slot == 0 && ok) {
            st_st4(dgamma + col, s2);
            st_st4(dbeta + col, s1);
        }
        const float fm = (float)M;
        m1 = make_float4(s1.x / fm, s1.y / fm, s1.z / fm, s1.w / fm);
        m2 = make_float4(s2.x / fm, s2.y / fm, s2.z / fm, s2.w / fm);
    }
    float4 sd = zero;
    auto emit = [&](const float4& gv, const float4& xv, int r) {
        float4 d = gv;
        if (use_bn) {
            d.x = gam.x * rstd.x * (gv.x - m1.x - xv.x * m2.x);
            d.y = gam.y * rstd.y * (gv.y - m1.y - xv.y * m2.y);
            d.z = gam.z * rstd.z * (gv.z - m1.z - xv.z * m2.z);
            d.w = gam.w * rstd.w * (gv.w - m1.w - xv.w * m2.w);
        }
        sd.x += d.x; sd.y += d.y; sd.z += d.z; sd.w += d.w;
        st_st4(dz + (size_t)r * N + col, d);
    };
    if (ok) {
        if (RPT > 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (slot + ST_SLOTS * i < M) emit(g[i], xh[i], slot + ST_SLOTS * i);
        } else {
            for (int r = slot; r < M; r += ST_SLOTS)
                emit(grad_of(r, !use_bn), use_bn ? xhat_of((size_t)r * N + col) : zero, r);
        }
    }
    if (dbias_lin != nullptr) {
        sd = st_block_sum(sd, scratch);
        if (slot == 0 && ok) st_st4(dbias_lin + col, sd);
    }
    if (outer_dl != nullptr) {
        sw = st_block_sum(sw, scratch);
        if (slot == 0 && ok) st_st4(outer_dw + col, sw);
    }
}

// column sums, two stages when the matrix is tall: (column block x row split) partials into `ws`, then a fixed-order
// combine — deterministic, and 32x more blocks in flight than one block per 32 columns.
__global__ void __launch_bounds__(HD_THREADS)
colsum_partial_kernel(const float* __restrict__ a, int lda, float* __restrict__ ws, int M, int N, int splits) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int cg = threadIdx.x % HD_COLS, rg = threadIdx.x / HD_COLS;
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    const int col = (blockIdx.x % nblk) * HD_COLS + cg, split = blockIdx.x / nblk;
    const int per = (M + splits - 1) / splits;
    const int r0 = split * per, r1 = r0 + per < M ? r0 + per : M;
    float s = 0.f;
    if (col < N) for (int m = r0 + rg; m < r1; m += HD_RG) s += a[(size_t)m * lda + col];
    s = col_reduce(s, scratch);
    if (col < N && rg == 0) ws[(size_t)split * N + col] = s;
}

__global__ void __launch_bounds__(HD_THREADS)
colsum_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int N, int splits) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= N) return;
    float s = 0.f;
    for (int w = 0; w < splits; ++w) s += ws[(size_t)w * N + col];
    out[col] = s;
}

__device__ __forceinline__ float block_sum(float v, float* scratch) {   // 256 threads
    v = rat_group_sum<64>(v);
    __syncthreads();
    if (rat_lane() == 0) scratch[rat_wave()] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

constexpr int LF_LANES = 16;                 // lanes per sample of logit_fwd
constexpr int LF_SAMPLES = HD_THREADS / LF_LANES;

__device__ __forceinline__ float logit_dot(const float* __restrict__ x, const float* __restrict__ w, int n, int sub, bool vec) {
    float acc = 0.f;
    if (vec) {
        for (int k = 4 * sub; k < n; k += 4 * LF_LANES) {
            const float4 xv = *reinterpret_cast<const float4*>(x + k), wv = *reinterpret_cast<const float4*>(w + k);
            acc = fmaf(xv.x, wv.x, acc);
            acc = fmaf(xv.y, wv.y, acc);
            acc = fmaf(xv.z, wv.z, acc);
            acc = fmaf(xv.w, wv.w, acc);
        }
    } else {
        for (int k = sub; k < n; k += LF_LANES) acc = fmaf(x[k], w[k], acc);
    }
    return acc;
}

// 16 lanes per sample: the fc dot over the cls row, the DNN's ONE-output Linear (dnn_in != nullptr: deep.py:135-137, formerly an
// N = 1 GEMM launch of its own) and the LR lookups are dealt over the lanes and met by a 16-lane sum.
__global__ void __launch_bounds__(HD_THREADS)
logit_fwd_kernel(const float* __restrict__ cls, int64_t cls_stride, const float* fc_w, const float* fc_b,
                 const float* dnn_out, const float* __restrict__ dnn_in, int64_t dnn_ld, const float* dnn_w, const float* dnn_b,
                 int dnn_k, const RatField* lr_fields, int nfields, const int32_t* idx, int64_t idx_stride,
                 const float* y_true, float* y_pred, float* loss_sum, int B, int d, int head, int vec_cls, int vec_dnn) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int sub = threadIdx.x % LF_LANES, b = blockIdx.x * LF_SAMPLES + threadIdx.x / LF_LANES;
    float part = 0.f;
    if (b < B) {
        part = logit_dot(cls + (int64_t)b * cls_stride, fc_w, d, sub, vec_cls != 0);
        if (dnn_in != nullptr) part += logit_dot(dnn_in + (int64_t)b * dnn_ld, dnn_w, dnn_k, sub, vec_dnn != 0);
        if (lr_fields != nullptr) {
            float lr = 0.f;
            for (int f = sub; f < nfields; f += LF_LANES) {
                const RatField fd = lr_fields[f];
                const int32_t* ids = idx + (int64_t)b * idx_stride + fd.col;
                for (int j = 0; j < fd.ncols; ++j) {
                    int id = ids[j];
                    id = id < 0 ? 0 : (id >= fd.vocab ? fd.vocab - 1 : id);
                    lr += fd.table[id];
                }
            }
            part += lr;
        }
    }
    part = rat_group_sum<LF_LANES>(part);
    float loss = 0.f;
    if (b < B && sub == 0) {
        float zl = part + fc_b[0];
        if (dnn_in != nullptr) zl += dnn_b[0];
        if (dnn_out != nullptr) zl += dnn_out[b];
        if (head == 1) {                                         // task = "regression": no output activation, mean squared error
            y_pred[b] = zl;
            if (y_true != nullptr) {
                const float e = zl - y_true[b];
                loss = e * e / (float)B;
            }
        } else {
            const float p = 1.0f / (1.0f + expf(-zl));
            y_pred[b] = p;
            if (y_true != nullptr) {
                const float t = y_true[b];
                const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.0f - p), -100.f);
                loss = -(t * lp + (1.0f - t) * l1p) / (float)B;
            }
        }
    }
    if (loss_sum != nullptr) {
        const float s = block_sum(loss, scratch);
        if (threadIdx.x == 0) atomicAdd(loss_sum, s);
    }
}

__global__ void __launch_bounds__(HD_THREADS)
logit_bwd_kernel(const float* __restrict__ y_pred, const float* __restrict__ y_true, const float* __restrict__ cls,
                 int64_t cls_stride, const float* fc_w, float* dlogit, float* dcls, int64_t dcls_stride, float* dfc_w,
                 float* dfc_b, float* ddnn_b, const RatField* lr_grad_fields, int nfields, const int32_t* idx, int64_t idx_stride,
                 float gscale, const float* gscale_dev, int B, int d, int head) {
    // One block = LB_SAMPLES samples.  Phase 1: thread = (sample, field slot): dlogit and the LR-table atomics of the fields f = slot,
    // slot + 16, ...  Phase 2: thread = (column k, sample group): dcls rows and the dfc_w partial sums, column-parallel, no LDS atomics.
    // ddnn_b (nullable): the bias gradient of the DNN's one-output Linear — the same sum over samples as dfc_b.
    RAT_DYN_SMEM(smem);
    float* dls = reinterpret_cast<float*>(smem);              // [LB_SAMPLES] dlogit of this block's samples
    float* part = dls + LB_SAMPLES;                           // [groups][d] partial dfc_w
    const int b0 = blockIdx.x * LB_SAMPLES;
    const int nb = B - b0 < LB_SAMPLES ? B - b0 : LB_SAMPLES;
    if (gscale_dev != nullptr) gscale *= *gscale_dev;         // the incoming loss gradient stays on the device (no host read-back)
    {
        const int sl = threadIdx.x % LB_SAMPLES, slot = threadIdx.x / LB_SAMPLES;
        float dl = 0.f;
        if (sl < nb) {
            const int b = b0 + sl;
            dl = gscale * (head == 1 ? 2.0f : 1.0f) * (y_pred[b] - y_true[b]) / (float)B;    // d BCE(sigmoid z) / dz = p - t; d MSE / dz = 2 (z - t)
            if (slot == 0) dlogit[b] = dl;
            if (lr_grad_fields != nullptr)
                for (int f = slot; f < nfields; f += HD_THREADS / LB_SAMPLES) {
                    const RatField fd = lr_grad_fields[f];
                    const int32_t* ids = idx + (int64_t)b * idx_stride + fd.col;
                    for (int j = 0; j < fd.ncols; ++j) {
                        int id = ids[j];
                        id = id < 0 ? 0 : (id >= fd.vocab ? fd.vocab - 1 : id);
                        if (id != fd.padding_idx) atomicAdd(fd.table + id, dl);
                    }
                }
        }
        if (slot == 0) dls[sl] = dl;
    }
    __syncthreads();
    const int groups = blockDim.x / d;                        // d <= blockDim.x (checked on the host)
    const int k = threadIdx.x % d, grp = threadIdx.x / d;
    if (grp < groups) {
        const float w = fc_w[k];
        float acc = 0.f;
        for (int s = grp; s < nb; s += groups) {
            const float dl = dls[s];
            dcls[(int64_t)(b0 + s) * dcls_stride + k] = dl * w;
            acc = fmaf(dl, cls[(int64_t)(b0 + s) * cls_stride + k], acc);
        }
        part[grp * d + k] = acc;
    }
    __syncthreads();
    if ((int)threadIdx.x < d) {
        float acc = 0.f;
        for (int gI = 0; gI < groups; ++gI) acc += part[gI * d + threadIdx.x];
        atomicAdd(&dfc_w[threadIdx.x], acc);
    }
    if (threadIdx.x == 0) {
        float dbias = 0.f;
        for (int sI = 0; sI < LB_SAMPLES; ++sI) dbias += dls[sI];
        atomicAdd(dfc_b, dbias);
        if (ddnn_b != nullptr) atomicAdd(ddnn_b, dbias);
    }
}

}  // namespace

extern "C" size_t rat_bn_workspace(int N) { return (size_t)2 * BN_SPLITS * (size_t)N * sizeof(float); }

extern "C" int rat_bn_relu_fwd(const float* z, float* a, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float* save_mean, float* save_rstd, float* workspace, int M, int N,
                               int training, int use_bn, float eps, float momentum, int act, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && z && a && act >= 0 && act <= 5, "bad args");
    const int blocks = ((N + HD_COLS - 1) / HD_COLS) * BN_SPLITS;
    if (use_bn) {
        RAT_REQUIRE(gamma && beta && running_mean && running_var, "null BN pointer");
        if (training) {
            RAT_REQUIRE(save_mean && save_rstd && workspace, "training BN needs save_mean/save_rstd/workspace");
            RAT_LAUNCH(bn_stats_kernel, blocks, HD_THREADS, HD_THREADS * sizeof(float), stream, z, workspace, M, N);
        }
    }
    RAT_LAUNCH(bn_relu_apply_kernel, blocks, HD_THREADS, 0, stream, z, a, gamma, beta, running_mean, running_var, save_mean,
               save_rstd, workspace, M, N, training, use_bn, eps, momentum, act);
    return rat_check_launch("rat_bn_relu_fwd");
}

namespace {
// row splits of the column sum: 32 for the prediction head's batches; long token matrices (the composed attention path: M = all
// tokens of the batch) get one split per ~2048 rows so that the partial-sum launch fills the chip
int colsum_splits(int M) {
    if (M < 64) return 1;
    if (M < 1024) return 4;
    const int s = M / 2048;
    return s < BN_SPLITS ? BN_SPLITS : (s > 2048 ? 2048 : s);
}
}  // namespace

extern "C" size_t rat_colsum_workspace(int M, int N) {
    const size_t need = (size_t)colsum_splits(M) * (size_t)N * sizeof(float);
    const size_t bn = rat_bn_workspace(N);
    return need > bn ? need : bn;                                   // never below the BatchNorm workspace older callers pass
}

extern "C" int rat_colsum(const float* a, int lda, float* out, float* workspace, int M, int N, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && a && out && workspace, "bad args");     // workspace: rat_colsum_workspace(M, N) bytes
    const int splits = colsum_splits(M);
    const int nblk = (N + HD_COLS - 1) / HD_COLS;
    RAT_LAUNCH(colsum_partial_kernel, nblk * splits, HD_THREADS, HD_THREADS * sizeof(float), stream, a, lda, workspace, M, N, splits);
    RAT_LAUNCH(colsum_final_kernel, (N + HD_THREADS - 1) / HD_THREADS, HD_THREADS, 0, stream, workspace, out, N, splits);
    return rat_check_launch("rat_colsum");
}

static int logit_fwd_launch(const float* cls, int64_t cls_stride, const float* fc_w, const float* fc_b, const float* dnn_out,
                            const float* dnn_in, int64_t dnn_ld, const float* dnn_w, const float* dnn_b, int dnn_k,
                            const RatField* lr_fields_dev, int nfields, const int32_t* idx, int64_t idx_stride, const float* y_true,
                            float* y_pred, float* loss_sum, int B, int d, int head, void* stream) {
    RAT_REQUIRE(B > 0 && d > 0 && cls && fc_w && fc_b && y_pred && (head == 0 || head == 1), "bad args");
    RAT_REQUIRE(lr_fields_dev == nullptr || idx != nullptr, "LR term needs idx");
    RAT_REQUIRE(dnn_in == nullptr || (dnn_w && dnn_b && dnn_k > 0), "the DNN's output layer needs its weight, bias and width");
    const auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int vec_cls = d % 4 == 0 && cls_stride % 4 == 0 && al(cls) && al(fc_w);
    const int vec_dnn = dnn_in != nullptr && dnn_k % 4 == 0 && dnn_ld % 4 == 0 && al(dnn_in) && al(dnn_w);
    RAT_LAUNCH(logit_fwd_kernel, (B + LF_SAMPLES - 1) / LF_SAMPLES, HD_THREADS, 16 * sizeof(float), stream, cls, cls_stride,
               fc_w, fc_b, dnn_out, dnn_in, dnn_ld, dnn_w, dnn_b, dnn_k, lr_fields_dev, nfields, idx, idx_stride, y_true, y_pred,
               loss_sum, B, d, head, vec_cls, vec_dnn);
    return rat_check_launch("rat_logit_fwd");
}

extern "C" int rat_logit_fwd(const float* cls, int64_t cls_stride, const float* fc_w, const float* fc_b, const float* dnn_out,
                             const RatField* lr_fields_dev, int nfields, const int32_t* idx, int64_t idx_stride,
                             const float* y_true, float* y_pred, float* loss_sum, int B, int d, int head, void* stream) {
    return logit_fwd_launch(cls, cls_stride, fc_w, fc_b, dnn_out, nullptr, 0, nullptr, nullptr, 0, lr_fields_dev, nfields, idx,
                            idx_stride, y_true, y_pred, loss_sum, B, d, head, stream);
}

extern "C" int rat_logit_fwd_dnn(const float* cls, int64_t cls_stride, const float* fc_w, const float* fc_b, const float* dnn_in,
                                 int64_t dnn_ld, const float* dnn_w, const float* dnn_b, int dnn_k, const RatField* lr_fields_dev,
                                 int nfields, const int32_t* idx, int64_t idx_stride, const float* y_true, float* y_pred,
                                 float* loss_sum, int B, int d, int head, void* stream) {
    RAT_REQUIRE(dnn_in != nullptr, "null dnn_in");
    return logit_fwd_launch(cls, cls_stride, fc_w, fc_b, nullptr, dnn_in, dnn_ld, dnn_w, dnn_b, dnn_k, lr_fields_dev, nfields, idx,
                            idx_stride, y_true, y_pred, loss_sum, B, d, head, stream);
}

static int logit_bwd_launch(const float* y_pred, const float* y_true, const float* cls, int64_t cls_stride, const float* fc_w,
                            float* dlogit, float* dcls, int64_t dcls_stride, float* dfc_w, float* dfc_b, float* ddnn_b,
                            const RatField* lr_grad_fields_dev, int nfields, const int32_t* idx, int64_t idx_stride, float gscale,
                            const float* gscale_dev, int B, int d, int head, void* stream) {
    RAT_REQUIRE(B > 0 && d > 0 && y_pred && y_true && cls && fc_w && dlogit && dcls && dfc_w && dfc_b && (head == 0 || head == 1), "bad args");
    RAT_REQUIRE(lr_grad_fields_dev == nullptr || idx != nullptr, "LR term needs idx");
    RAT_REQUIRE(d <= HD_THREADS, "embedding_dim above the block size");
    RAT_LAUNCH(logit_bwd_kernel, (B + LB_SAMPLES - 1) / LB_SAMPLES, HD_THREADS, (size_t)(LB_SAMPLES + HD_THREADS) * sizeof(float), stream,
               y_pred, y_true, cls, cls_stride, fc_w, dlogit, dcls, dcls_stride, dfc_w, dfc_b, ddnn_b, lr_grad_fields_dev, nfields,
               idx, idx_stride, gscale, gscale_dev, B, d, head);
    return rat_check_launch("rat_logit_bwd");
}

extern "C" int rat_logit_bwd(const float* y_pred, const float* y_true, const float* cls, int64_t cls_stride,
                             const float* fc_w, float* dlogit, float* dcls, int64_t dcls_stride, float* dfc_w, float* dfc_b,
                             const RatField* lr_grad_fields_dev, int nfields, const int32_t* idx, int64_t idx_stride,
                             float gscale, const float* gscale_dev, int B, int d, int head, void* stream) {
    return logit_bwd_launch(y_pred, y_true, cls, cls_stride, fc_w, dlogit, dcls, dcls_stride, dfc_w, dfc_b, nullptr, lr_grad_fields_dev,
                            nfields, idx, idx_stride, gscale, gscale_dev, B, d, head, stream);
}

extern "C" int rat_logit_bwd_dnn(const float* y_pred, const float* y_true, const float* cls, int64_t cls_stride,
                                 const float* fc_w, float* dlogit, float* dcls, int64_t dcls_stride, float* dfc_w, float* dfc_b,
                                 float* ddnn_b, const RatField* lr_grad_fields_dev, int nfields, const int32_t* idx,
                                 int64_t idx_stride, float gscale, const float* gscale_dev, int B, int d, int head, void* stream) {
    RAT_REQUIRE(ddnn_b != nullptr, "null ddnn_b");
    return logit_bwd_launch(y_pred, y_true, cls, cls_stride, fc_w, dlogit, dcls, dcls_stride, dfc_w, dfc_b, ddnn_b, lr_grad_fields_dev,
                            nfields, idx, idx_stride, gscale, gscale_dev, B, d, head, stream);
}

extern "C" int rat_bn_relu_bwd(const float* z, const float* a, const float* da, float* dz, const float* gamma,
                               const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                               float* workspace, int M, int N, int use_bn, int act, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && z && a && da && dz && act >= 0 && act <= 5, "bad args");
    const int blocks = ((N + HD_COLS - 1) / HD_COLS) * BN_SPLITS;
    if (use_bn) {
        RAT_REQUIRE(gamma && save_mean && save_rstd && dgamma && dbeta && workspace, "null BN pointer");
        RAT_LAUNCH(bn_bwd_stats_kernel, blocks, HD_THREADS, HD_THREADS * sizeof(float), stream, z, a, da, save_mean, save_rstd,
                   workspace, M, N, act);
    }
    RAT_LAUNCH(bn_relu_bwd_apply_kernel, blocks, HD_THREADS, 0, stream, z, a, da, dz, gamma, save_mean, save_rstd, dgamma,
               dbeta, workspace, M, N, use_bn, act);
    return rat_check_launch("rat_bn_relu_bwd");
}

// ---- SyncBN entry points (see the kernels above; the collectives between them are the caller's: torch.distributed / RCCL)
extern "C" int rat_bn_local_stats(const float* z, float* stats, float* workspace, int M, int N, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && z && stats && workspace, "bad args");
    const int blocks = ((N + HD_COLS - 1) / HD_COLS) * BN_SPLITS;
    RAT_LAUNCH(bn_stats_kernel, blocks, HD_THREADS, HD_THREADS * sizeof(float), stream, z, workspace, M, N);
    RAT_LAUNCH(bn_local_finalize_kernel, (N + HD_THREADS - 1) / HD_THREADS, HD_THREADS, 0, stream, z, workspace, stats, M, N);
    return rat_check_launch("rat_bn_local_stats");
}

extern "C" int rat_bn_relu_fwd_sync(const float* z, float* a, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float* save_mean, float* save_rstd, const float* all_stats, int world,
                                    int M, int N, float eps, float momentum, int act, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && world >= 1 && z && a && gamma && beta && running_mean && running_var && save_mean && save_rstd &&
                all_stats && act >= 0 && act <= 5, "bad args");
    const int blocks = ((N + HD_COLS - 1) / HD_COLS) * BN_SPLITS;
    RAT_LAUNCH(bn_relu_apply_sync_kernel, blocks, HD_THREADS, 0, stream, z, a, gamma, beta, running_mean, running_var, save_mean,
               save_rstd, all_stats, world, M, N, eps, momentum, act);
    return rat_check_launch("rat_bn_relu_fwd_sync");
}

extern "C" int rat_bn_bwd_local_sums(const float* z, const float* a, const float* da, const float* save_mean,
                                     const float* save_rstd, float* sums, float* workspace, int M, int N, int act, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && z && a && da && save_mean && save_rstd && sums && workspace && act >= 0 && act <= 5, "bad args");
    const int blocks = ((N + HD_COLS - 1) / HD_COLS) * BN_SPLITS;
    RAT_LAUNCH(bn_bwd_stats_kernel, blocks, HD_THREADS, HD_THREADS * sizeof(float), stream, z, a, da, save_mean, save_rstd,
               workspace, M, N, act);
    RAT_LAUNCH(bn_bwd_sums_final_kernel, (N + HD_THREADS - 1) / HD_THREADS, HD_THREADS, 0, stream, workspace, sums, N);
    return rat_check_launch("rat_bn_bwd_local_sums");
}

extern "C" int rat_bn_relu_bwd_sync(const float* z, const float* a, const float* da, float* dz, const float* gamma,
                                    const float* save_mean, const float* save_rstd, const float* local_sums,
                                    const float* global_sums, float* dgamma, float* dbeta, const float* all_stats, int world,
                                    int M, int N, int act, void* stream) {
    RAT_REQUIRE(M > 0 && N > 0 && world >= 1 && z && a && da && dz && gamma && save_mean && save_rstd && local_sums &&
                global_sums && dgamma && dbeta && all_stats && act >= 0 && act <= 5, "bad args");
    const int blocks = ((N + HD_COLS - 1) / HD_COLS) * BN_SPLITS;
    RAT_LAUNCH(bn_relu_bwd_apply_sync_kernel, blocks, HD_THREADS, 0, stream, z, a, da, dz, gamma, save_mean, save_rstd, local_sums,
               global_sums, dgamma, dbeta, all_stats, world, M, N, act);
    return rat_check_launch("rat_bn_relu_bwd_sync");
}

// ---- column-strip entry points (ABI v7): rat_bn_relu_fwd / rat_bn_relu_bwd (+ rat_colsum of dz) in ONE launch each, no workspace.
// They need N % 4 == 0 and 16-byte aligned matrices / vectors (every nn.Linear width of the shipped configs); rat_bn_strip_ok says so.
namespace {
bool st_aligned(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

extern "C" int rat_bn_strip_ok(int M, int N) { return M > 0 && N > 0 && (N % 4) == 0; }

extern "C" int rat_bn_act_fwd_strip(const float* z, float* a, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float* save_mean, float* save_rstd, int M, int N, int training, int use_bn,
                                    float eps, float momentum, int act, void* stream) {
    RAT_REQUIRE(rat_bn_strip_ok(M, N) && z && a && act >= 0 && act <= 5, "bad args");
    RAT_REQUIRE(st_aligned(z) && st_aligned(a) && st_aligned(gamma) && st_aligned(beta) && st_aligned(running_mean) &&
                st_aligned(running_var) && st_aligned(save_mean) && st_aligned(save_rstd), "strip kernels need 16-byte aligned pointers");
    if (use_bn) {
        RAT_REQUIRE(gamma && beta && running_mean && running_var, "null BN pointer");
        if (training) RAT_REQUIRE(save_mean && save_rstd, "training BN needs save_mean/save_rstd");
    }
    const int ngroups = (N + 7) / 8, per_xcd = (ngroups + 7) / 8;
    const unsigned blocks = 8u * per_xcd;
    const size_t smem = (size_t)(ST_THREADS / 64) * 2 * 4 * sizeof(float);
#define ST_FWD(R) RAT_LAUNCH((bn_act_fwd_strip_kernel<R>), blocks, ST_THREADS, smem, stream, z, a, gamma, beta, running_mean, running_var, \
                             save_mean, save_rstd, M, N, ngroups, per_xcd, training, use_bn, eps, momentum, act)
    if (M <= 2 * ST_SLOTS) ST_FWD(2);
    else if (M <= 4 * ST_SLOTS) ST_FWD(4);
    else if (M <= 8 * ST_SLOTS) ST_FWD(8);
    else if (M <= 16 * ST_SLOTS) ST_FWD(16);
    else ST_FWD(0);
#undef ST_FWD
    return rat_check_launch("rat_bn_act_fwd_strip");
}

static int bn_act_bwd_strip_launch(const float* z, const float* a, const float* da, float* dz, const float* gamma,
                                   const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, float* dbias_lin,
                                   const float* outer_dl, const float* outer_w, float* outer_dw, int M, int N, int use_bn, int act,
                                   void* stream) {
    RAT_REQUIRE(rat_bn_strip_ok(M, N) && z && a && (da || outer_dl) && dz && act >= 0 && act <= 5, "bad args");
    RAT_REQUIRE(outer_dl == nullptr || (outer_w && outer_dw && st_aligned(outer_w) && st_aligned(outer_dw)), "bad outer-product operands");
    RAT_REQUIRE(st_aligned(z) && st_aligned(a) && st_aligned(da) && st_aligned(dz) && st_aligned(gamma) && st_aligned(save_mean) &&
                st_aligned(save_rstd) && st_aligned(dgamma) && st_aligned(dbeta) && st_aligned(dbias_lin),
                "strip kernels need 16-byte aligned pointers");
    if (use_bn) RAT_REQUIRE(gamma && save_mean && save_rstd && dgamma && dbeta, "null BN pointer");
    const int ngroups = (N + 7) / 8, per_xcd = (ngroups + 7) / 8;
    const unsigned blocks = 8u * per_xcd;
    const size_t smem = (size_t)(ST_THREADS / 64) * 2 * 4 * sizeof(float);
#define ST_BWD(R) RAT_LAUNCH((bn_act_bwd_strip_kernel<R>), blocks, ST_THREADS, smem, stream, z, a, da, dz, gamma, save_mean, save_rstd, \
                             dgamma, dbeta, dbias_lin, outer_dl, outer_w, outer_dw, M, N, ngroups, per_xcd, use_bn, act)
    if (M <= 2 * ST_SLOTS) ST_BWD(2);
    else if (M <= 4 * ST_SLOTS) ST_BWD(4);
    else if (M <= 8 * ST_SLOTS) ST_BWD(8);
    else if (M <= 16 * ST_SLOTS) ST_BWD(16);
    else ST_BWD(0);
#undef ST_BWD
    return rat_check_launch("rat_bn_act_bwd_strip");
}

extern "C" int rat_bn_act_bwd_strip(const float* z, const float* a, const float* da, float* dz, const float* gamma,
                                    const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, float* dbias_lin,
                                    int M, int N, int use_bn, int act, void* stream) {
    RAT_REQUIRE(da != nullptr, "null da");
    return bn_act_bwd_strip_launch(z, a, da, dz, gamma, save_mean, save_rstd, dgamma, dbeta, dbias_lin, nullptr, nullptr, nullptr, M, N,
                                   use_bn, act, stream);
}

// the last hidden layer in front of a ONE-output Linear (the DNN's logit): da = dl (x) w is formed in the kernel, dw[c] = sum_r dl[r] a[r][c]
// comes back with it — replaces rat_sgemm (da = dl w), rat_sgemm (dw = dl^T a) and the strip call above
extern "C" int rat_bn_act_bwd_strip_outer(const float* z, const float* a, const float* dl, const float* w, float* dw, float* dz,
                                          const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma,
                                          float* dbeta, float* dbias_lin, int M, int N, int use_bn, int act, void* stream) {
    RAT_REQUIRE(dl && w && dw, "null outer-product operand");
    return bn_act_bwd_strip_launch(z, a, nullptr, dz, gamma, save_mean, save_rstd, dgamma, dbeta, dbias_lin, dl, w, dw, M, N, use_bn, act,
                                   stream);
}
