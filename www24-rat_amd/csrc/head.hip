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

constexpr int LB_SAMPLES = 64;              // samples per block of logit_bwd (one wave computes their dlogit)
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

__global__ void __launch_bounds__(HD_THREADS)
logit_fwd_kernel(const float* __restrict__ cls, int64_t cls_stride, const float* fc_w, const float* fc_b,
                 const float* dnn_out, const RatField* lr_fields, int nfields, const int32_t* idx, int64_t idx_stride,
                 const float* y_true, float* y_pred, float* loss_sum, int B, int d, int head) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.f;
    if (b < B) {
        float zl = fc_b[0];
        const float* c = cls + (int64_t)b * cls_stride;
        for (int k = 0; k < d; ++k) zl = fmaf(c[k], fc_w[k], zl);
        if (dnn_out != nullptr) zl += dnn_out[b];
        if (lr_fields != nullptr) {
            float lr = 0.f;
            for (int f = 0; f < nfields; ++f) {
                const RatField fd = lr_fields[f];
                const int32_t* ids = idx + (int64_t)b * idx_stride + fd.col;
                for (int j = 0; j < fd.ncols; ++j) {
                    int id = ids[j];
                    id = id < 0 ? 0 : (id >= fd.vocab ? fd.vocab - 1 : id);
                    lr += fd.table[id];
                }
            }
            zl += lr;
        }
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
                 float* dfc_b, const RatField* lr_grad_fields, int nfields, const int32_t* idx, int64_t idx_stride,
                 float gscale, const float* gscale_dev, int B, int d, int head) {
    // One block = LB_SAMPLES samples.  Phase 1: one thread per sample (dlogit, LR-table atomics, dfc_b partial).  Phase 2:
    // thread = (column k, sample group): dcls rows and the dfc_w partial sums are produced column-parallel — no LDS atomics
    // (the first version issued d LDS atomics per sample onto the same d addresses and ran on B/256 CUs only).
    RAT_DYN_SMEM(smem);
    float* dls = reinterpret_cast<float*>(smem);              // [LB_SAMPLES] dlogit of this block's samples
    float* part = dls + LB_SAMPLES;                           // [groups][d] partial dfc_w
    const int b0 = blockIdx.x * LB_SAMPLES;
    const int nb = B - b0 < LB_SAMPLES ? B - b0 : LB_SAMPLES;
    if (gscale_dev != nullptr) gscale *= *gscale_dev;         // the incoming loss gradient stays on the device (no host read-back)
    float dbias = 0.f;
    if ((int)threadIdx.x < LB_SAMPLES) {
        float dl = 0.f;
        if ((int)threadIdx.x < nb) {
            const int b = b0 + threadIdx.x;
            dl = gscale * (head == 1 ? 2.0f : 1.0f) * (y_pred[b] - y_true[b]) / (float)B;    // d BCE(sigmoid z) / dz = p - t; d MSE / dz = 2 (z - t)
            dlogit[b] = dl;
            if (lr_grad_fields != nullptr)
                for (int f = 0; f < nfields; ++f) {
                    const RatField fd = lr_grad_fields[f];
                    const int32_t* ids = idx + (int64_t)b * idx_stride + fd.col;
                    for (int j = 0; j < fd.ncols; ++j) {
                        int id = ids[j];
                        id = id < 0 ? 0 : (id >= fd.vocab ? fd.vocab - 1 : id);
                        if (id != fd.padding_idx) atomicAdd(fd.table + id, dl);
                    }
                }
        }
        dls[threadIdx.x] = dl;
        dbias = rat_group_sum<64>(dl);                        // LB_SAMPLES == 64: one wave
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
    if (threadIdx.x == 0) atomicAdd(dfc_b, dbias);
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

extern "C" int rat_logit_fwd(const float* cls, int64_t cls_stride, const float* fc_w, const float* fc_b, const float* dnn_out,
                             const RatField* lr_fields_dev, int nfields, const int32_t* idx, int64_t idx_stride,
                             const float* y_true, float* y_pred, float* loss_sum, int B, int d, int head, void* stream) {
    RAT_REQUIRE(B > 0 && d > 0 && cls && fc_w && fc_b && y_pred && (head == 0 || head == 1), "bad args");
    RAT_REQUIRE(lr_fields_dev == nullptr || idx != nullptr, "LR term needs idx");
    RAT_LAUNCH(logit_fwd_kernel, (B + HD_THREADS - 1) / HD_THREADS, HD_THREADS, 16 * sizeof(float), stream, cls, cls_stride,
               fc_w, fc_b, dnn_out, lr_fields_dev, nfields, idx, idx_stride, y_true, y_pred, loss_sum, B, d, head);
    return rat_check_launch("rat_logit_fwd");
}

extern "C" int rat_logit_bwd(const float* y_pred, const float* y_true, const float* cls, int64_t cls_stride,
                             const float* fc_w, float* dlogit, float* dcls, int64_t dcls_stride, float* dfc_w, float* dfc_b,
                             const RatField* lr_grad_fields_dev, int nfields, const int32_t* idx, int64_t idx_stride,
                             float gscale, const float* gscale_dev, int B, int d, int head, void* stream) {
    RAT_REQUIRE(B > 0 && d > 0 && y_pred && y_true && cls && fc_w && dlogit && dcls && dfc_w && dfc_b && (head == 0 || head == 1), "bad args");
    RAT_REQUIRE(lr_grad_fields_dev == nullptr || idx != nullptr, "LR term needs idx");
    RAT_REQUIRE(d <= HD_THREADS, "embedding_dim above the block size");
    RAT_LAUNCH(logit_bwd_kernel, (B + LB_SAMPLES - 1) / LB_SAMPLES, HD_THREADS, (size_t)(LB_SAMPLES + HD_THREADS) * sizeof(float), stream,
               y_pred, y_true, cls, cls_stride, fc_w, dlogit, dcls, dcls_stride, dfc_w, dfc_b, lr_grad_fields_dev, nfields,
               idx, idx_stride, gscale, gscale_dev, B, d, head);
    return rat_check_launch("rat_logit_bwd");
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
