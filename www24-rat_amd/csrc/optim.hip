// optim.hip — K4/K5: L2 regulariser gradient, global grad-norm, and fused clip + Adam over flat fp32 buffers.
//   BaseModel.add_regularization (fuxictr/pytorch/models/base_model.py:79-94, get_regularizer torch_utils.py:65-81),
//   nn.utils.clip_grad_norm_(params, 10.) and Adam.step() (base_model.py:224-225, torch_utils.py:41-49).
// HBM-bound streaming kernels: 16-byte accesses, grid-stride, <= 2048 blocks.
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int OPT_THREADS = 256;

__device__ __forceinline__ float opt_block_sum(float v, float* scratch) {
    v = rat_group_sum<64>(v);
    __syncthreads();
    if (rat_lane() == 0) scratch[rat_wave()] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ void __launch_bounds__(OPT_THREADS)
l2_reg_kernel(const float* __restrict__ w, float* __restrict__ g, int64_t n, float lambda, const float* lambda_scale_dev,
              float* reg_out) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    if (lambda_scale_dev != nullptr) lambda *= *lambda_scale_dev;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wv = w[i];
        g[i] += lambda * wv;
        acc = fmaf(wv, wv, acc);
    }
    if (reg_out != nullptr) {
        const float s = opt_block_sum(acc, scratch);
        if (threadIdx.x == 0) atomicAdd(reg_out, 0.5f * lambda * s);
    }
}

__global__ void __launch_bounds__(OPT_THREADS)
sumsq_kernel(const float* __restrict__ g, int64_t n, float* out) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    float acc = 0.f;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    int64_t head = 0;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {                 // 16-byte requests, two in flight per thread (HBM-bound sweep)
        const int64_t n4 = n >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float a0 = 0.f, a1 = 0.f;
        int64_t i = tid;
        for (; i + nthr < n4; i += 2 * nthr) {
            const float4 u = g4[i], w = g4[i + nthr];
            a0 = fmaf(u.x, u.x, a0); a0 = fmaf(u.y, u.y, a0); a0 = fmaf(u.z, u.z, a0); a0 = fmaf(u.w, u.w, a0);
            a1 = fmaf(w.x, w.x, a1); a1 = fmaf(w.y, w.y, a1); a1 = fmaf(w.z, w.z, a1); a1 = fmaf(w.w, w.w, a1);
        }
        if (i < n4) {
            const float4 u = g4[i];
            a0 = fmaf(u.x, u.x, a0); a0 = fmaf(u.y, u.y, a0); a0 = fmaf(u.z, u.z, a0); a0 = fmaf(u.w, u.w, a0);
        }
        acc = a0 + a1;
        head = n4 << 2;
    }
    for (int64_t i = head + tid; i < n; i += nthr) {
        const float v = g[i];
        acc = fmaf(v, v, acc);
    }
    const float s = opt_block_sum(acc, scratch);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__global__ void __launch_bounds__(OPT_THREADS)
clip_adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                 const float* norm_sq, float max_norm, float step_size, float beta1, float beta2, float eps,
                 float inv_sqrt_bc2) {
    float coef = 1.0f;
    if (norm_sq != nullptr) {
        coef = max_norm / (sqrtf(*norm_sq) + 1e-6f);
        coef = coef < 1.0f ? coef : 1.0f;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gv = g[i] * coef;
        const float mv = beta1 * m[i] + (1.0f - beta1) * gv;
        const float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        w[i] -= step_size * mv / denom;
    }
}

int opt_blocks(int64_t n) {
    int64_t b = (n + OPT_THREADS * 4 - 1) / (OPT_THREADS * 4);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace

extern "C" int rat_l2_reg(const float* w, float* g, int64_t n, float lambda, const float* lambda_scale_dev, float* reg_out,
                          void* stream) {
    RAT_REQUIRE(n > 0 && w && g, "bad args");
    RAT_LAUNCH(l2_reg_kernel, opt_blocks(n), OPT_THREADS, 16 * sizeof(float), stream, w, g, n, lambda, lambda_scale_dev, reg_out);
    return rat_check_launch("rat_l2_reg");
}

extern "C" int rat_sumsq(const float* g, int64_t n, float* norm_sq_out, void* stream) {
    RAT_REQUIRE(n > 0 && g && norm_sq_out, "bad args");
    RAT_LAUNCH(sumsq_kernel, opt_blocks(n), OPT_THREADS, 16 * sizeof(float), stream, g, n, norm_sq_out);
    return rat_check_launch("rat_sumsq");
}

extern "C" int rat_clip_adam(float* w, const float* g, float* m, float* v, int64_t n, const float* norm_sq, float max_norm,
                             float lr, float beta1, float beta2, float eps, int step, void* stream) {
    RAT_REQUIRE(n > 0 && w && g && m && v && step >= 1, "bad args");
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    RAT_LAUNCH(clip_adam_kernel, opt_blocks(n), OPT_THREADS, 0, stream, w, g, m, v, n, norm_sq, max_norm,
               (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)));
    return rat_check_launch("rat_clip_adam");
}

// ---- inverted dropout with a counter-based generator (nn.Dropout of RAT_m2.py:83,135 `emb_dropout` and deep.py:133-134
// `net_dropout`).  mask(i) depends only on (seed, i), so backward re-derives it instead of storing it:
// y[i] = keep(seed, i) ? x[i] / (1 - p) : 0.  torch's Philox stream cannot be matched bit for bit; parity for p > 0 is
// statistical (SURVEY.md §7 hard part 5).
namespace {

__global__ void __launch_bounds__(OPT_THREADS)
dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, uint32_t threshold, float scale, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = rat_hash32(seed, (uint64_t)i) >= threshold ? x[i] * scale : 0.f;
}
}  // namespace

extern "C" int rat_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    RAT_REQUIRE(n > 0 && x && y && p >= 0.f && p < 1.f, "bad args");
    const uint32_t threshold = (uint32_t)((double)p * 4294967296.0);
    RAT_LAUNCH(dropout_kernel, opt_blocks(n), OPT_THREADS, 0, stream, x, y, n, threshold, 1.0f / (1.0f - p), seed);
    return rat_check_launch("rat_dropout");
}
