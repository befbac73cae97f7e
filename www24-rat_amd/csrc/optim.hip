// optim.hip — K4/K5: L2 regulariser gradient, global grad-norm, and fused clip + Adam over flat fp32 buffers.
//   BaseModel.add_regularization (fuxictr/pytorch/models/base_model.py:79-94, get_regularizer torch_utils.py:65-81),
//   nn.utils.clip_grad_norm_(params, 10.) and Adam.step() (base_model.py:224-225, torch_utils.py:41-49).
// HBM-bound streaming kernels: 16-byte accesses, grid-stride, <= 2048 blocks.
#include "rat_device.h"
#include "../../include/rat_hip.h"
#include <initializer_list>

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

// torch.optim's other optimizers as get_optimizer builds them — getattr(torch.optim, name)(params, lr=lr), torch_utils.py:41-49 — i.e.
// with torch's DEFAULT hyper-parameters: SGD (no momentum), Adagrad (lr_decay 0, initial accumulator 0, eps 1e-10), RMSprop (alpha 0.99,
// eps 1e-8, no momentum, not centered).  Operation order as in torch's single-tensor implementations (addcmul_ / addcdiv_):
//   kind 1 SGD      w -= lr g
//   kind 2 Adagrad  s += g g ;            w -= lr g / (sqrt(s) + eps)
//   kind 3 RMSprop  s = alpha s + ((1 - alpha) g) g ;  w -= lr g / (sqrt(s) + eps)
// (kind 0 = Adam keeps its own code below.)  `v` is the one state buffer these use; `m` is untouched.
__device__ __forceinline__ void opt_other1(int kind, float& w, float gv, float& s, float lr, float p0, float eps) {
    if (kind == 1) {
        w -= lr * gv;
    } else if (kind == 2) {
        s = s + gv * gv;
        w -= lr * gv / (sqrtf(s) + eps);
    } else {
        s = p0 * s;
        s = s + (1.0f - p0) * gv * gv;
        w -= lr * gv / (sqrtf(s) + eps);
    }
}

__global__ void __launch_bounds__(OPT_THREADS)
clip_other_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, int64_t n, const float* norm_sq,
                  float max_norm, float lr, int kind, float p0, float eps) {
    float coef = 1.0f;
    if (norm_sq != nullptr) {
        coef = max_norm / (sqrtf(*norm_sq) + 1e-6f);
        coef = coef < 1.0f ? coef : 1.0f;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float wv = w[i], sv = kind == 1 ? 0.f : v[i];
        opt_other1(kind, wv, g[i] * coef, sv, lr, p0, eps);
        w[i] = wv;
        if (kind != 1) v[i] = sv;
    }
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

// ---- one-sweep forms (ABI v4).  The regulariser gradient lambda*W is never written into g: the norm pass evaluates
// sum (g + lambda w)^2 (and the regulariser's VALUE) from g and w, the Adam pass forms g + lambda w in registers and leaves
// g = 0 behind for the next backward — 2 sweeps (g, w | g, w, m, v -> w, m, v, g) where l2_reg + sumsq + clip_adam + the
// next step's zero-fill made 4 (w, g -> g | g | g, w, m, v -> w, m, v | -> g).  Elements [0, n_split) carry lam_a (the
// "embedding_layer" tensors, base_model.py:86), the rest lam_b (net_regularizer).
__device__ __forceinline__ float4 opt_ld4(const float* p, int64_t i4) { return reinterpret_cast<const float4*>(p)[i4]; }
// streamed-once data (the gradient, the moments): loads AND stores carry the non-temporal hint, so that the sweep leaves the caches
// to W — which the next step's embedding gather reads at random.  Same-box A/B inside the training step (rocprofv3, round 4,
// profiles/round4/r4_nt_ab.txt): rat_clip_adam_fused 377-416 -> 309-315 us, the rat_gather_fwd behind it 99-104 -> 86-91 us; the
// stores alone (round 3's -DRAT_OPT_NT) or the gather's grid stores alone changed nothing.  -DRAT_OPT_PLAIN restores plain accesses.
__device__ __forceinline__ float4 opt_ld4_stream(const float* p, int64_t i4) {
#if !defined(RAT_OPT_PLAIN) && !defined(RAT_EMU)
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i4);
    return make_float4(t[0], t[1], t[2], t[3]);
#else
    return reinterpret_cast<const float4*>(p)[i4];
#endif
}
__device__ __forceinline__ void opt_st4(float* p, int64_t i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }
// streaming store for data nothing reads before the next step's sweep (moments, the zeroed gradient)
__device__ __forceinline__ void opt_st4_stream(float* p, int64_t i4, float4 v) {
#if !defined(RAT_OPT_PLAIN) && !defined(RAT_EMU)
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p) + i4);
#else
    reinterpret_cast<float4*>(p)[i4] = v;
#endif
}

#ifndef RAT_ADAM_UNROLL
#define RAT_ADAM_UNROLL 1               // pieces per trip of clip_adam_fused_kernel's vector loop (A/B: profiles/round5/r5_sumsq_ab.txt)
#endif
#ifndef RAT_SUMSQ_UNROLL
#define RAT_SUMSQ_UNROLL 4
#endif
__global__ void __launch_bounds__(OPT_THREADS)
sumsq_reg_kernel(const float* __restrict__ g, const float* __restrict__ w, int64_t n, int64_t n_split, float lam_a, float lam_b,
                 const float* lam_scale_dev, float* norm_sq_out, float* reg_out, int vec) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const float sc = lam_scale_dev != nullptr ? *lam_scale_dev : 1.0f;
    const float la = lam_a * sc, lb = lam_b * sc;
    float acc = 0.f, wa = 0.f, wb = 0.f;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    int64_t head = 0;
    if (vec) {
        const int64_t n4 = n >> 2, s4 = n_split >> 2;
        auto one = [&](int64_t i, const float4& gv, const float4& wv) {
            const bool a = i < s4;
            const float l = a ? la : lb;
            float t;
            t = fmaf(l, wv.x, gv.x); acc = fmaf(t, t, acc);
            t = fmaf(l, wv.y, gv.y); acc = fmaf(t, t, acc);
            t = fmaf(l, wv.z, gv.z); acc = fmaf(t, t, acc);
            t = fmaf(l, wv.w, gv.w); acc = fmaf(t, t, acc);
            float q = wv.x * wv.x;
            q = fmaf(wv.y, wv.y, q); q = fmaf(wv.z, wv.z, q); q = fmaf(wv.w, wv.w, q);
            if (a) wa += q; else wb += q;
        };
        int64_t i = tid;
        // RAT_SUMSQ_UNROLL pieces per trip, all 2 U loads requested before the first is consumed (same-box A/B: profiles/round5/r5_sumsq_ab.txt)
        for (; i + (int64_t)(RAT_SUMSQ_UNROLL - 1) * nthr < n4; i += (int64_t)RAT_SUMSQ_UNROLL * nthr) {
            float4 gv[RAT_SUMSQ_UNROLL], wv[RAT_SUMSQ_UNROLL];
#pragma unroll
            for (int u = 0; u < RAT_SUMSQ_UNROLL; ++u) {
                gv[u] = opt_ld4_stream(g, i + (int64_t)u * nthr);
                wv[u] = opt_ld4(w, i + (int64_t)u * nthr);
            }
#pragma unroll
            for (int u = 0; u < RAT_SUMSQ_UNROLL; ++u) one(i + (int64_t)u * nthr, gv[u], wv[u]);
        }
        for (; i < n4; i += nthr) one(i, opt_ld4_stream(g, i), opt_ld4(w, i));
        head = n4 << 2;
    }
    for (int64_t i = head + tid; i < n; i += nthr) {
        const bool a = i < n_split;
        const float wv = w[i], t = fmaf(a ? la : lb, wv, g[i]);
        acc = fmaf(t, t, acc);
        if (a) wa = fmaf(wv, wv, wa); else wb = fmaf(wv, wv, wb);
    }
    const float s = opt_block_sum(acc, scratch);
    if (threadIdx.x == 0) atomicAdd(norm_sq_out, s);
    if (reg_out != nullptr) {                                       // (lambda/2) ||W||^2 with the UNSCALED lambdas: the loss term
        const float r = opt_block_sum(0.5f * lam_a * wa + 0.5f * lam_b * wb, scratch);
        if (threadIdx.x == 0) atomicAdd(reg_out, r);
    }
}

__device__ __forceinline__ float opt_adam1(float& w, float g, float& m, float& v, float lam, float coef, float step_size, float beta1,
                                           float beta2, float eps, float inv_sqrt_bc2) {
    const float gv = fmaf(lam, w, g) * coef;
    m = beta1 * m + (1.0f - beta1) * gv;
    v = beta2 * v + (1.0f - beta2) * gv * gv;
    w -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
    return 0.f;
}

__global__ void __launch_bounds__(OPT_THREADS)
clip_adam_fused_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                       int64_t n_split, float lam_a, float lam_b, const float* lam_scale_dev, const float* norm_sq, float max_norm,
                       const float* __restrict__ hyper, float beta1, float beta2, float eps, int zero_g, int vec, int kind) {
    float coef = 1.0f;
    if (norm_sq != nullptr) {
        coef = max_norm / (sqrtf(*norm_sq) + 1e-6f);
        coef = coef < 1.0f ? coef : 1.0f;
    }
    const float sc = lam_scale_dev != nullptr ? *lam_scale_dev : 1.0f;
    const float la = lam_a * sc, lb = lam_b * sc;
    const float step_size = hyper[0], inv_sqrt_bc2 = hyper[1];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    int64_t head = 0;
    if (kind != 0) {                                              // SGD / Adagrad / RMSprop (opt_other1): beta1 carries alpha, hyper[2] = lr
        const float lr = hyper[2];
        for (int64_t i = tid; i < n; i += nthr) {
            float wv = w[i], sv = kind == 1 ? 0.f : v[i];
            opt_other1(kind, wv, fmaf(i < n_split ? la : lb, wv, g[i]) * coef, sv, lr, beta1, eps);
            w[i] = wv;
            if (kind != 1) v[i] = sv;
            if (zero_g) g[i] = 0.f;
        }
        return;
    }
    if (vec) {
        const int64_t n4 = n >> 2, s4 = n_split >> 2;
        auto one = [&](int64_t i, float4 wv, float4 mv, float4 vv, const float4& gv) {
            const float l = i < s4 ? la : lb;
            opt_adam1(wv.x, gv.x, mv.x, vv.x, l, coef, step_size, beta1, beta2, eps, inv_sqrt_bc2);
            opt_adam1(wv.y, gv.y, mv.y, vv.y, l, coef, step_size, beta1, beta2, eps, inv_sqrt_bc2);
            opt_adam1(wv.z, gv.z, mv.z, vv.z, l, coef, step_size, beta1, beta2, eps, inv_sqrt_bc2);
            opt_adam1(wv.w, gv.w, mv.w, vv.w, l, coef, step_size, beta1, beta2, eps, inv_sqrt_bc2);
            opt_st4(w, i, wv); opt_st4_stream(m, i, mv); opt_st4_stream(v, i, vv);
            if (zero_g) opt_st4_stream(g, i, make_float4(0.f, 0.f, 0.f, 0.f));
        };
        int64_t i = tid;
#if RAT_ADAM_UNROLL > 1
        for (; i + (int64_t)(RAT_ADAM_UNROLL - 1) * nthr < n4; i += (int64_t)RAT_ADAM_UNROLL * nthr) {
            float4 wv[RAT_ADAM_UNROLL], mv[RAT_ADAM_UNROLL], vv[RAT_ADAM_UNROLL], gv[RAT_ADAM_UNROLL];
#pragma unroll
            for (int u = 0; u < RAT_ADAM_UNROLL; ++u) {
                const int64_t k = i + (int64_t)u * nthr;
                wv[u] = opt_ld4(w, k); mv[u] = opt_ld4_stream(m, k); vv[u] = opt_ld4_stream(v, k); gv[u] = opt_ld4_stream(g, k);
            }
#pragma unroll
            for (int u = 0; u < RAT_ADAM_UNROLL; ++u) one(i + (int64_t)u * nthr, wv[u], mv[u], vv[u], gv[u]);
        }
#endif
        for (; i < n4; i += nthr) one(i, opt_ld4(w, i), opt_ld4_stream(m, i), opt_ld4_stream(v, i), opt_ld4_stream(g, i));
        head = n4 << 2;
    }
    for (int64_t i = head + tid; i < n; i += nthr) {
        float wv = w[i], mv = m[i], vv = v[i];
        opt_adam1(wv, g[i], mv, vv, i < n_split ? la : lb, coef, step_size, beta1, beta2, eps, inv_sqrt_bc2);
        w[i] = wv; m[i] = mv; v[i] = vv;
        if (zero_g) g[i] = 0.f;
    }
}

// the optimizer's clock on the device (so that a captured step — a hipGraph replay — needs no new kernel arguments):
// *step += 1; hyper[0] = lr / (1 - beta1^step), hyper[1] = 1 / sqrt(1 - beta2^step), hyper[2] = lr   (torch.optim.Adam's
// bias corrections, evaluated in double like the host does)
__global__ void adam_tick_kernel(int32_t* step, const float* lr, float beta1, float beta2, float* hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int s = *step + 1;
    *step = s;
    const double bc1 = 1.0 - pow((double)beta1, (double)s);
    const double bc2 = 1.0 - pow((double)beta2, (double)s);
    hyper[0] = (float)((double)*lr / bc1);
    hyper[1] = (float)(1.0 / sqrt(bc2));
    hyper[2] = *lr;
}

// the bookkeeping a training iteration starts with, in ONE launch: the clock tick above, the step's accumulator scalars cleared
// (BCE sum | clip norm^2 | regulariser value | spare) and every BatchNorm layer's num_batches_tracked advanced
__global__ void step_begin_kernel(int32_t* step, const float* lr, float beta1, float beta2, float* hyper, float* scalars, int nscalars,
                                  int64_t* counters, int ncounters) {
    if (blockIdx.x != 0) return;
    for (int i = threadIdx.x; i < nscalars; i += blockDim.x) scalars[i] = 0.f;
    for (int i = threadIdx.x; i < ncounters; i += blockDim.x) counters[i] += 1;
    if (threadIdx.x != 0) return;
    const int s = *step + 1;
    *step = s;
    const double bc1 = 1.0 - pow((double)beta1, (double)s);
    const double bc2 = 1.0 - pow((double)beta2, (double)s);
    hyper[0] = (float)((double)*lr / bc1);
    hyper[1] = (float)(1.0 / sqrt(bc2));
    hyper[2] = *lr;
}

int opt_blocks(int64_t n) {
    int64_t b = (n + OPT_THREADS * 4 - 1) / (OPT_THREADS * 4);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// grids of the reducing sweeps: every work-group ends with one or two atomic adds onto ONE address, which the L2 serialises (~10 ns
// each).  2048 work-groups hide that behind a 0.5 GB stream; on a small model (BASELINE configs[0]: 2.4 M parameters) they WERE the kernel
// — 50 us for a 10 MB read.  Below 16 M elements a work-group takes at least 8192 of them.
int red_blocks(int64_t n) {
    if (n >= ((int64_t)16 << 20)) return opt_blocks(n);
    int64_t b = (n + 8191) / 8192;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace

extern "C" int rat_l2_reg(const float* w, float* g, int64_t n, float lambda, const float* lambda_scale_dev, float* reg_out,
                          void* stream) {
    RAT_REQUIRE(n > 0 && w && g, "bad args");
    RAT_LAUNCH(l2_reg_kernel, red_blocks(n), OPT_THREADS, 16 * sizeof(float), stream, w, g, n, lambda, lambda_scale_dev, reg_out);
    return rat_check_launch("rat_l2_reg");
}

extern "C" int rat_sumsq(const float* g, int64_t n, float* norm_sq_out, void* stream) {
    RAT_REQUIRE(n > 0 && g && norm_sq_out, "bad args");
    RAT_LAUNCH(sumsq_kernel, red_blocks(n), OPT_THREADS, 16 * sizeof(float), stream, g, n, norm_sq_out);
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

static bool opt_vec_ok(std::initializer_list<const void*> ps, int64_t n_split) {
    for (const void* p : ps)
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    return (n_split & 3) == 0;
}

extern "C" int rat_adam_tick(int32_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_out, void* stream) {
    RAT_REQUIRE(step_dev && lr_dev && hyper_out, "bad args");
    RAT_LAUNCH(adam_tick_kernel, 1, 64, 0, stream, step_dev, lr_dev, beta1, beta2, hyper_out);
    return rat_check_launch("rat_adam_tick");
}

extern "C" int rat_step_begin(int32_t* step_dev, const float* lr_dev, float beta1, float beta2, float* hyper_out, float* scalars,
                              int nscalars, int64_t* counters, int ncounters, void* stream) {
    RAT_REQUIRE(step_dev && lr_dev && hyper_out && nscalars >= 0 && ncounters >= 0 && (scalars || nscalars == 0) && (counters || ncounters == 0),
                "bad args");
    RAT_LAUNCH(step_begin_kernel, 1, 64, 0, stream, step_dev, lr_dev, beta1, beta2, hyper_out, scalars, nscalars, counters, ncounters);
    return rat_check_launch("rat_step_begin");
}

extern "C" int rat_sumsq_reg(const float* g, const float* w, int64_t n, int64_t n_split, float lam_a, float lam_b,
                             const float* lam_scale_dev, float* norm_sq_out, float* reg_out, void* stream) {
    RAT_REQUIRE(n > 0 && g && w && norm_sq_out && n_split >= 0 && n_split <= n, "bad args");
    const int vec = opt_vec_ok({g, w}, n_split) ? 1 : 0;
    RAT_LAUNCH(sumsq_reg_kernel, red_blocks(n), OPT_THREADS, 16 * sizeof(float), stream, g, w, n, n_split, lam_a, lam_b, lam_scale_dev,
               norm_sq_out, reg_out, vec);
    return rat_check_launch("rat_sumsq_reg");
}

extern "C" int rat_clip_adam_fused(float* w, float* g, float* m, float* v, int64_t n, int64_t n_split, float lam_a, float lam_b,
                                   const float* lam_scale_dev, const float* norm_sq, float max_norm, const float* hyper_dev,
                                   float beta1, float beta2, float eps, int zero_g, void* stream) {
    RAT_REQUIRE(n > 0 && w && g && m && v && hyper_dev && n_split >= 0 && n_split <= n, "bad args");
    const int vec = opt_vec_ok({w, g, m, v}, n_split) ? 1 : 0;
    RAT_LAUNCH(clip_adam_fused_kernel, opt_blocks(n), OPT_THREADS, 0, stream, w, g, m, v, n, n_split, lam_a, lam_b, lam_scale_dev,
               norm_sq, max_norm, hyper_dev, beta1, beta2, eps, zero_g, vec, 0);
    return rat_check_launch("rat_clip_adam_fused");
}

extern "C" int rat_clip_opt_fused(float* w, float* g, float* state, int64_t n, int64_t n_split, float lam_a, float lam_b,
                                  const float* lam_scale_dev, const float* norm_sq, float max_norm, const float* hyper_dev, int kind,
                                  float p0, float eps, int zero_g, void* stream) {
    RAT_REQUIRE(n > 0 && w && g && hyper_dev && n_split >= 0 && n_split <= n && kind >= 1 && kind <= 3 && (kind == 1 || state), "bad args");
    RAT_LAUNCH(clip_adam_fused_kernel, opt_blocks(n), OPT_THREADS, 0, stream, w, g, (float*)nullptr, state, n, n_split, lam_a, lam_b,
               lam_scale_dev, norm_sq, max_norm, hyper_dev, p0, 0.f, eps, zero_g, 0, kind);
    return rat_check_launch("rat_clip_opt_fused");
}

extern "C" int rat_clip_opt(float* w, const float* g, float* state, int64_t n, const float* norm_sq, float max_norm, float lr, int kind,
                            float p0, float eps, void* stream) {
    RAT_REQUIRE(n > 0 && w && g && kind >= 1 && kind <= 3 && (kind == 1 || state), "bad args");
    RAT_LAUNCH(clip_other_kernel, opt_blocks(n), OPT_THREADS, 0, stream, w, g, state, n, norm_sq, max_norm, lr, kind, p0, eps);
    return rat_check_launch("rat_clip_opt");
}

// ---- inverted dropout with a counter-based generator (nn.Dropout of RAT_m2.py:83,135 `emb_dropout` and deep.py:133-134
// `net_dropout`).  mask(i) depends only on (seed, i), so backward re-derives it instead of storing it:
// y[i] = keep(seed, i) ? x[i] / (1 - p) : 0.  torch's Philox stream cannot be matched bit for bit; parity for p > 0 is
// statistical (SURVEY.md §7 hard part 5).
namespace {

__global__ void __launch_bounds__(OPT_THREADS)
dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, uint32_t threshold, float scale, uint64_t seed,
               const uint64_t* __restrict__ seed_dev) {
    if (seed_dev != nullptr) seed = *seed_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = rat_hash32(seed, (uint64_t)i) >= threshold ? x[i] * scale : 0.f;
}
// one training step's seeds: counter += 1, seeds[i] = mix(base, counter, i) — the whole state of the step's dropout masks lives on
// the device, so a captured step (graph.StepGraph) draws new masks on every replay without a host call
__global__ void dropout_seeds_kernel(uint64_t* seeds, int n, uint64_t base, uint64_t* counter) {
    const uint64_t c = counter[0] + 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t hi = rat_hash32(base ^ (0xD1B54A32D192ED03ull * c), (uint64_t)(2 * i));
        const uint64_t lo = rat_hash32(base + c, (uint64_t)(2 * i + 1));
        seeds[i] = (hi << 32) | lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) counter[0] = c;
}
}  // namespace

extern "C" int rat_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
    RAT_REQUIRE(n > 0 && x && y && p >= 0.f && p < 1.f, "bad args");
    const uint32_t threshold = (uint32_t)((double)p * 4294967296.0);
    RAT_LAUNCH(dropout_kernel, opt_blocks(n), OPT_THREADS, 0, stream, x, y, n, threshold, 1.0f / (1.0f - p), seed,
               (const uint64_t*)nullptr);
    return rat_check_launch("rat_dropout");
}

extern "C" int rat_dropout_dev(const float* x, float* y, int64_t n, float p, const uint64_t* seed_dev, void* stream) {
    RAT_REQUIRE(n > 0 && x && y && seed_dev && p >= 0.f && p < 1.f, "bad args");
    const uint32_t threshold = (uint32_t)((double)p * 4294967296.0);
    RAT_LAUNCH(dropout_kernel, opt_blocks(n), OPT_THREADS, 0, stream, x, y, n, threshold, 1.0f / (1.0f - p), (uint64_t)0, seed_dev);
    return rat_check_launch("rat_dropout_dev");
}

extern "C" int rat_dropout_seeds(uint64_t* seeds_dev, int n, uint64_t base_seed, uint64_t* counter_dev, void* stream) {
    RAT_REQUIRE(seeds_dev && counter_dev && n >= 1, "bad args");
    RAT_LAUNCH(dropout_seeds_kernel, 1, 64, 0, stream, seeds_dev, n, base_seed, counter_dev);
    return rat_check_launch("rat_dropout_seeds");
}
