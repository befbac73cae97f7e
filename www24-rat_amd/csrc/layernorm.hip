// layernorm.hip — K2c: stand-alone LayerNorm over selected token rows, forward and backward.
//
// The cascaded variant RAT_m1 (RAT_m1.py:166-209) needs LayerNorm outside the attention kernel: in front of the block
// MLP (PreNorm(FeedForward), RAT_m1.py:201) and as the final `self.norm` of each Transformer (RAT_m1.py:198,209), of
// which only token 0 of every sequence is read afterwards (RAT_m1.py:125,128) — hence the ROW STRIDES: the kernel
// normalises row r at x + r * x_stride and writes a compact [nrows][d] result; backward scatters dx back to the strided
// rows (the caller zero-fills the rest), optionally adding a residual gradient laid out like dx.
//
// HBM-bound: 16 lanes per row (4 rows per wave), every lane keeps its d/16 values in registers (two-pass mean /
// variance, biased, like nn.LayerNorm).  dgamma / dbeta are per-work-group partials reduced in a fixed order.
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_LPR = 16;                         // lanes per row
constexpr int LN_RPB = LN_THREADS / LN_LPR;        // rows per work-group pass
constexpr int LN_MAXPER = 32;                      // d <= 512
constexpr int LN_MAXBLOCKS = 1024;

struct LnArgs {
    const float* x;
    int64_t x_stride;
    const float* dy;           // backward: [nrows][d] compact
    const float* gamma;
    const float* beta;
    const float* add;          // backward: optional residual gradient, laid out like dx
    float* y;                  // forward: [nrows][d] compact / backward: dx (strided)
    int64_t y_stride;
    float* slabs;              // backward: [blocks][2 d] = dgamma | dbeta partials
    int64_t nrows;
    int d;
    float eps;
};

template <int PER>
__global__ void __launch_bounds__(LN_THREADS) ln_fwd_kernel(LnArgs a) {
    const int sub = threadIdx.x % LN_LPR, slot = threadIdx.x / LN_LPR;
    float gam[PER], bet[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = sub + LN_LPR * k;
        gam[k] = c < a.d ? a.gamma[c] : 0.f;
        bet[k] = c < a.d ? a.beta[c] : 0.f;
    }
    for (int64_t r0 = (int64_t)blockIdx.x * LN_RPB; r0 < a.nrows; r0 += (int64_t)gridDim.x * LN_RPB) {
        const int64_t r = r0 + slot;
        const bool on = r < a.nrows;
        const float* xr = a.x + (on ? r : 0) * a.x_stride;
        float v[PER];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = sub + LN_LPR * k;
            v[k] = (on && c < a.d) ? xr[c] : 0.f;
            s += v[k];
        }
        const float mean = rat_group_sum<LN_LPR>(s) / (float)a.d;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = sub + LN_LPR * k;
            const float t = c < a.d ? v[k] - mean : 0.f;
            q += t * t;
        }
        const float rstd = 1.0f / sqrtf(rat_group_sum<LN_LPR>(q) / (float)a.d + a.eps);
        if (on) {
            float* yr = a.y + r * a.y_stride;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = sub + LN_LPR * k;
                if (c < a.d) yr[c] = (v[k] - mean) * rstd * gam[k] + bet[k];
            }
        }
    }
}

template <int PER>
__global__ void __launch_bounds__(LN_THREADS) ln_bwd_kernel(LnArgs a) {
    __shared__ float red[LN_RPB][LN_LPR * PER + 1];
    const int sub = threadIdx.x % LN_LPR, slot = threadIdx.x / LN_LPR;
    float gam[PER], dg[PER], db[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int c = sub + LN_LPR * k;
        gam[k] = c < a.d ? a.gamma[c] : 0.f;
        dg[k] = db[k] = 0.f;
    }
    for (int64_t r0 = (int64_t)blockIdx.x * LN_RPB; r0 < a.nrows; r0 += (int64_t)gridDim.x * LN_RPB) {
        const int64_t r = r0 + slot;
        const bool on = r < a.nrows;
        const float* xr = a.x + (on ? r : 0) * a.x_stride;
        const float* gr = a.dy + (on ? r : 0) * (int64_t)a.d;
        float v[PER], g[PER];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = sub + LN_LPR * k;
            const bool in = on && c < a.d;
            v[k] = in ? xr[c] : 0.f;
            g[k] = in ? gr[c] : 0.f;
            s += v[k];
        }
        const float mean = rat_group_sum<LN_LPR>(s) / (float)a.d;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int c = sub + LN_LPR * k;
            v[k] = c < a.d ? v[k] - mean : 0.f;
            q += v[k] * v[k];
        }
        const float rstd = 1.0f / sqrtf(rat_group_sum<LN_LPR>(q) / (float)a.d + a.eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            v[k] *= rstd;                                     // xhat
            db[k] += g[k];
            dg[k] += g[k] * v[k];
            g[k] *= gam[k];                                   // d(xhat)
            s1 += g[k];
            s2 += g[k] * v[k];
        }
        s1 = rat_group_sum<LN_LPR>(s1) / (float)a.d;
        s2 = rat_group_sum<LN_LPR>(s2) / (float)a.d;
        if (on) {
            float* dxr = a.y + r * a.y_stride;
            const float* ar = a.add ? a.add + r * a.y_stride : nullptr;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = sub + LN_LPR * k;
                if (c < a.d) {
                    const float t = rstd * (g[k] - s1 - v[k] * s2);
                    dxr[c] = ar ? ar[c] + t : t;
                }
            }
        }
    }
    // per-work-group dgamma | dbeta: row slots combined through LDS in a fixed order
    float* slab = a.slabs + (int64_t)blockIdx.x * 2 * a.d;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) red[slot][sub + LN_LPR * k] = pass ? db[k] : dg[k];
        __syncthreads();
        for (int c = threadIdx.x; c < a.d; c += LN_THREADS) {
            float t = 0.f;
            for (int s = 0; s < LN_RPB; ++s) t += red[s][c];
            slab[pass * a.d + c] = t;
        }
    }
}

int ln_blocks(int64_t nrows) {
    const int64_t need = (nrows + LN_RPB - 1) / LN_RPB;
    return (int)(need < LN_MAXBLOCKS ? need : LN_MAXBLOCKS);
}

template <template <int> class Launcher>
int ln_dispatch(int d, const LnArgs& a, int blocks, void* stream) {
    const int per = (d + LN_LPR - 1) / LN_LPR;
    if (per <= 1) return Launcher<1>::go(a, blocks, stream);
    if (per <= 4) return Launcher<4>::go(a, blocks, stream);
    if (per <= 8) return Launcher<8>::go(a, blocks, stream);
    return Launcher<LN_MAXPER>::go(a, blocks, stream);
}
template <int PER>
struct FwdLauncher {
    static int go(const LnArgs& a, int blocks, void* stream) {
        RAT_LAUNCH((ln_fwd_kernel<PER>), blocks, LN_THREADS, 0, stream, a);
        return rat_check_launch("rat_layernorm_fwd");
    }
};
template <int PER>
struct BwdLauncher {
    static int go(const LnArgs& a, int blocks, void* stream) {
        RAT_LAUNCH((ln_bwd_kernel<PER>), blocks, LN_THREADS, 0, stream, a);
        return rat_check_launch("rat_layernorm_bwd");
    }
};

int ln_check(int64_t nrows, int d, int64_t x_stride, int64_t y_stride) {
    RAT_REQUIRE(nrows > 0 && d > 0, "bad dims");
    RAT_REQUIRE(d <= LN_LPR * LN_MAXPER, "d above 512 not supported");
    RAT_REQUIRE(x_stride >= d && y_stride >= d, "row stride smaller than d");
    return 0;
}

}  // namespace

extern "C" int rat_layernorm_fwd(const float* x, int64_t x_stride, float* y, const float* gamma, const float* beta,
                                 int64_t nrows, int d, float eps, void* stream) {
    if (ln_check(nrows, d, x_stride, d)) return -1;
    RAT_REQUIRE(x && y && gamma && beta, "null pointer");
    LnArgs a{};
    a.x = x;
    a.x_stride = x_stride;
    a.y = y;
    a.y_stride = d;
    a.gamma = gamma;
    a.beta = beta;
    a.nrows = nrows;
    a.d = d;
    a.eps = eps;
    return ln_dispatch<FwdLauncher>(d, a, ln_blocks(nrows), stream);
}

extern "C" size_t rat_layernorm_bwd_workspace(int64_t nrows, int d) {
    return (size_t)ln_blocks(nrows > 0 ? nrows : 1) * 2 * (size_t)(d > 0 ? d : 0) * sizeof(float);
}

extern "C" int rat_layernorm_bwd(const float* x, int64_t x_stride, const float* dy, const float* gamma, const float* add,
                                 float* dx, int64_t dx_stride, float* dgamma, float* dbeta, float* workspace,
                                 size_t workspace_bytes, int64_t nrows, int d, float eps, void* stream) {
    if (ln_check(nrows, d, x_stride, dx_stride)) return -1;
    RAT_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && workspace, "null pointer");
    RAT_REQUIRE(workspace_bytes >= rat_layernorm_bwd_workspace(nrows, d), "workspace too small");
    LnArgs a{};
    a.x = x;
    a.x_stride = x_stride;
    a.dy = dy;
    a.gamma = gamma;
    a.add = add;
    a.y = dx;
    a.y_stride = dx_stride;
    a.slabs = workspace;
    a.nrows = nrows;
    a.d = d;
    a.eps = eps;
    const int blocks = ln_blocks(nrows);
    if (ln_dispatch<BwdLauncher>(d, a, blocks, stream)) return -1;
    float* outs[2] = {dgamma, dbeta};
    const int64_t sizes[2] = {d, d};
    const int64_t offs[2] = {0, d};
    return rat_launch_reduce_slabs(workspace, blocks, 2 * (int64_t)d, outs, offs, sizes, 2, stream);
}
