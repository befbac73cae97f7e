// sparse.hip — the row-sparse / deterministic embedding-gradient path (BASELINE.json configs[3]: 100 M rows x 8 GPUs).
//
// The reference produces DENSE table gradients (nn.Embedding(sparse=False), fuxictr/pytorch/layers/embedding.py:158-178;
// autograd's embedding_dense_backward; clip_grad_norm_ + Adam over every row, base_model.py:221-225, torch_utils.py:41-49).
// That cannot be carried to a 25.6 GB table: a dense gradient + a dense Adam pass + a 25.6 GB all-reduce per step.  Here a
// step touches only the rows its batch names:
//
//   plan    : every (sample, id column) pair of the batch -> key = the id's GLOBAL row in the flat table block; stable radix
//             sort of (key, pair position); segment heads -> unique rows.  (The device radix sort / scan are rocPRIM's —
//             library primitives like a GEMM would be; everything around them is written here.)
//   reduce  : one lane group per unique row sums the gradient rows of its segment IN SORTED ORDER (the sort is stable, so the
//             order is the batch order: bit-reproducible, no atomics) -> (row ids, gradient rows), or straight into a dense
//             gradient table (the deterministic replacement of rat_gather_bwd's fp32 atomics, selectable for every config);
//   exchange: data parallelism all-gathers (row ids, gradient rows) and runs plan + reduce again over the gathered lists;
//   update  : rat_adam_rows — clip coefficient + Adam on the touched rows only ("lazy" Adam: the moments of untouched rows do
//             not decay, the bias correction uses the global step; declared deviation from the reference's dense Adam,
//             exact when every row is touched every step), rat_sumsq_rows for the global gradient norm.
#include "rat_device.h"
#include "../../include/rat_hip.h"

#ifdef RAT_EMU
#include <algorithm>
#include <numeric>
#else
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#endif

namespace {

constexpr int SP_THREADS = 256;

struct PlanView {            // carve-up of the caller's workspace for n entries
    uint32_t* keys_a;
    uint32_t* keys_b;        // sorted keys
    uint32_t* vals_a;
    uint32_t* vals_b;        // sorted pair positions
    uint32_t* seg_id;        // inclusive scan of the head flags
    uint32_t* seg_start;     // [n + 1]
    void* temp;              // rocPRIM temporary storage
    size_t temp_bytes;
};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

size_t prim_temp_bytes(int64_t n) {
#ifdef RAT_EMU
    (void)n;
    return 256;
#else
    size_t sort_b = 0, scan_b = 0;
    uint32_t* kp = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sort_b, kp, kp, kp, kp, (size_t)n, 0, 32, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, scan_b, kp, kp, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return align256(sort_b > scan_b ? sort_b : scan_b) + 256;
#endif
}

PlanView carve(void* ws, int64_t n) {
    PlanView v{};
    char* p = static_cast<char*>(ws);
    const size_t arr = align256((size_t)(n + 1) * sizeof(uint32_t));
    v.keys_a = reinterpret_cast<uint32_t*>(p); p += arr;
    v.keys_b = reinterpret_cast<uint32_t*>(p); p += arr;
    v.vals_a = reinterpret_cast<uint32_t*>(p); p += arr;
    v.vals_b = reinterpret_cast<uint32_t*>(p); p += arr;
    v.seg_id = reinterpret_cast<uint32_t*>(p); p += arr;
    v.seg_start = reinterpret_cast<uint32_t*>(p); p += arr;
    v.temp = p;
    return v;
}

int sp_blocks(int64_t n) {
    int64_t b = (n + SP_THREADS - 1) / SP_THREADS;
    return (int)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

// ---- keys -------------------------------------------------------------------------------------------------------
// ids of the batch: entry e = (row bt of idx, id column c); target_only: only the rows with t == 0 (stride T) take part
__global__ void __launch_bounds__(SP_THREADS)
keys_from_ids_kernel(const int32_t* __restrict__ idx, const RatField* __restrict__ fields, const int32_t* __restrict__ col2field,
                     const float* flat_base, int width, uint32_t invalid, int64_t nrows, int T, int L, int target_only,
                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t n = nrows * L;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / L;
        const int c = (int)(e - r * L);
        const int64_t bt = target_only ? r * T : r;
        const int fi = col2field[c];
        uint32_t key = invalid;
        if (fi >= 0) {
            const RatField f = fields[fi];
            // out-of-vocabulary ids: ONE policy on every path (forward gather, atomic scatter, this plan) — clamped into the table
            // for memory safety and REPORTED by rat_check_ids; the gradient goes to the row the forward read
            int id = idx[bt * L + c];
            id = id < 0 ? 0 : (id >= f.vocab ? f.vocab - 1 : id);
            if (id != f.padding_idx)
                key = (uint32_t)((f.table - flat_base) / width + id);
        }
        keys[e] = key;
        vals[e] = (uint32_t)e;
    }
}

// gathered row lists of `world` ranks, each `cap` long with counts[r] valid entries
__global__ void __launch_bounds__(SP_THREADS)
keys_from_rows_kernel(const int32_t* __restrict__ rows, const int32_t* __restrict__ counts, int64_t cap, int world, uint32_t invalid,
                      uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t n = cap * world;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cap;
        const int64_t i = e - r * cap;
        const int32_t row = rows[e];
        keys[e] = (i < counts[r] && row >= 0 && (uint32_t)row < invalid) ? (uint32_t)row : invalid;
        vals[e] = (uint32_t)e;
    }
}

__global__ void __launch_bounds__(SP_THREADS)
head_flags_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ flags, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// seg_start[s] = first sorted position of segment s; count = number of segments with a valid key; seg_start[count] = end
__global__ void __launch_bounds__(SP_THREADS)
segment_starts_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ seg_id, uint32_t invalid, int64_t n,
                      uint32_t* __restrict__ seg_start, int32_t* __restrict__ count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool head = i == 0 || keys[i] != keys[i - 1];
        if (head) {
            seg_start[seg_id[i] - 1] = (uint32_t)i;               // for the invalid tail this IS seg_start[count]
            if (keys[i] == invalid) *count = (int32_t)(seg_id[i] - 1);
        }
        if (i == n - 1 && keys[i] != invalid) {
            *count = (int32_t)seg_id[i];
            seg_start[seg_id[i]] = (uint32_t)n;
        }
    }
}

int build_segments(PlanView& v, int64_t n, uint32_t invalid, unsigned end_bit, int32_t* count, void* stream) {
#ifdef RAT_EMU
    (void)end_bit;
    (void)stream;
    std::vector<uint32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return v.keys_a[a] < v.keys_a[b]; });
    for (int64_t i = 0; i < n; ++i) {
        v.keys_b[i] = v.keys_a[order[(size_t)i]];
        v.vals_b[i] = v.vals_a[order[(size_t)i]];
    }
    uint32_t run = 0;
    for (int64_t i = 0; i < n; ++i) {
        run += (i == 0 || v.keys_b[i] != v.keys_b[i - 1]) ? 1u : 0u;
        v.seg_id[i] = run;
    }
#else
    hipStream_t s = (hipStream_t)stream;
    size_t tb = v.temp_bytes;
    if (rocprim::radix_sort_pairs(v.temp, tb, v.keys_a, v.keys_b, v.vals_a, v.vals_b, (size_t)n, 0, end_bit, s) != hipSuccess)
        return rat_fail("rat_sparse_plan: radix sort failed");
    RAT_LAUNCH(head_flags_kernel, sp_blocks(n), SP_THREADS, 0, stream, v.keys_b, v.keys_a, n);      // keys_a is free now: flags
    tb = v.temp_bytes;
    if (rocprim::inclusive_scan(v.temp, tb, v.keys_a, v.seg_id, (size_t)n, rocprim::plus<uint32_t>(), s) != hipSuccess)
        return rat_fail("rat_sparse_plan: scan failed");
#endif
    RAT_LAUNCH(segment_starts_kernel, sp_blocks(n), SP_THREADS, 0, stream, v.keys_b, v.seg_id, invalid, n, v.seg_start, count);
    return rat_check_launch("rat_sparse_plan");
}

unsigned bits_for(uint64_t maxval) {
    unsigned b = 1;
    while (b < 32 && (maxval >> b) != 0) ++b;
    return b;
}

// ---- reduce -----------------------------------------------------------------------------------------------------
// SRC 0: token grid (+ DNN-branch rows on the target sample)   SRC 1: row list   SRC 2: one scalar per sample (width 1)
struct ReduceArgs {
    const uint32_t* keys;
    const uint32_t* vals;
    const uint32_t* seg_start;
    const int32_t* count;
    int64_t max_segments;
    const float* src;        // dgrid / gathered rows / dlogit
    const float* dflat;      // SRC 0 only (nullable)
    const int32_t* col2field;
    int T, L, S, F, d, target_only;
    int32_t* out_rows;       // nullable
    float* out_grads;        // nullable: [segment][d]
    float* dense_base;       // nullable: dense gradient block, row r at dense_base + r * d
};

template <int SRC>
__device__ __forceinline__ void entry_sources(const ReduceArgs& a, uint32_t pos, const float*& p0, const float*& p1) {
    p1 = nullptr;
    if (SRC == 1) {
        p0 = a.src + (int64_t)pos * a.d;
    } else {
        const int64_t r = pos / a.L;
        const int c = (int)(pos - r * a.L);
        if (SRC == 2) {
            p0 = a.src + r;                                        // dlogit[b]: the plan ran over the target rows only
        } else {
            const int64_t bt = a.target_only ? r * a.T : r;
            const int fi = a.col2field[c];
            p0 = a.src + ((bt * a.S) + 1 + fi) * a.d;
            if (a.dflat != nullptr && (bt % a.T) == 0) p1 = a.dflat + ((bt / a.T) * a.F + fi) * a.d;
        }
    }
}

// G lanes per segment (G = d / 4 rounded up to a power of two, <= 64), 16 bytes per lane
template <int SRC, int G>
__global__ void __launch_bounds__(SP_THREADS) reduce_vec_kernel(ReduceArgs a) {
    const int sub = threadIdx.x % G;
    const int64_t nseg = *a.count;
    const int c4 = a.d >> 2;
    for (int64_t s = (int64_t)blockIdx.x * (SP_THREADS / G) + threadIdx.x / G; s < nseg; s += (int64_t)gridDim.x * (SP_THREADS / G)) {
        const uint32_t e0 = a.seg_start[s], e1 = a.seg_start[s + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sub < c4) {
            for (uint32_t e = e0; e < e1; ++e) {
                const float* p0;
                const float* p1;
                entry_sources<SRC>(a, a.vals[e], p0, p1);
                float4 v = reinterpret_cast<const float4*>(p0)[sub];
                if (p1 != nullptr) {
                    const float4 w = reinterpret_cast<const float4*>(p1)[sub];
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            const uint32_t row = a.keys[e0];
            if (a.out_grads != nullptr) reinterpret_cast<float4*>(a.out_grads + s * a.d)[sub] = acc;
            if (a.dense_base != nullptr) reinterpret_cast<float4*>(a.dense_base + (int64_t)row * a.d)[sub] = acc;
            if (sub == 0 && a.out_rows != nullptr) a.out_rows[s] = (int32_t)row;
        }
    }
}

// any width: one lane per (segment, column)
template <int SRC>
__global__ void __launch_bounds__(SP_THREADS) reduce_scalar_kernel(ReduceArgs a) {
    const int64_t nseg = *a.count;
    const int64_t nitems = nseg * a.d;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < nitems; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = it / a.d;
        const int c = (int)(it - s * a.d);
        const uint32_t e0 = a.seg_start[s], e1 = a.seg_start[s + 1];
        float acc = 0.f;
        for (uint32_t e = e0; e < e1; ++e) {
            const float* p0;
            const float* p1;
            entry_sources<SRC>(a, a.vals[e], p0, p1);
            float v = p0[SRC == 2 ? 0 : c];
            if (p1 != nullptr) v += p1[c];
            acc += v;
        }
        const uint32_t row = a.keys[e0];
        if (a.out_grads != nullptr) a.out_grads[s * a.d + c] = acc;
        if (a.dense_base != nullptr) a.dense_base[(int64_t)row * a.d + c] = acc;
        if (c == 0 && a.out_rows != nullptr) a.out_rows[s] = (int32_t)row;
    }
}

template <int SRC>
int launch_reduce(const ReduceArgs& a, bool aligned, void* stream) {
    const int64_t segs = a.max_segments;
    if (a.d % 4 == 0 && aligned && SRC != 2) {
        const int c4 = a.d / 4;
        const int G = c4 <= 1 ? 1 : c4 <= 2 ? 2 : c4 <= 4 ? 4 : c4 <= 8 ? 8 : c4 <= 16 ? 16 : c4 <= 32 ? 32 : 64;
        RAT_REQUIRE(c4 <= 64, "row width above 256 floats");
        int64_t blocks = (segs + (SP_THREADS / G) - 1) / (SP_THREADS / G);
        blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
        switch (G) {
            case 1: RAT_LAUNCH((reduce_vec_kernel<SRC, 1>), (unsigned)blocks, SP_THREADS, 0, stream, a); break;
            case 2: RAT_LAUNCH((reduce_vec_kernel<SRC, 2>), (unsigned)blocks, SP_THREADS, 0, stream, a); break;
            case 4: RAT_LAUNCH((reduce_vec_kernel<SRC, 4>), (unsigned)blocks, SP_THREADS, 0, stream, a); break;
            case 8: RAT_LAUNCH((reduce_vec_kernel<SRC, 8>), (unsigned)blocks, SP_THREADS, 0, stream, a); break;
            case 16: RAT_LAUNCH((reduce_vec_kernel<SRC, 16>), (unsigned)blocks, SP_THREADS, 0, stream, a); break;
            case 32: RAT_LAUNCH((reduce_vec_kernel<SRC, 32>), (unsigned)blocks, SP_THREADS, 0, stream, a); break;
            default: RAT_LAUNCH((reduce_vec_kernel<SRC, 64>), (unsigned)blocks, SP_THREADS, 0, stream, a); break;
        }
    } else {
        int64_t blocks = (segs * a.d + SP_THREADS - 1) / SP_THREADS;
        blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
        RAT_LAUNCH((reduce_scalar_kernel<SRC>), (unsigned)blocks, SP_THREADS, 0, stream, a);
    }
    return rat_check_launch("rat_sparse_reduce");
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- optimizer on row lists ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SP_THREADS)
sumsq_rows_kernel(const float* __restrict__ g, const int32_t* __restrict__ count, int d, float* out) {
    RAT_DYN_SMEM(smem);
    float* scratch = reinterpret_cast<float*>(smem);
    const int64_t n = (int64_t)(*count) * d;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        acc = fmaf(v, v, acc);
    }
    acc = rat_group_sum<64>(acc);
    __syncthreads();
    if (rat_lane() == 0) scratch[rat_wave()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, scratch[0] + scratch[1] + scratch[2] + scratch[3]);
}

__global__ void __launch_bounds__(SP_THREADS)
adam_rows_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, const int32_t* __restrict__ rows,
                 const float* __restrict__ g, const int32_t* __restrict__ count, int d, const float* norm_sq, float max_norm,
                 float step_size, float beta1, float beta2, float eps, float inv_sqrt_bc2) {
    float coef = 1.0f;
    if (norm_sq != nullptr) {
        coef = max_norm / (sqrtf(*norm_sq) + 1e-6f);
        coef = coef < 1.0f ? coef : 1.0f;
    }
    const int64_t n = (int64_t)(*count) * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / d;
        const int64_t o = (int64_t)rows[s] * d + (i - s * d);
        const float gv = g[i] * coef;
        const float mv = beta1 * m[o] + (1.0f - beta1) * gv;
        const float vv = beta2 * v[o] + (1.0f - beta2) * gv * gv;
        m[o] = mv;
        v[o] = vv;
        w[o] -= step_size * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    }
}

// the same with the step size and bias correction read from the device (rat_adam_tick's hyper[0..1]: capturable in a hipGraph)
__global__ void __launch_bounds__(SP_THREADS)
adam_rows_dev_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, const int32_t* __restrict__ rows,
                     const float* __restrict__ g, const int32_t* __restrict__ count, int d, const float* norm_sq, float max_norm,
                     const float* __restrict__ hyper, float beta1, float beta2, float eps) {
    float coef = 1.0f;
    if (norm_sq != nullptr) {
        coef = max_norm / (sqrtf(*norm_sq) + 1e-6f);
        coef = coef < 1.0f ? coef : 1.0f;
    }
    const float step_size = hyper[0], inv_sqrt_bc2 = hyper[1];
    const int64_t n = (int64_t)(*count) * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / d;
        const int64_t o = (int64_t)rows[s] * d + (i - s * d);
        const float gv = g[i] * coef;
        const float mv = beta1 * m[o] + (1.0f - beta1) * gv;
        const float vv = beta2 * v[o] + (1.0f - beta2) * gv * gv;
        m[o] = mv;
        v[o] = vv;
        w[o] -= step_size * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    }
}

// (unique rows, gradient rows) -> the dense gradient block: plain stores (rows are unique; the block holds the caller's zeros)
__global__ void __launch_bounds__(SP_THREADS)
scatter_rows_kernel(float* __restrict__ dense, const int32_t* __restrict__ rows, const float* __restrict__ g,
                    const int32_t* __restrict__ count, int d) {
    const int64_t n = (int64_t)(*count) * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i / d;
        dense[(int64_t)rows[s] * d + (i - s * d)] = g[i];
    }
}

// the same for `lists` lists of capacity `cap` each ([lists][cap] rows, [lists][cap][d] gradient rows, counts[lists]): what the
// owner-partitioned exchange leaves on every rank — one reduced list per owner, disjoint row ranges
__global__ void __launch_bounds__(SP_THREADS)
scatter_rows_lists_kernel(float* __restrict__ dense, const int32_t* __restrict__ rows, const float* __restrict__ g,
                          const int32_t* __restrict__ counts, int64_t cap, int lists, int d) {
    for (int k = 0; k < lists; ++k) {
        const int64_t n = (int64_t)counts[k] * d;
        const int32_t* rk = rows + (int64_t)k * cap;
        const float* gk = g + (int64_t)k * cap * d;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t s = i / d;
            dense[(int64_t)rk[s] * d + (i - s * d)] = gk[i];
        }
    }
}


// ---- owner-partitioned exchange of the row lists (data parallelism, ABI v8) -----------------------------------------------------
// Rank k owns the rows [k per, (k + 1) per) of a table family.  The per-owner counts of a rank's list are a function of the batch's
// IDS alone (the plan), so they are produced — and exchanged, and read back by the host — at the START of the step, while the
// forward runs; when the gradients exist the host already knows the exact split sizes: no host stall in the step, exact traffic.

// counts[k] = unique rows of the plan inside owner k's range (the plan's segments are sorted by row: a binary search per range edge)
__global__ void __launch_bounds__(SP_THREADS)
owner_counts_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ seg_start, const int32_t* __restrict__ count,
                    int64_t per, int world, int32_t* __restrict__ counts) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k >= world) return;
    const int64_t nseg = *count;
    auto below = [&](int64_t edge) {                 // segments whose row is < edge
        int64_t lo = 0, hi = nseg;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)keys[seg_start[mid]] < edge) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int64_t a = k == 0 ? 0 : below((int64_t)k * per);
    const int64_t b = k == world - 1 ? nseg : below((int64_t)(k + 1) * per);
    counts[k] = (int32_t)(b - a);
}

// One chunk per peer: [rows A, padded to 4][gradient rows A: nA x d][rows B, padded to 4][values B, padded to 4] (32-bit words; d % 4
// == 0, so every section starts on a 16-byte boundary).  mat[s][f][k] = rows of family f that rank s holds for owner k.
struct OwnerXfer {
    const int32_t* mat;      // [world][2][world], on the device
    int world, rank, d;
    int32_t* rows_a; float* grads_a;     // PACK: this rank's lists (sorted by row = owner after owner)   UNPACK: the received pairs,
    int32_t* rows_b; float* vals_b;      //       sources in rank order
    float* wire;             // PACK: send buffer   UNPACK: receive buffer
    int32_t* totals;         // UNPACK: {pairs received A, pairs received B}
    const float* extra_src;  // UNPACK (nullable): n_extra floats copied to extra_dst (the label table's partial gradient -> the
    float* extra_dst;        //                    header of the list this rank will all-gather)
    int n_extra;
};

__device__ __forceinline__ int64_t owner_pad4(int64_t n) { return (n + 3) & ~(int64_t)3; }
__device__ __forceinline__ int64_t owner_chunk(int64_t na, int64_t nb, int d) { return owner_pad4(na) + na * d + 2 * owner_pad4(nb); }

template <bool PACK>
__global__ void __launch_bounds__(SP_THREADS) owner_xfer_kernel(OwnerXfer a, int blocks_per_peer) {
    const int peer = (int)(blockIdx.x / blocks_per_peer);
    const int sub = (int)(blockIdx.x % blocks_per_peer);
    const int W = a.world;
    auto cnt = [&](int j, int f) { return (int64_t)(PACK ? a.mat[((int64_t)a.rank * 2 + f) * W + j] : a.mat[((int64_t)j * 2 + f) * W + a.rank]); };
    int64_t wire_off = 0, la = 0, lb = 0;
    for (int j = 0; j < peer; ++j) {
        const int64_t na = cnt(j, 0), nb = cnt(j, 1);
        wire_off += owner_chunk(na, nb, a.d);
        la += na;
        lb += nb;
    }
    const int64_t na = cnt(peer, 0), nb = cnt(peer, 1);
    float* w = a.wire + wire_off;
    int32_t* w_rows_a = reinterpret_cast<int32_t*>(w);
    float* w_grads_a = w + owner_pad4(na);
    int32_t* w_rows_b = reinterpret_cast<int32_t*>(w_grads_a + na * a.d);
    float* w_vals_b = reinterpret_cast<float*>(w_rows_b) + owner_pad4(nb);
    const int64_t t0 = (int64_t)sub * blockDim.x + threadIdx.x, stride = (int64_t)blocks_per_peer * blockDim.x;
    for (int64_t i = t0; i < na; i += stride) {
        if (PACK) w_rows_a[i] = a.rows_a[la + i]; else a.rows_a[la + i] = w_rows_a[i];
    }
    const int64_t n4 = na * (a.d >> 2);
    float4* g4 = reinterpret_cast<float4*>(a.grads_a + la * a.d);
    float4* w4 = reinterpret_cast<float4*>(w_grads_a);
    for (int64_t i = t0; i < n4; i += stride) {
        if (PACK) w4[i] = g4[i]; else g4[i] = w4[i];
    }
    for (int64_t i = t0; i < nb; i += stride) {
        if (PACK) {
            w_rows_b[i] = a.rows_b[lb + i];
            w_vals_b[i] = a.vals_b[lb + i];
        } else {
            a.rows_b[lb + i] = w_rows_b[i];
            a.vals_b[lb + i] = w_vals_b[i];
        }
    }
    if (!PACK && blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            int64_t ta = 0, tb = 0;
            for (int j = 0; j < W; ++j) { ta += cnt(j, 0); tb += cnt(j, 1); }
            a.totals[0] = (int32_t)ta;
            a.totals[1] = (int32_t)tb;
        }
        for (int i = threadIdx.x; i < a.n_extra; i += blockDim.x) a.extra_dst[i] = a.extra_src[i];
    }
}

// every rank's all-gathered list — [count A, count B, -, -][n_extra floats, padded to 4][rows A: cap_a][gradient rows A: cap_a x d]
// [rows B: cap_b][values B: cap_b], `stride` words apart — into the (zeroed) dense gradient blocks: plain stores (the owners' row
// ranges are disjoint); the n_extra floats (the label table's partial gradients) are summed over the ranks in rank order
__global__ void __launch_bounds__(SP_THREADS)
owner_scatter_kernel(float* __restrict__ dense_a, float* __restrict__ dense_b, float* __restrict__ extra_out,
                     const float* __restrict__ lists, int64_t stride, int world, int64_t cap_a, int64_t cap_b, int d, int n_extra) {
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    const int64_t xpad = owner_pad4(n_extra);
    for (int k = 0; k < world; ++k) {
        const float* base = lists + (int64_t)k * stride;
        const int32_t* hdr = reinterpret_cast<const int32_t*>(base);
        const int32_t* rows_a = reinterpret_cast<const int32_t*>(base + 4 + xpad);
        const float* grads_a = base + 4 + xpad + cap_a;
        const int32_t* rows_b = reinterpret_cast<const int32_t*>(grads_a + cap_a * d);
        const float* vals_b = reinterpret_cast<const float*>(rows_b) + cap_b;
        const int64_t na = (int64_t)hdr[0] * (d >> 2);
        const int q = d >> 2;
        for (int64_t i = t0; i < na; i += step) {
            const int64_t s_ = i / q;
            reinterpret_cast<float4*>(dense_a + (int64_t)rows_a[s_] * d)[i - s_ * q] = reinterpret_cast<const float4*>(grads_a)[i];
        }
        const int64_t nb = hdr[1];
        for (int64_t i = t0; i < nb; i += step) dense_b[rows_b[i]] = vals_b[i];
    }
    for (int64_t i = t0; i < n_extra; i += step) {
        float acc = 0.f;
        for (int k = 0; k < world; ++k) acc += lists[(int64_t)k * stride + 4 + i];
        extra_out[i] = acc;
    }
}

}  // namespace

extern "C" size_t rat_sparse_workspace(int64_t n) {
    if (n < 1) n = 1;
    return 6 * align256((size_t)(n + 1) * sizeof(uint32_t)) + prim_temp_bytes(n);
}

extern "C" int rat_sparse_plan_ids(const int32_t* idx, const RatField* fields_dev, const int32_t* col2field_dev, int nfields,
                                   const float* flat_base, int width, int64_t total_rows, int B, int T, int L, int target_only,
                                   void* workspace, size_t workspace_bytes, int32_t* count_out, void* stream) {
    RAT_REQUIRE(idx && fields_dev && col2field_dev && flat_base && workspace && count_out, "null pointer");
    RAT_REQUIRE(B > 0 && T > 0 && L > 0 && nfields > 0 && width > 0 && total_rows > 0 && total_rows < 0x7fffffffLL, "bad dims");
    const int64_t nrows = target_only ? (int64_t)B : (int64_t)B * T;
    const int64_t n = nrows * L;
    RAT_REQUIRE(n < 0x7fffffffLL, "too many (sample, id) pairs for 32-bit positions");
    RAT_REQUIRE(workspace_bytes >= rat_sparse_workspace(n), "workspace too small");
    PlanView v = carve(workspace, n);
    v.temp_bytes = workspace_bytes - (size_t)(static_cast<char*>(v.temp) - static_cast<char*>(workspace));
    RAT_LAUNCH(keys_from_ids_kernel, sp_blocks(n), SP_THREADS, 0, stream, idx, fields_dev, col2field_dev, flat_base, width,
               (uint32_t)total_rows, nrows, T, L, target_only, v.keys_a, v.vals_a);
    return build_segments(v, n, (uint32_t)total_rows, bits_for((uint64_t)total_rows), count_out, stream);
}

extern "C" int rat_sparse_plan_rows(const int32_t* rows, const int32_t* counts_dev, int64_t cap, int world, int64_t total_rows,
                                    void* workspace, size_t workspace_bytes, int32_t* count_out, void* stream) {
    RAT_REQUIRE(rows && counts_dev && workspace && count_out, "null pointer");
    RAT_REQUIRE(cap > 0 && world > 0 && total_rows > 0 && total_rows < 0x7fffffffLL && cap * world < 0x7fffffffLL, "bad dims");
    const int64_t n = cap * world;
    RAT_REQUIRE(workspace_bytes >= rat_sparse_workspace(n), "workspace too small");
    PlanView v = carve(workspace, n);
    v.temp_bytes = workspace_bytes - (size_t)(static_cast<char*>(v.temp) - static_cast<char*>(workspace));
    RAT_LAUNCH(keys_from_rows_kernel, sp_blocks(n), SP_THREADS, 0, stream, rows, counts_dev, cap, world, (uint32_t)total_rows,
               v.keys_a, v.vals_a);
    return build_segments(v, n, (uint32_t)total_rows, bits_for((uint64_t)total_rows), count_out, stream);
}

extern "C" int rat_sparse_reduce_grid(const void* workspace, const int32_t* count_dev, const float* dgrid, const float* dflat,
                                      const int32_t* col2field_dev, int B, int T, int L, int nfields, int d, int target_only,
                                      int32_t* out_rows, float* out_grads, float* dense_base, void* stream) {
    RAT_REQUIRE(workspace && count_dev && dgrid && col2field_dev && (out_grads || dense_base), "null pointer");
    const int64_t n = (target_only ? (int64_t)B : (int64_t)B * T) * L;
    PlanView v = carve(const_cast<void*>(workspace), n);
    ReduceArgs a{};
    a.keys = v.keys_b; a.vals = v.vals_b; a.seg_start = v.seg_start; a.count = count_dev; a.max_segments = n;
    a.src = dgrid; a.dflat = dflat; a.col2field = col2field_dev;
    a.T = T; a.L = L; a.S = nfields + 1; a.F = nfields; a.d = d; a.target_only = target_only;
    a.out_rows = out_rows; a.out_grads = out_grads; a.dense_base = dense_base;
    return launch_reduce<0>(a, al16(dgrid) && al16(dflat) && al16(out_grads) && al16(dense_base), stream);
}

extern "C" int rat_sparse_reduce_rows(const void* workspace, const int32_t* count_dev, const float* src_rows, int64_t cap, int world,
                                      int d, int32_t* out_rows, float* out_grads, void* stream) {
    RAT_REQUIRE(workspace && count_dev && src_rows && out_rows && out_grads && cap > 0 && world > 0 && d > 0, "bad args");
    const int64_t n = cap * world;
    PlanView v = carve(const_cast<void*>(workspace), n);
    ReduceArgs a{};
    a.keys = v.keys_b; a.vals = v.vals_b; a.seg_start = v.seg_start; a.count = count_dev; a.max_segments = n;
    a.src = src_rows; a.d = d; a.L = 1; a.T = 1;
    a.out_rows = out_rows; a.out_grads = out_grads;
    return launch_reduce<1>(a, al16(src_rows) && al16(out_grads), stream);
}

extern "C" int rat_sparse_reduce_scalar(const void* workspace, const int32_t* count_dev, const float* per_sample, int B, int L,
                                        int32_t* out_rows, float* out_vals, float* dense_base, void* stream) {
    RAT_REQUIRE(workspace && count_dev && per_sample && (out_vals || dense_base) && B > 0 && L > 0, "bad args");
    const int64_t n = (int64_t)B * L;
    PlanView v = carve(const_cast<void*>(workspace), n);
    ReduceArgs a{};
    a.keys = v.keys_b; a.vals = v.vals_b; a.seg_start = v.seg_start; a.count = count_dev; a.max_segments = n;
    a.src = per_sample; a.d = 1; a.L = L; a.T = 1; a.target_only = 1;
    a.out_rows = out_rows; a.out_grads = out_vals; a.dense_base = dense_base;
    return launch_reduce<2>(a, false, stream);
}

extern "C" int rat_sumsq_rows(const float* grads, const int32_t* count_dev, int64_t max_rows, int d, float* norm_sq_out, void* stream) {
    RAT_REQUIRE(grads && count_dev && norm_sq_out && d > 0 && max_rows > 0, "bad args");
    int64_t blocks = (max_rows * d + SP_THREADS * 4 - 1) / (SP_THREADS * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    RAT_LAUNCH(sumsq_rows_kernel, (unsigned)blocks, SP_THREADS, 16 * sizeof(float), stream, grads, count_dev, d, norm_sq_out);
    return rat_check_launch("rat_sumsq_rows");
}

extern "C" int rat_adam_rows(float* w_base, float* m_base, float* v_base, const int32_t* rows, const float* grads,
                             const int32_t* count_dev, int64_t max_rows, int d, const float* norm_sq, float max_norm, float lr,
                             float beta1, float beta2, float eps, int step, void* stream) {
    RAT_REQUIRE(w_base && m_base && v_base && rows && grads && count_dev && d > 0 && max_rows > 0 && step >= 1, "bad args");
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    int64_t blocks = (max_rows * d + SP_THREADS * 4 - 1) / (SP_THREADS * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    RAT_LAUNCH(adam_rows_kernel, (unsigned)blocks, SP_THREADS, 0, stream, w_base, m_base, v_base, rows, grads, count_dev, d, norm_sq,
               max_norm, (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)));
    return rat_check_launch("rat_adam_rows");
}

extern "C" int rat_adam_rows_dev(float* w_base, float* m_base, float* v_base, const int32_t* rows, const float* grads,
                                 const int32_t* count_dev, int64_t max_rows, int d, const float* norm_sq, float max_norm,
                                 const float* hyper_dev, float beta1, float beta2, float eps, void* stream) {
    RAT_REQUIRE(w_base && m_base && v_base && rows && grads && count_dev && hyper_dev && d > 0 && max_rows > 0, "bad args");
    int64_t blocks = (max_rows * d + SP_THREADS * 4 - 1) / (SP_THREADS * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    RAT_LAUNCH(adam_rows_dev_kernel, (unsigned)blocks, SP_THREADS, 0, stream, w_base, m_base, v_base, rows, grads, count_dev, d, norm_sq,
               max_norm, hyper_dev, beta1, beta2, eps);
    return rat_check_launch("rat_adam_rows_dev");
}

extern "C" int rat_scatter_rows(float* dense_base, const int32_t* rows, const float* grads, const int32_t* count_dev, int64_t max_rows,
                                int d, void* stream) {
    RAT_REQUIRE(dense_base && rows && grads && count_dev && d > 0 && max_rows > 0, "bad args");
    int64_t blocks = (max_rows * d + SP_THREADS * 4 - 1) / (SP_THREADS * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    RAT_LAUNCH(scatter_rows_kernel, (unsigned)blocks, SP_THREADS, 0, stream, dense_base, rows, grads, count_dev, d);
    return rat_check_launch("rat_scatter_rows");
}

extern "C" int rat_scatter_rows_lists(float* dense_base, const int32_t* rows, const float* grads, const int32_t* counts_dev, int64_t cap,
                                      int lists, int d, void* stream) {
    RAT_REQUIRE(dense_base && rows && grads && counts_dev && d > 0 && cap > 0 && lists > 0, "bad args");
    int64_t blocks = (cap * d + SP_THREADS * 4 - 1) / (SP_THREADS * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    RAT_LAUNCH(scatter_rows_lists_kernel, (unsigned)blocks, SP_THREADS, 0, stream, dense_base, rows, grads, counts_dev, cap, lists, d);
    return rat_check_launch("rat_scatter_rows_lists");
}

extern "C" int rat_owner_counts(const void* workspace, const int32_t* count_dev, int64_t n, int64_t rows_per_owner, int world,
                                int32_t* counts_out, void* stream) {
    RAT_REQUIRE(workspace && count_dev && counts_out && n > 0 && rows_per_owner > 0 && world > 0 && world <= 4096, "bad args");
    PlanView v = carve(const_cast<void*>(workspace), n);
    RAT_LAUNCH(owner_counts_kernel, (unsigned)((world + SP_THREADS - 1) / SP_THREADS), SP_THREADS, 0, stream, v.keys_b, v.seg_start,
               count_dev, rows_per_owner, world, counts_out);
    return rat_check_launch("rat_owner_counts");
}

static int owner_blocks_per_peer(int64_t max_pairs, int d, int world) {
    int64_t b = (max_pairs * (d / 4 + 1) + SP_THREADS * 8 - 1) / (SP_THREADS * 8);
    const int64_t limit = 2048 / world > 1 ? 2048 / world : 1;
    return (int)(b < 1 ? 1 : (b > limit ? limit : b));
}

extern "C" int rat_owner_pack(const int32_t* mat_dev, int world, int rank, int d, const int32_t* rows_a, const float* grads_a,
                              const int32_t* rows_b, const float* vals_b, int64_t max_pairs, float* wire, void* stream) {
    RAT_REQUIRE(mat_dev && rows_a && grads_a && wire && world > 0 && rank >= 0 && rank < world && d > 0 && d % 4 == 0, "bad args");
    RAT_REQUIRE(al16(grads_a) && al16(wire), "gradient rows and the wire buffer must be 16-byte aligned");
    OwnerXfer a{};
    a.mat = mat_dev; a.world = world; a.rank = rank; a.d = d;
    a.rows_a = const_cast<int32_t*>(rows_a); a.grads_a = const_cast<float*>(grads_a);
    a.rows_b = const_cast<int32_t*>(rows_b); a.vals_b = const_cast<float*>(vals_b);
    a.wire = wire;
    const int bpp = owner_blocks_per_peer(max_pairs, d, world);
    RAT_LAUNCH((owner_xfer_kernel<true>), (unsigned)(bpp * world), SP_THREADS, 0, stream, a, bpp);
    return rat_check_launch("rat_owner_pack");
}

extern "C" int rat_owner_unpack(const int32_t* mat_dev, int world, int rank, int d, const float* wire, int64_t max_pairs,
                                int32_t* rows_a, float* grads_a, int32_t* rows_b, float* vals_b, int32_t* totals,
                                const float* extra_src, float* extra_dst, int n_extra, void* stream) {
    RAT_REQUIRE(mat_dev && rows_a && grads_a && wire && totals && world > 0 && rank >= 0 && rank < world && d > 0 && d % 4 == 0, "bad args");
    RAT_REQUIRE(al16(grads_a) && al16(wire), "gradient rows and the wire buffer must be 16-byte aligned");
    RAT_REQUIRE(n_extra == 0 || (extra_src && extra_dst), "null pointer");
    OwnerXfer a{};
    a.mat = mat_dev; a.world = world; a.rank = rank; a.d = d;
    a.rows_a = rows_a; a.grads_a = grads_a; a.rows_b = rows_b; a.vals_b = vals_b;
    a.wire = const_cast<float*>(wire); a.totals = totals;
    a.extra_src = extra_src; a.extra_dst = extra_dst; a.n_extra = n_extra;
    const int bpp = owner_blocks_per_peer(max_pairs, d, world);
    RAT_LAUNCH((owner_xfer_kernel<false>), (unsigned)(bpp * world), SP_THREADS, 0, stream, a, bpp);
    return rat_check_launch("rat_owner_unpack");
}

extern "C" int rat_owner_scatter(float* dense_a, float* dense_b, float* extra_out, const float* lists, int64_t stride, int world,
                                 int64_t cap_a, int64_t cap_b, int d, int n_extra, void* stream) {
    RAT_REQUIRE(dense_a && lists && world > 0 && stride > 0 && cap_a >= 0 && cap_b >= 0 && d > 0 && d % 4 == 0, "bad args");
    RAT_REQUIRE((cap_b == 0 || dense_b) && (n_extra == 0 || extra_out), "null pointer");
    RAT_REQUIRE(cap_a % 4 == 0 && cap_b % 4 == 0 && stride % 4 == 0 && al16(lists) && al16(dense_a), "sections must be 16-byte aligned");
    int64_t blocks = (cap_a * (d / 4) + SP_THREADS * 4 - 1) / (SP_THREADS * 4);
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    RAT_LAUNCH(owner_scatter_kernel, (unsigned)blocks, SP_THREADS, 0, stream, dense_a, dense_b, extra_out, lists, stride, world, cap_a,
               cap_b, d, n_extra);
    return rat_check_launch("rat_owner_scatter");
}
