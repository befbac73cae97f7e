// rat_common.hip — error reporting, ABI version, slab reduction shared by the backward kernels.
#include "rat_device.h"
#include <stdlib.h>
#include <vector>
#include "../../include/rat_hip.h"

static thread_local std::string g_last_error;

const char* rat_set_error(const std::string& msg) {
    g_last_error = msg;
    return g_last_error.c_str();
}
int rat_fail(const std::string& msg) {
    rat_set_error(msg);
    return -1;
}
int rat_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rat_fail(std::string(what) + ": " + hipGetErrorString(e));
    return 0;
}

// Diagnostic knobs.  The product never sets them, and no launch reads the environment: the table is filled ONCE, when the library is
// loaded, from the RAT_* variables below (what the A/B scripts under tools/ set before they start a process), and
// rat_debug_set_knob changes an entry at run time (what the tests do around single calls).
//   max_blocks (RAT_MAX_BLOCKS=<n>)            caps the grid of the persistent encoder kernels so that small test problems make every
//                                              work-group loop over SEVERAL chunks (persistent accumulators, double-buffered row maps)
//   attn_bwd_ph (RAT_ATTN_BWD_PH=0)            0: pass 2 of attn_bwd3_kernel recomputes P instead of reading pass 1's copy (L <= 12)
//   attn_fwd_core_mfma (RAT_ATTN_FWD_CORE=mfma|mfma32|valu) 1: attn_fwd3m_kernel (bf16x3 QK^T / PV; correct, measured slower); 2 / 3: force the
//                                              exact-fp32 matrix-pipe core of attn_fwd3_kernel on (L <= 32) / off (0: by length, 28 ... 32 tokens)
//   attn_bwd_core_mfma (RAT_ATTN_BWD_CORE=mfma|valu) 1 / 0: force the matrix-pipe / VALU backward core whatever L says (-1: by L)
//   ffn_bwd_t3 (RAT_FFN_BWD=t3)                1: round 3's feed-forward backward kernel
//   sgemm_split_target (RAT_SGEMM_SPLIT_TARGET=<n>)  work-groups the split-K rule of the head's GEMMs aims for (0: built-in)
static int g_knobs[RAT_KNOB_COUNT];
static const char* const g_knob_names[RAT_KNOB_COUNT] = {"max_blocks", "attn_bwd_ph", "attn_fwd_core_mfma", "attn_bwd_core_mfma", "ffn_bwd_t3",
                                                         "sgemm_split_target"};
static const bool g_knobs_loaded = [] {
    auto env = [](const char* n) { const char* e = getenv(n); return e ? std::string(e) : std::string(); };
    const std::string mb = env("RAT_MAX_BLOCKS"), ph = env("RAT_ATTN_BWD_PH"), fc = env("RAT_ATTN_FWD_CORE"), bc = env("RAT_ATTN_BWD_CORE"),
                      ff = env("RAT_FFN_BWD"), st = env("RAT_SGEMM_SPLIT_TARGET");
    g_knobs[RAT_KNOB_MAX_BLOCKS] = mb.empty() ? 0 : atoi(mb.c_str());
    g_knobs[RAT_KNOB_ATTN_BWD_PH] = (ph.empty() || ph[0] != '0') ? 1 : 0;
    g_knobs[RAT_KNOB_ATTN_FWD_CORE_MFMA] = fc == "mfma" ? 1 : (fc == "mfma32" ? 2 : (fc == "valu" ? 3 : 0));
    g_knobs[RAT_KNOB_ATTN_BWD_CORE_MFMA] = bc == "mfma" ? 1 : (bc == "valu" ? 0 : -1);
    g_knobs[RAT_KNOB_FFN_BWD_T3] = ff.rfind("t3", 0) == 0 ? 1 : 0;
    g_knobs[RAT_KNOB_SGEMM_SPLIT_TARGET] = st.empty() ? 0 : (atoi(st.c_str()) > 0 ? atoi(st.c_str()) : 0);
    return true;
}();
int rat_knob(int which) { return g_knobs[which]; }
// diagnostic hook (not part of include/rat_hip.h): set knob `name`, returns its previous value; unknown name: INT_MIN
extern "C" int rat_debug_set_knob(const char* name, int value) {
    for (int i = 0; i < RAT_KNOB_COUNT; ++i)
        if (name != nullptr && std::string(name) == g_knob_names[i]) {
            const int old = g_knobs[i];
            g_knobs[i] = value;
            return old;
        }
    return -2147483647 - 1;
}
int rat_max_blocks() {
    const int n = g_knobs[RAT_KNOB_MAX_BLOCKS];
    return n > 0 && n < 256 ? n : 256;
}

static unsigned long long* g_prof = nullptr;
unsigned long long* rat_prof_buffer() { return g_prof; }
// diagnostic hook: 64 x u64 device buffer receiving per-phase cycle sums from a -DRAT_PROF build (slots: 0 attn_fwd,
// 12 attn_bwd, 24 ffn_fwd, 36 ffn_bwd); a no-op for the product build, whose kernels contain no stamps.
extern "C" void rat_debug_set_prof(void* device_u64x64) { g_prof = static_cast<unsigned long long*>(device_u64x64); }

extern "C" int rat_version(void) { return RAT_ABI_VERSION; }
extern "C" const char* rat_last_error(void) { return g_last_error.c_str(); }

// out[p] = sum over slabs in a fixed order => bitwise reproducible for a fixed grid.  One launch covers all the
// parameter tensors of a backward kernel: 64 parameters x 4 slab-quarters per 256-thread block (coalesced across
// parameters), quarters combined through LDS in a fixed order.
struct RatReduceArgs {
    const float* slabs;
    int nslabs;
    int64_t stride;
    int nouts;
    float* out[8];
    int64_t off[8];
    int64_t size[8];
    int64_t first_block[9];     // block range of each output
};

__global__ void __launch_bounds__(256) rat_reduce_slabs_kernel(RatReduceArgs r) {
    __shared__ float part[256];
    int o = 0;
    while (o + 1 < r.nouts && (int64_t)blockIdx.x >= r.first_block[o + 1]) ++o;
    const int lane_p = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int64_t p = ((int64_t)blockIdx.x - r.first_block[o]) * 64 + lane_p;
    float s = 0.f;
    if (p < r.size[o]) {
        const int per = (r.nslabs + 3) / 4;
        const int w0 = quarter * per, w1 = (w0 + per < r.nslabs) ? w0 + per : r.nslabs;
        const float* src = r.slabs + r.off[o] + p;
        int w = w0;
        for (; w + 8 <= w1; w += 8) {                 // eight requests in flight, added in slab order (same sum as the plain loop)
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(int64_t)(w + k) * r.stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; w < w1; ++w) s += src[(int64_t)w * r.stride];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (quarter == 0 && p < r.size[o]) r.out[o][p] = ((part[lane_p] + part[64 + lane_p]) + part[128 + lane_p]) + part[192 + lane_p];
}

// ---- deferred slab reductions (ABI v7: rat_reduce_defer_begin / rat_reduce_defer_end).  Every backward kernel of an encoder layer ends
// with a slab reduction of its own — twelve launches of ~6 us per step at depth 4, a tenth of the step at BASELINE configs[0].  Nothing
// reads those gradients before the optimizer, so a caller that gives every layer its OWN slab workspace may record the reductions
// instead and run them all in one launch.  The record is per host thread; the sums and their order are those of the single launches.
struct RatReduceEntry {
    const float* src;           // slabs + offset of this output
    float* out;
    int64_t size, stride;
    int nslabs, first_block;
};
constexpr int RAT_REDUCE_BATCH = 80;                  // 80 x 40 bytes of kernel arguments
struct RatReduceTable {
    RatReduceEntry e[RAT_REDUCE_BATCH];
    int n, blocks;
};
static thread_local bool g_defer = false;
static thread_local std::vector<RatReduceEntry> g_deferred;

// `first_block < 0` marks a VECTOR entry (size, stride multiples of 4, 16-byte aligned pointers): a lane sums four neighbouring positions
// with 16-byte loads — 256 positions per block instead of 64, the same slab order and the same quarter tree per position, so the sums
// are bit-identical to the scalar form's (round 5: 236 MB of slabs per north-star step, 49 -> 47 us).
__global__ void __launch_bounds__(256) rat_reduce_slabs_batch_kernel(RatReduceTable t) {
    __shared__ float4 part4[256];
    float* part = reinterpret_cast<float*>(part4);
    int o = 0;
    auto first = [&](int i) { const int f = t.e[i].first_block; return f < 0 ? -f - 1 : f; };
    while (o + 1 < t.n && (int)blockIdx.x >= first(o + 1)) ++o;
    const RatReduceEntry& r = t.e[o];
    const bool vec = r.first_block < 0;
    const int lane_p = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int per = (r.nslabs + 3) / 4;
    const int w0 = quarter * per, w1 = (w0 + per < r.nslabs) ? w0 + per : r.nslabs;
    if (vec) {
        const int64_t p = (((int64_t)blockIdx.x - first(o)) * 64 + lane_p) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < r.size) {
            const float* src = r.src + p;
            int w = w0;
            for (; w + 4 <= w1; w += 4) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(src + (int64_t)(w + k) * r.stride);
#pragma unroll
                for (int k = 0; k < 4; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
            }
            for (; w < w1; ++w) {
                const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)w * r.stride);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        part4[threadIdx.x] = s;
        __syncthreads();
        if (quarter == 0 && p < r.size) {
            const float4 a = part4[lane_p], b = part4[64 + lane_p], c = part4[128 + lane_p], d = part4[192 + lane_p];
            *reinterpret_cast<float4*>(r.out + p) = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z,
                                                               ((a.w + b.w) + c.w) + d.w);
        }
        return;
    }
    const int64_t p = ((int64_t)blockIdx.x - first(o)) * 64 + lane_p;
    float s = 0.f;
    if (p < r.size) {                                  // the arithmetic of rat_reduce_slabs_kernel, summand for summand
        const float* src = r.src + p;
        int w = w0;
        for (; w + 8 <= w1; w += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(int64_t)(w + k) * r.stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; w < w1; ++w) s += src[(int64_t)w * r.stride];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (quarter == 0 && p < r.size) r.out[p] = ((part[lane_p] + part[64 + lane_p]) + part[128 + lane_p]) + part[192 + lane_p];
}

static int rat_flush_deferred(void* stream) {
    size_t i = 0;
    while (i < g_deferred.size()) {
        RatReduceTable t{};
        int blocks = 0;
        while (i < g_deferred.size() && t.n < RAT_REDUCE_BATCH) {
            RatReduceEntry e = g_deferred[i++];
            const bool vec = e.size % 4 == 0 && e.stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(e.src) | reinterpret_cast<uintptr_t>(e.out)) & 15) == 0;
            e.first_block = vec ? -blocks - 1 : blocks;
            blocks += (int)((e.size + (vec ? 255 : 63)) / (vec ? 256 : 64));
            t.e[t.n++] = e;
        }
        t.blocks = blocks;
        RAT_LAUNCH(rat_reduce_slabs_batch_kernel, (unsigned)blocks, 256, 0, stream, t);
        if (rat_check_launch("rat_reduce_slabs (batch)")) {
            g_deferred.clear();
            return -1;
        }
    }
    g_deferred.clear();
    return 0;
}

extern "C" int rat_reduce_defer_begin(void) {
    RAT_REQUIRE(!g_defer, "rat_reduce_defer_begin: already recording");
    g_defer = true;
    g_deferred.clear();
    return 0;
}

extern "C" int rat_reduce_defer_end(void* stream, int run) {
    RAT_REQUIRE(g_defer, "rat_reduce_defer_end without rat_reduce_defer_begin");
    g_defer = false;
    if (!run) {                                         // the caller is unwinding from an error: drop the record
        g_deferred.clear();
        return 0;
    }
    return rat_flush_deferred(stream);
}

int rat_launch_reduce_slabs(const float* slabs, int nslabs, int64_t stride, float* const* outs_host,
                            const int64_t* offsets, const int64_t* sizes, int nouts, void* stream) {
    if (g_defer) {
        for (int i = 0; i < nouts; ++i) {
            if (!outs_host[i] || sizes[i] <= 0) continue;
            g_deferred.push_back(RatReduceEntry{slabs + offsets[i], outs_host[i], sizes[i], stride, nslabs, 0});
        }
        return 0;
    }
    RatReduceArgs r{};
    r.slabs = slabs;
    r.nslabs = nslabs;
    r.stride = stride;
    int64_t blocks = 0;
    for (int i = 0; i < nouts; ++i) {
        if (!outs_host[i] || sizes[i] <= 0) continue;
        if (r.nouts >= 8) return rat_fail("rat_reduce_slabs: too many outputs");
        r.out[r.nouts] = outs_host[i];
        r.off[r.nouts] = offsets[i];
        r.size[r.nouts] = sizes[i];
        r.first_block[r.nouts] = blocks;
        blocks += (sizes[i] + 63) / 64;
        ++r.nouts;
    }
    if (r.nouts == 0) return 0;
    r.first_block[r.nouts] = blocks;
    RAT_LAUNCH(rat_reduce_slabs_kernel, (unsigned)blocks, 256, 0, stream, r);
    return rat_check_launch("rat_reduce_slabs");
}

// dst[c][r] = src[r][c]: transposed copies of the (small) weight matrices so that the backward kernels fetch every MFMA B
// fragment with one 16-byte load instead of four strided dwords
__global__ void __launch_bounds__(256) rat_transpose_kernel(const float* src, float* dst, int R, int C) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < R * C; e += gridDim.x * 256) {
        const int c = e / R, r = e - c * R;          // consecutive threads write consecutive dst elements
        dst[e] = src[(size_t)r * C + c];
    }
}
int rat_launch_transpose(const float* src, float* dst, int R, int C, void* stream) {
    const int blocks = (R * C + 255) / 256;
    RAT_LAUNCH(rat_transpose_kernel, (unsigned)(blocks < 256 ? blocks : 256), 256, 0, stream, src, dst, R, C);
    return rat_check_launch("rat_transpose");
}

// weights -> fragment-major bf16x3 planes (rat_device.h RatWPlanes): one thread per (n tile, K step, lane)
// `valid` (RatSplitJob.reserved): 0, or n_valid | k_valid << 16 — the matrix has only n_valid of the N rows / k_valid of the K columns the
// planes cover (0 = all of them); the rest of the planes is zeros.  That is how a narrower layer runs inside fixed-size tiles.
__device__ __forceinline__ void rat_split_bounds(int N, int K, int valid, int& nv, int& kv) {
    nv = (valid & 0xffff) ? (valid & 0xffff) : N;
    kv = (valid >> 16) ? (valid >> 16) : K;
}
// RatSplitJob.perm bits 8 ... 31 (ABI v9): the ROWS of `w` that the job reads are `blk` consecutive rows out of every `stride` — row
// index i (n without transpose, k with it) -> (i / blk) * stride + i % blk.  That is how a head group's Q | K | V rows (three blocks of 80
// rows, heads * dim_head rows apart) are read in place from to_qkv.weight even when a 32-wide k step straddles two blocks.
__device__ __forceinline__ int rat_split_row(int i, int perm) {
    const int blk = (perm >> 8) & 0xfff, stride = (perm >> 20) & 0xfff;
    return blk == 0 ? i : (i / blk) * stride + i % blk;
}
__global__ void __launch_bounds__(256) rat_split_weights_kernel(const float* __restrict__ w, int N, int K, int ld, int transpose, int perm,
                                                                rat_u4* __restrict__ out, int ntiles, int steps, int valid) {
    int nv, kv;
    rat_split_bounds(N, K, valid, nv, kv);
    const int total = ntiles * steps * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int lane = e & 63, fs = e >> 6, nt = fs / steps, s = fs - nt * steps;
        const int n = 16 * nt + (lane & 15), g = lane >> 4;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // perm: k slot j of lane group g <-> k = 32 s + 4 g + j (j < 4), 32 s + 16 + 4 g + (j - 4) (j >= 4) — the order in which
            // two stacked 16-row accumulator tiles present their rows as a B fragment (ffn.hip)
            const int k = 32 * s + ((perm & 1) ? (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : 8 * g + j);
            v[j] = (n < nv && k < kv) ? (transpose ? w[(size_t)rat_split_row(k, perm) * ld + n] : w[(size_t)rat_split_row(n, perm) * ld + k]) : 0.f;
        }
        rat_u4 h, m, l;
        rat_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), h, m, l);
        rat_u4* o = out + ((size_t)fs * 3) * 64 + lane;
        o[0] = h;
        o[64] = m;
        o[128] = l;
    }
}
// every job of a step in one launch: `per_job` consecutive work-groups per job (kernel-argument table), striding over its fragments
constexpr int RAT_SPLIT_BATCH = 48;
struct RatSplitTable {
    RatSplitJob job[RAT_SPLIT_BATCH];
};
__global__ void __launch_bounds__(256) rat_split_weights_batch_kernel(RatSplitTable t, int per_job) {
    const int bx = blockIdx.x % per_job;
    const RatSplitJob& jb = t.job[blockIdx.x / per_job];
    const float* __restrict__ w = jb.w;
    rat_u4* __restrict__ out = static_cast<rat_u4*>(jb.out);
    const int N = jb.N, K = jb.K, ld = jb.ld, transpose = jb.transpose, perm = jb.perm;
    int nv, kv;
    rat_split_bounds(N, K, jb.reserved, nv, kv);
    const int ntiles = (N + 15) / 16, steps = (K + 31) / 32, total = ntiles * steps * 64;
    for (int e = bx * 256 + threadIdx.x; e < total; e += per_job * 256) {
        const int lane = e & 63, fs = e >> 6, nt = fs / steps, s = fs - nt * steps;
        const int n = 16 * nt + (lane & 15), g = lane >> 4;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 32 * s + ((perm & 1) ? (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)) : 8 * g + j);      // as rat_split_weights_kernel
            v[j] = (n < nv && k < kv) ? (transpose ? w[(size_t)rat_split_row(k, perm) * ld + n] : w[(size_t)rat_split_row(n, perm) * ld + k]) : 0.f;
        }
        rat_u4 h, m, l;
        rat_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), h, m, l);
        rat_u4* o = out + ((size_t)fs * 3) * 64 + lane;
        o[0] = h;
        o[64] = m;
        o[128] = l;
    }
}
extern "C" int rat_split_weights_batch(const RatSplitJob* jobs_host, int njobs, void* stream) {
    RAT_REQUIRE(njobs >= 0 && (njobs == 0 || jobs_host != nullptr), "bad job list");
    for (int j0 = 0; j0 < njobs; j0 += RAT_SPLIT_BATCH) {
        const int nb = njobs - j0 < RAT_SPLIT_BATCH ? njobs - j0 : RAT_SPLIT_BATCH;
        RatSplitTable t{};
        int most = 1;
        for (int j = 0; j < nb; ++j) {
            const RatSplitJob& jb = jobs_host[j0 + j];
            RAT_REQUIRE(jb.w && jb.out && jb.N > 0 && jb.K > 0 && jb.ld > 0, "bad split job");
            RAT_REQUIRE((reinterpret_cast<uintptr_t>(jb.out) & 15) == 0, "split job output must be 16-byte aligned");
            t.job[j] = jb;
            const int frag = ((jb.N + 15) / 16) * ((jb.K + 31) / 32) * 64;
            most = most > (frag + 255) / 256 ? most : (frag + 255) / 256;
        }
        RAT_LAUNCH(rat_split_weights_batch_kernel, (unsigned)(most * nb), 256, 0, stream, t, most);
        if (rat_check_launch("rat_split_weights_batch")) return -1;
    }
    return 0;
}

int rat_launch_split_weights(const float* w, int N, int K, int ld, int transpose, void* out, void* stream, int perm, int valid) {
    const int ntiles = (N + 15) / 16, steps = (K + 31) / 32;
    const int blocks = (ntiles * steps * 64 + 255) / 256;
    RAT_LAUNCH(rat_split_weights_kernel, (unsigned)blocks, 256, 0, stream, w, N, K, ld, transpose, perm, static_cast<rat_u4*>(out), ntiles,
               steps, valid);
    return rat_check_launch("rat_split_weights");
}
