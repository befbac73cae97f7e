// Lane-accurate HOST emulation of the small HIP subset the rat kernels use.  TEST INFRASTRUCTURE ONLY.
//
// tests/emu/build_emu.py compiles www24-rat_amd/csrc/*.hip with g++ -DRAT_EMU against this header into
// tests/emu/librat_emu.so so that index arithmetic, LDS layouts, barriers and the MFMA lane maps can be
// debugged in the build container, which has no GPU.  One OS thread per GPU thread, blocks run one after
// another, __syncthreads()/wave collectives are real barriers.  The product (rat_amd) never loads this.
//
// MFMA lane maps follow /opt/skills/guides/cdna_hip_programming.md §3 (v_mfma_f32_16x16x4_f32):
//   A: lane l holds A[i = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][j = l & 15];
//   C/D: col = l & 15, row = (l >> 4) * 4 + reg.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static

namespace emu {
// Sense-reversing barrier that YIELDS instead of sleeping on a futex: hundreds of OS threads share a handful of cores, a waiter gives
// its time slice to a thread that still has to arrive, and the last arriver releases everybody without a system call per waiter
// (measured on the 8-core build container: 21 s -> 16 s for a bf16x3 attention test; yield-then-futex was slower than either).
struct YieldBarrier {
    explicit YieldBarrier(unsigned n) : n_(n) {}
    void arrive_and_wait() {
        const unsigned gen = gen_.load(std::memory_order_acquire);
        if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            count_.store(0, std::memory_order_relaxed);
            gen_.store(gen + 1, std::memory_order_release);
            return;
        }
        while (gen_.load(std::memory_order_acquire) == gen) std::this_thread::yield();
    }
    const unsigned n_;
    std::atomic<unsigned> count_{0}, gen_{0};
};
struct Block {
    unsigned nthreads;
    std::unique_ptr<YieldBarrier> block_barrier;
    std::vector<std::unique_ptr<YieldBarrier>> wave_barrier;
    std::vector<float> xa, xb;          // per-wave 64-entry exchange buffers
    std::vector<unsigned long long> xu;
    std::vector<unsigned> xw;           // per-wave 64 x 8-dword exchange buffer (bf16 MFMA fragments, transposed LDS reads)
    std::vector<char> smem;
};
extern thread_local Block* g_block;
extern thread_local dim3 g_tid, g_bid, g_bdim, g_gdim;
extern thread_local unsigned g_coll;     // wave collectives this thread has taken part in
inline int lane() { return g_tid.x & 63; }
inline int wave() { return g_tid.x >> 6; }
// The exchange buffers of a wave are DOUBLE-buffered by the parity of the collective's ordinal: a collective is write -> wave barrier ->
// read, and a lane can only overwrite a slot two collectives later, i.e. after the barrier of the collective in between, which every
// lane reaches only after its reads — so no second barrier is needed (all lanes of a wave execute the same sequence of collectives,
// as on the GPU).
inline int slot() { return (int)(wave() * 2 + (g_coll++ & 1u)); }
inline void wave_sync() { g_block->wave_barrier[wave()]->arrive_and_wait(); }
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::g_tid)
#define blockIdx (emu::g_bid)
#define blockDim (emu::g_bdim)
#define gridDim (emu::g_gdim)

static inline void __syncthreads() { emu::g_block->block_barrier->arrive_and_wait(); }

template <class T>
static inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    float* buf = &emu::g_block->xa[emu::slot() * 64];
    std::memcpy(&buf[emu::lane()], &v, 4);
    emu::wave_sync();
    T r;
    std::memcpy(&r, &buf[src_lane & 63], 4);
    return r;
}
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu_exchange(v, emu::lane() ^ mask); }
template <class T> static inline T __shfl(T v, int src, int = 64) { return emu_exchange(v, src); }
template <class T> static inline T __shfl_down(T v, int delta, int = 64) {
    int s = emu::lane() + delta;
    return emu_exchange(v, s < 64 ? s : emu::lane());
}

static inline float atomicAdd(float* p, float v) {
    std::atomic_ref<float> r(*p);
    float old = r.load();
    while (!r.compare_exchange_weak(old, old + v)) {}
    return old;
}
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline int atomicAdd(int* p, int v) { return std::atomic_ref<int>(*p).fetch_add(v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return std::atomic_ref<unsigned>(*p).fetch_add(v); }
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    std::atomic_ref<unsigned> r(*p);
    unsigned old = r.load();
    while (old < v && !r.compare_exchange_weak(old, v)) {}
    return old;
}

typedef float f32x4 __attribute__((vector_size(16)));

static inline f32x4 emu_mfma_f32_16x16x4f32(float a, float b, f32x4 c) {
    const int sl = emu::slot();
    float* wa = &emu::g_block->xa[sl * 64];
    float* wb = &emu::g_block->xb[sl * 64];
    const int l = emu::lane();
    wa[l] = a;
    wb[l] = b;
    emu::wave_sync();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float s = c[r];
        for (int k = 0; k < 4; ++k) s = std::fmaf(wa[k * 16 + row], wb[k * 16 + col], s);
        c[r] = s;
    }
    return c;
}

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks (block = lane / 4), K = 1:
//   D[reg r][lane l] = C[r][l] + A(lane 4*(l/4) + r) * B(lane l)      (lane map measured on gfx950: tools/probes/mfma4x4_probe.hip)
static inline f32x4 emu_mfma_f32_4x4x1f32(float a, float b, f32x4 c) {
    float* wa = &emu::g_block->xa[emu::slot() * 64];
    const int l = emu::lane();
    wa[l] = a;
    emu::wave_sync();
    for (int r = 0; r < 4; ++r) c[r] = std::fmaf(wa[(l & ~3) + r], b, c[r]);
    return c;
}

// ---- bf16 MFMA and the transposed LDS read of gfx950 (cdna_hip_programming.md §3 / T10) -------------------------------
typedef short bf16x8 __attribute__((vector_size(16)));
typedef short s16x4 __attribute__((vector_size(8)));

static inline float emu_bf16_to_f32(short v) {
    const unsigned u = ((unsigned)(unsigned short)v) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// v_mfma_f32_16x16x32_bf16: A: lane l holds A[row l & 15][k = 8 (l >> 4) + j]; B: lane l holds B[k = 8 (l >> 4) + j][col l & 15];
// C/D: col = l & 15, row = 4 (l >> 4) + reg.  bf16 x bf16 products are exact in fp32; they are summed in k order in fp32.
static inline f32x4 emu_mfma_f32_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    unsigned* w = &emu::g_block->xw[emu::slot() * 64 * 8];
    const int l = emu::lane();
    std::memcpy(&w[l * 8], &a, 16);
    std::memcpy(&w[l * 8 + 4], &b, 16);
    emu::wave_sync();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float s = c[r];
        for (int g = 0; g < 4; ++g) {
            short av[8], bv[8];
            std::memcpy(av, &w[(g * 16 + row) * 8], 16);
            std::memcpy(bv, &w[(g * 16 + col) * 8 + 4], 16);
            for (int j = 0; j < 8; ++j) s += emu_bf16_to_f32(av[j]) * emu_bf16_to_f32(bv[j]);
        }
        c[r] = s;
    }
    return c;
}

// ds_read_b64_tr_b16: within every group of 16 consecutive lanes, lane 4q + p supplies the address of 4 consecutive 16-bit
// elements (row q of a 4 x 16 block, columns 4p .. 4p+3); lane i of the group receives column i: element q = row q.
static inline s16x4 emu_lds_tr16(const unsigned short* p) {
    unsigned* w = &emu::g_block->xw[emu::slot() * 64 * 8];
    const int l = emu::lane();
    std::memcpy(&w[l * 8], &p, sizeof(p));
    emu::wave_sync();
    const int base = l & ~15, i = l & 15;
    s16x4 r;
    for (int q = 0; q < 4; ++q) {
        const unsigned short* src;
        std::memcpy(&src, &w[(base + 4 * q + (i >> 2)) * 8], sizeof(src));
        r[q] = (short)src[i & 3];
    }
    return r;
}

#define RAT_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), (smem), [&]() { kernel(__VA_ARGS__); })
#define RAT_DYN_SMEM(name) char* name = emu::g_block->smem.data()
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
