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
struct Block {
    unsigned nthreads;
    std::unique_ptr<std::barrier<>> block_barrier;
    std::vector<std::unique_ptr<std::barrier<>>> wave_barrier;
    std::vector<float> xa, xb;          // per-wave 64-entry exchange buffers
    std::vector<unsigned long long> xu;
    std::vector<char> smem;
};
extern thread_local Block* g_block;
extern thread_local dim3 g_tid, g_bid, g_bdim, g_gdim;
inline int lane() { return g_tid.x & 63; }
inline int wave() { return g_tid.x >> 6; }
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
    float* buf = &emu::g_block->xa[emu::wave() * 64];
    std::memcpy(&buf[emu::lane()], &v, 4);
    emu::wave_sync();
    T r;
    std::memcpy(&r, &buf[src_lane & 63], 4);
    emu::wave_sync();
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
static inline int atomicAdd(int* p, int v) { return std::atomic_ref<int>(*p).fetch_add(v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return std::atomic_ref<unsigned>(*p).fetch_add(v); }

typedef float f32x4 __attribute__((vector_size(16)));

static inline f32x4 emu_mfma_f32_16x16x4f32(float a, float b, f32x4 c) {
    float* wa = &emu::g_block->xa[emu::wave() * 64];
    float* wb = &emu::g_block->xb[emu::wave() * 64];
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
    emu::wave_sync();
    return c;
}

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks (block = lane / 4), K = 1:
//   D[reg r][lane l] = C[r][l] + A(lane 4*(l/4) + r) * B(lane l)      (lane map measured on gfx950: tools/probes/mfma4x4_probe.hip)
static inline f32x4 emu_mfma_f32_4x4x1f32(float a, float b, f32x4 c) {
    float* wa = &emu::g_block->xa[emu::wave() * 64];
    const int l = emu::lane();
    wa[l] = a;
    emu::wave_sync();
    for (int r = 0; r < 4; ++r) c[r] = std::fmaf(wa[(l & ~3) + r], b, c[r]);
    emu::wave_sync();
    return c;
}

#define RAT_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), (smem), [&]() { kernel(__VA_ARGS__); })
#define RAT_DYN_SMEM(name) char* name = emu::g_block->smem.data()
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
