// Diagnostic probe (not product): the 3-way bf16 split of fp32 GEMM operands on v_mfma_f32_16x16x32_bf16.
//   x = h + m + l exactly (h, m, l bf16), six of the nine cross products, fp32 accumulation inside the MFMA.
// Measures on the MI355X:
//   (a) the A / B / C lane maps of the instruction (exact small-integer data, asymmetric B);
//   (b) accuracy: C = A B^T for random fp32 data, K = 64 / 256 / 1024, against an fp64 host reference, for
//         - the exact-fp32 MFMA chain (v_mfma_f32_16x16x4_f32),
//         - the 6-product bf16 split (truncation split and round-to-nearest split),
//         - the 3-product 2-way split (to show why it is NOT enough);
//   (c) issue rate: 6 bf16 MFMAs vs 8 fp32 MFMAs per K = 32 step, register-fed, 1 and 2 waves per SIMD;
//   (d) co-execution: a wave of bf16 MFMAs beside a wave running the register split (VALU) on the same SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o bf16x3_probe bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((vector_size(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned u) { return __builtin_bit_cast(float, u); }

// 8 floats -> three packed bf16x8 planes.  RNE = 0: truncation (exact 8 + 8 + 8 bit split); RNE = 1: round-to-nearest pieces.
template <int RNE>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    unsigned hp[4], mp[4], lp[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        unsigned h0, h1;
        if (RNE) {
            const unsigned u0 = fbits(x0), u1 = fbits(x1);
            h0 = (u0 + 0x7fffu + ((u0 >> 16) & 1u)) & 0xffff0000u;
            h1 = (u1 + 0x7fffu + ((u1 >> 16) & 1u)) & 0xffff0000u;
        } else {
            h0 = fbits(x0) & 0xffff0000u;
            h1 = fbits(x1) & 0xffff0000u;
        }
        const float r0 = x0 - bitsf(h0), r1 = x1 - bitsf(h1);
        unsigned m0, m1;
        if (RNE) {
            const unsigned u0 = fbits(r0), u1 = fbits(r1);
            m0 = (u0 + 0x7fffu + ((u0 >> 16) & 1u)) & 0xffff0000u;
            m1 = (u1 + 0x7fffu + ((u1 >> 16) & 1u)) & 0xffff0000u;
        } else {
            m0 = fbits(r0) & 0xffff0000u;
            m1 = fbits(r1) & 0xffff0000u;
        }
        const float s0 = r0 - bitsf(m0), s1 = r1 - bitsf(m1);
        unsigned l0, l1;
        if (RNE) {
            const unsigned u0 = fbits(s0), u1 = fbits(s1);
            l0 = (u0 + 0x7fffu + ((u0 >> 16) & 1u)) & 0xffff0000u;
            l1 = (u1 + 0x7fffu + ((u1 >> 16) & 1u)) & 0xffff0000u;
        } else {
            l0 = fbits(s0) & 0xffff0000u;
            l1 = fbits(s1) & 0xffff0000u;
        }
        hp[p] = (h0 >> 16) | h1;
        mp[p] = (m0 >> 16) | m1;
        lp[p] = (l0 >> 16) | l1;
    }
    h = __builtin_bit_cast(bf16x8, (u32x4){hp[0], hp[1], hp[2], hp[3]});
    m = __builtin_bit_cast(bf16x8, (u32x4){mp[0], mp[1], mp[2], mp[3]});
    l = __builtin_bit_cast(bf16x8, (u32x4){lp[0], lp[1], lp[2], lp[3]});
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// one wave computes a 16x16 tile of C = A[16][K] * B[16][K]^T; mode 0 fp32 MFMA, 1 six-product trunc, 2 six-product RNE, 3 three-product
__global__ void gemm_tile(const float* A, const float* B, float* C, int K, int mode) {
    const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + g], B[r * K + k + g], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            float a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = A[r * K + k + 8 * g + j]; b[j] = B[r * K + k + 8 * g + j]; }
            bf16x8 ah, am, al, bh, bm, bl;
            if (mode == 2) { split8<1>(a, ah, am, al); split8<1>(b, bh, bm, bl); }
            else { split8<0>(a, ah, am, al); split8<0>(b, bh, bm, bl); }
            if (mode != 3) {                  // small terms first
                acc = MFMA_BF16(al, bh, acc);
                acc = MFMA_BF16(ah, bl, acc);
                acc = MFMA_BF16(am, bm, acc);
            }
            acc = MFMA_BF16(am, bh, acc);
            acc = MFMA_BF16(ah, bm, acc);
            acc = MFMA_BF16(ah, bh, acc);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) C[(4 * g + q) * 16 + r] = acc[q];       // C/D: col = lane & 15, row = 4 * (lane >> 4) + reg
}

// (c)/(d): throughput.  mode 0: 8 fp32 MFMAs per step; 1: 6 bf16 MFMAs per step; 2: split8 x2 only (VALU); 3: waves < 4 MFMA (bf16), waves >= 4 VALU split
__global__ void __launch_bounds__(512) rate(float* out, int iters, int mode, int nwaves_active) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (w >= nwaves_active) return;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = 0.37f * (l + j) - 3.f; b[j] = 0.11f * (l * j) + 0.5f; }
    bf16x8 ah, am, al, bh, bm, bl;
    split8<0>(a, ah, am, al);
    split8<0>(b, bh, bm, bl);
    float sink = 0.f;
    const bool do_mfma32 = mode == 0, do_bf = mode == 1 || (mode == 3 && w < 4), do_split = mode == 2 || (mode == 3 && w >= 4);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma32) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[t], 0, 0, 0);
        }
        if (do_bf) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = MFMA_BF16(al, bh, acc[t]);
                acc[t] = MFMA_BF16(ah, bl, acc[t]);
                acc[t] = MFMA_BF16(am, bm, acc[t]);
                acc[t] = MFMA_BF16(am, bh, acc[t]);
                acc[t] = MFMA_BF16(ah, bm, acc[t]);
                acc[t] = MFMA_BF16(ah, bh, acc[t]);
            }
        }
        if (do_split) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] += 1.0e-3f * (float)it;                     // keep the split from being hoisted
                split8<0>(a, ah, am, al);
                sink += __builtin_bit_cast(u32x4, ah)[0] + __builtin_bit_cast(u32x4, am)[1] + __builtin_bit_cast(u32x4, al)[2];
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + sink;
}

// (e) ds_read_b64_tr_b16 semantics: LDS holds u16 element ids of a [64 rows][64 cols] row-major tile (128-byte rows); lane
// 4q+p of every 16-lane group g supplies the address of (row R0 + 4g + q, cols C0 + 4p .. +3); hypothesis (guide T10): lane i of
// the group receives, in element q, the value at (row R0 + 4g + q, col C0 + i).
__global__ void tr_probe(unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 64];
    for (int e = threadIdx.x; e < 64 * 64; e += 64) tile[e] = (unsigned short)e;
    __syncthreads();
    const int l = threadIdx.x, grp = l >> 4, q = (l >> 2) & 3, p = l & 3;
    const int R0 = 8, C0 = 16;
    const unsigned addr = (unsigned)(size_t)(&tile[(R0 + 4 * grp + q) * 64 + C0 + 4 * p]);
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    out[2 * l] = (unsigned)v;
    out[2 * l + 1] = (unsigned)(v >> 32);
}

static double now_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    // ---- (a) lane maps with exact integers: A[i][k] = i + 2k - 20, B[j][k] = 3j - k (both exactly representable in bf16)
    {
        const int K = 32;
        std::vector<float> A(16 * K), B(16 * K), C(256), R(256);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < K; ++k) { A[i * K + k] = (float)(i + 2 * k - 20); B[i * K + k] = (float)(3 * i - k); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[j * K + k]; R[i * 16 + j] = (float)s; }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        gemm_tile<<<1, 64>>>(dA, dB, dC, K, 1);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int e = 0; e < 256; ++e) bad += C[e] != R[e];
        printf("(a) lane maps (A[row l&15][k=8(l>>4)+j], B same, C col=l&15 row=4(l>>4)+reg): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    // ---- (b) accuracy
    for (int K : {64, 256, 1024}) {
        std::vector<float> A(16 * K), B(16 * K), C(256);
        srand(7 + K);
        auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0) * (float)std::exp(((rand() / (double)RAND_MAX) - 0.5) * 4.0); };
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        std::vector<double> R(256), S(256);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[j * K + k]; sa += std::fabs((double)A[i * K + k] * B[j * K + k]); }
            R[i * 16 + j] = s; S[i * 16 + j] = sa;
        }
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const char* names[4] = {"fp32 MFMA chain      ", "bf16x3 6 products trunc", "bf16x3 6 products rne  ", "bf16x2 3 products      "};
        for (int mode = 0; mode < 4; ++mode) {
            gemm_tile<<<1, 64>>>(dA, dB, dC, K, mode);
            hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
            double worst = 0, rms = 0;
            for (int e = 0; e < 256; ++e) { const double err = std::fabs(C[e] - R[e]) / S[e]; worst = err > worst ? err : worst; rms += err * err; }
            printf("(b) K=%4d %s  max |err|/sum|ab| = %.3e   rms = %.3e   (fp32 eps = 5.96e-8)\n", K, names[mode], worst, std::sqrt(rms / 256));
        }
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    // ---- (e) transposed LDS read
    {
        unsigned* d;
        hipMalloc(&d, 128 * 4);
        tr_probe<<<1, 64>>>(d);
        unsigned h[128];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const int grp = l >> 4, i = l & 15;
            for (int q = 0; q < 4; ++q) {
                const unsigned got = (h[2 * l + (q >> 1)] >> (16 * (q & 1))) & 0xffffu;
                const unsigned want = (unsigned)((8 + 4 * grp + q) * 64 + 16 + i);
                bad += got != want;
            }
        }
        printf("(e) ds_read_b64_tr_b16: lane i of a 16-lane group gets column i of the group's 4 rows, row q in element q: %s (%d mismatches)\n",
               bad ? "WRONG" : "confirmed", bad);
        if (bad) for (int l = 0; l < 20; ++l) printf("    lane %2d: %5u %5u %5u %5u\n", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16);
        hipFree(d);
    }
    // ---- (c)/(d) rates: 256 blocks x 512 threads, nwaves active per block
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    struct { int mode, waves; const char* what; } runs[] = {
        {0, 4, "fp32 MFMA, 32 per step-set (4 tiles x 8), 1 wave/SIMD"}, {0, 8, "fp32 MFMA, 2 waves/SIMD"},
        {1, 4, "bf16 MFMA, 24 per step-set (4 tiles x 6), 1 wave/SIMD"}, {1, 8, "bf16 MFMA, 2 waves/SIMD"},
        {2, 4, "split8 x4 per iteration (VALU only), 1 wave/SIMD"}, {2, 8, "split8 x4, 2 waves/SIMD"},
        {3, 8, "waves 0-3 bf16 MFMA + waves 4-7 split8 (co-execution)"}};
    for (auto& r : runs) {
        rate<<<256, 512>>>(out, 100, r.mode, r.waves);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        rate<<<256, 512>>>(out, iters, r.mode, r.waves);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        const double ms = now_ms(e0, e1);
        printf("(c) %-62s %8.3f ms  = %7.1f ns per iteration per wave\n", r.what, ms, ms * 1e6 / iters);
    }
    printf("    one iteration = the K = 32 step of FOUR 16x16 tiles: fp32 needs 32 MFMAs (32 cycles each), bf16x3 24 MFMAs (16 cycles each)\n");
    return 0;
}
