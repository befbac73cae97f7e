// Diagnostic probe (not product).  VERDICT r3 item 1(a): does the bf16 matrix instruction the default kernels run
// (v_mfma_f32_16x16x32_bf16) co-execute with VALU work of ANOTHER wave on the same SIMD?  Rounds 1-3 only ever probed
// v_mfma_f32_16x16x4_f32, which issues at the fp32 VECTOR rate.
//
// Part 1 (pure streams): waves 0-3 issue MFMAs back to back (4 independent accumulators), their SIMD partners (waves 4-7) run
//   independent v_fma_f32 / v_pk_fma_f32 chains.  Rates alone / together, for the fp32 and the bf16 instruction.
// Part 2 (the shapes of attn_bwd3_kernel): role G = the bf16x3 Q|K|V projection as b3_gemm_rows runs it (A fragments from LDS
//   planes, pre-split weight fragments from L2, six MFMAs per product, fp32 epilogue stores to LDS); role V = pass 1 of the
//   backward attention core on fp32 LDS tiles (float2 row reads, packed FMAs, exp2).  One "chunk" = G x GREP + V x VREP, which
//   matches a chunk's MFMA : VALU volume (2688 MFMAs against two VALU passes).  Arrangements:
//     seq8    all 8 waves run G, barrier, all 8 waves run V, barrier                      (today's kernel)
//     spec    waves 0-3 run ALL of G while waves 4-7 run ALL of V, barrier                (wave specialisation)
//     stagger waves 0-3: G half then V half; waves 4-7: V half then G half; barrier between the halves
//                                                                                        (MI355X_MICROARCH.md "Two waves per SIMD" item 9)
//   plus each role alone on 8 and on 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------ part 1
// mfma_kind 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16.  valu_kind 0: v_fma_f32, 1: v_pk_fma_f32.
// mode bit 0: waves 0-3 work, bit 1: waves 4-7 work.
template <int MK, int VK>
__global__ void __launch_bounds__(512) pure(float* out, long long* cyc, int iters, int mode) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const float a = 1.0f + l * 1e-3f, b = 0.5f - l * 1e-3f;
    float res = 0.f;
    const long long t0 = clock64();
    if (w < 4 && !(mode & 4)) {
        if (mode & 1) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            if (MK == 0) {
                for (int i = 0; i < iters; ++i) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
                }
            } else {
                bf16x8 fa, fb;
#pragma unroll
                for (int e = 0; e < 8; ++e) { fa[e] = (short)(0x3f80 + l + e); fb[e] = (short)(0x3f00 + 2 * l + e); }
                for (int i = 0; i < iters; ++i) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, c3, 0, 0, 0);
                }
            }
            res = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else if (mode & 6) {                                           // mode 4: all eight waves run the VALU loop (two per SIMD)
        if (VK == 0) {
            float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, v4 = a + 2, v5 = b + 2, v6 = a + 3, v7 = b + 3;
            for (int i = 0; i < iters; ++i) {                         // 16 v_fma_f32 per iteration
                v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
                v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
                v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
                v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
            }
            res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
        } else {
            f32x2 pa = {a, b}, pb = {b, a};
            f32x2 v0 = {a, b}, v1 = {b, a}, v2 = {a + 1, b}, v3 = {b + 1, a}, v4 = {a + 2, b}, v5 = {b + 2, a}, v6 = {a + 3, b}, v7 = {b + 3, a};
            for (int i = 0; i < iters; ++i) {                         // 16 v_pk_fma_f32 per iteration
                v0 = __builtin_elementwise_fma(v0, pa, pb); v1 = __builtin_elementwise_fma(v1, pa, pb);
                v2 = __builtin_elementwise_fma(v2, pa, pb); v3 = __builtin_elementwise_fma(v3, pa, pb);
                v4 = __builtin_elementwise_fma(v4, pa, pb); v5 = __builtin_elementwise_fma(v5, pa, pb);
                v6 = __builtin_elementwise_fma(v6, pa, pb); v7 = __builtin_elementwise_fma(v7, pa, pb);
                v0 = __builtin_elementwise_fma(v0, pa, pb); v1 = __builtin_elementwise_fma(v1, pa, pb);
                v2 = __builtin_elementwise_fma(v2, pa, pb); v3 = __builtin_elementwise_fma(v3, pa, pb);
                v4 = __builtin_elementwise_fma(v4, pa, pb); v5 = __builtin_elementwise_fma(v5, pa, pb);
                v6 = __builtin_elementwise_fma(v6, pa, pb); v7 = __builtin_elementwise_fma(v7, pa, pb);
            }
            res = v0.x + v1.y + v2.x + v3.y + v4.x + v5.y + v6.x + v7.y;
        }
    }
    const long long t1 = clock64();
    if (blockIdx.x == 0 && l == 0) cyc[w] = t1 - t0;
    out[(blockIdx.x * 512 + threadIdx.x) % 4096] = res;
}

template <int MK, int VK>
static void run_pure(float* out, long long* cyc, const char* mname, const char* vname) {
    const int iters = 20000;
    double m_alone = 0, v_alone = 0;
    for (int mode = 1; mode <= 3; ++mode) {
        long long h[8];
        pure<MK, VK><<<256, 512>>>(out, cyc, iters, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
        const double m = (double)h[0] / iters / 4.0, v = (double)h[4] / iters / 16.0;
        if (mode == 1) m_alone = m;
        if (mode == 2) v_alone = v;
        if (mode == 1) printf("  %-28s alone          : %6.2f cycles per MFMA\n", mname, m);
        if (mode == 2) printf("  %-28s alone          : %6.2f cycles per VALU instruction (one wave per SIMD)\n", vname, v);
        if (mode == 3) printf("  together on each SIMD (waves w, w+4)        : %6.2f cycles per MFMA (x%.2f), %6.2f cycles per VALU instruction (x%.2f)\n",
                              m, m / m_alone, v, v / v_alone);
    }
    {
        long long h[8];
        pure<MK, VK><<<256, 512>>>(out, cyc, iters, 4);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
        printf("  %-28s on TWO waves per SIMD : %6.2f cycles per VALU instruction per SIMD\n", vname, (double)h[4] / iters / 16.0 / 2.0);
    }
}

// ------------------------------------------------------------------------------------------------ part 2
constexpr int ROWS = 64, LDQ = 244, LDT = 84, H = 8, DH = 10, I = 80;
constexpr int XP = 64 * 128;                                       // one bf16 plane of a [64][64] tile
constexpr size_t OFF_QKV = 3 * XP, OFF_OB = OFF_QKV + (size_t)ROWS * LDQ * 4, OFF_DOB = OFF_OB + (size_t)ROWS * LDT * 4,
                 OFF_MISC = OFF_DOB + (size_t)ROWS * LDT * 4, SMEM = OFF_MISC + (size_t)ROWS * H * 4 * 2;
struct B3 { bf16x8 h, m, l; };
__device__ __forceinline__ int plane_off(int r, int o) { return r * 128 + 16 * (o ^ (r & 7)); }
__device__ __forceinline__ B3 row_frag(const char* base, int mt, int s) {
    const int l = threadIdx.x & 63, a = plane_off(16 * mt + (l & 15), 4 * s + (l >> 4));
    return B3{*reinterpret_cast<const bf16x8*>(base + a), *reinterpret_cast<const bf16x8*>(base + XP + a),
              *reinterpret_cast<const bf16x8*>(base + 2 * XP + a)};
}
__device__ __forceinline__ B3 wfrag(const uint4* wb, int nt, int s) {
    const uint4* p = wb + ((size_t)(nt * 2 + s) * 3) * 64 + (threadIdx.x & 63);
    return B3{__builtin_bit_cast(bf16x8, p[0]), __builtin_bit_cast(bf16x8, p[64]), __builtin_bit_cast(bf16x8, p[128])};
}
__device__ __forceinline__ void mfma3x2(f32x4 (&acc)[2], const B3 (&a)[2], const B3& b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].l, b.h, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b.l, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].m, b.m, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].m, b.h, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b.m, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].h, b.h, acc[i], 0, 0, 0);
}
// role G: the row pair `mp` (rows 32 mp .. +31) against column tiles nt0, nt0 + ntstep, ... < 15 (K = 64: two K-steps)
__device__ __forceinline__ void role_g(const char* xp, const uint4* wb, float* qkv, int mp, int nt0, int ntstep) {
    const int l = threadIdx.x & 63, mt0 = 2 * mp;
    int nt = nt0;
    if (nt >= 15) return;
    B3 b = wfrag(wb, nt, 0);
    B3 a[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) a[i][s] = row_frag(xp, mt0 + i, s);
    for (; nt < 15; nt += ntstep) {
        f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool last = s == 1;
            const B3 bn = wfrag(wb, last ? (nt + ntstep < 15 ? nt + ntstep : nt) : nt, last ? 0 : s + 1);
            const B3 as[2] = {a[0][s], a[1][s]};
            mfma3x2(acc, as, b);
            b = bn;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) qkv[(size_t)((mt0 + i) * 16 + (l >> 4) * 4 + r) * LDQ + 16 * nt + (l & 15)] = acc[i][r];
        __builtin_amdgcn_sched_barrier(0);
    }
}
struct V10 {
    float v[DH];
    __device__ __forceinline__ void load(const float* p) {
#pragma unroll
        for (int c = 0; c < DH; c += 2) { const float2 t = *reinterpret_cast<const float2*>(p + c); v[c] = t.x; v[c + 1] = t.y; }
    }
    __device__ __forceinline__ float dot(const V10& o) const {
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < DH; c += 2) { const f32x2 x = {v[c], v[c + 1]}, y = {o.v[c], o.v[c + 1]}; acc = __builtin_elementwise_fma(x, y, acc); }
        return acc.x + acc.y;
    }
    __device__ __forceinline__ void axpy(float a, const V10& x) {
#pragma unroll
        for (int c = 0; c < DH; ++c) v[c] = fmaf(a, x.v[c], v[c]);
    }
};
// role V: pass 1 of the backward core for tasks t0, t0 + tstep, ... < ntasks  (task = (sequence, head, query))
__device__ __forceinline__ float role_v(const float* qkv, float* ob, const float* dob, const float* lses, float* dlt, int L, int nsq,
                                        int t0, int tstep) {
    const int ntasks = nsq * H * L;
    float sink = 0.f;
    for (int task = t0; task < ntasks; task += tstep) {
        const int i = task % L, h = (task / L) % H, sq = task / (L * H);
        const int row_i = sq * L + i, ho = h * DH;
        float* opp = ob + (size_t)row_i * LDT + ho;
        V10 q, go, dq, kv;
        q.load(qkv + (size_t)row_i * LDQ + ho);
        go.load(dob + (size_t)row_i * LDT + ho);
        kv.load(opp);
        const float delta = go.dot(kv);
#pragma unroll
        for (int c = 0; c < DH; ++c) dq.v[c] = 0.f;
        dlt[row_i * H + h] = delta;
        const float lse = lses[row_i * H + h];
        const float* kbase = qkv + (size_t)(sq * L) * LDQ + I + ho;
        for (int j = 0; j < L; ++j) {
            const float* kp = kbase + (size_t)j * LDQ;
            kv.load(kp + I);
            const float dp = go.dot(kv);
            kv.load(kp);
            const float p = __builtin_amdgcn_exp2f(q.dot(kv) * 0.45f - lse);
            dq.axpy(p * (dp - delta), kv);
        }
#pragma unroll
        for (int c = 0; c < DH; c += 2) *reinterpret_cast<float2*>(opp + c) = make_float2(dq.v[c] * 0.3f, dq.v[c + 1] * 0.3f);
        sink += dq.v[0];
    }
    return sink;
}

enum { ARR_SEQ8 = 0, ARR_SPEC = 1, ARR_STAGGER = 2, ARR_G8 = 3, ARR_V8 = 4, ARR_G4 = 5, ARR_V4 = 6, ARR_SPEC_PRIO = 7, ARR_N = 8 };
static const char* arr_name[ARR_N] = {"seq8    (all G | all V: today)", "spec    (waves 0-3 G || waves 4-7 V)", "stagger (G half/V half, crossed)",
                                      "G alone on 8 waves", "V alone on 8 waves", "G alone on waves 0-3", "V alone on waves 4-7",
                                      "spec + s_setprio 3 on the V waves"};

template <int GREP, int VREP>
__global__ void __launch_bounds__(512) shapes(const uint4* wb, float* out, long long* cyc, int L, int iters, int arr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* xp = smem;
    float* qkv = reinterpret_cast<float*>(smem + OFF_QKV);
    float* ob = reinterpret_cast<float*>(smem + OFF_OB);
    float* dob = reinterpret_cast<float*>(smem + OFF_DOB);
    float* lses = reinterpret_cast<float*>(smem + OFF_MISC);
    float* dlt = lses + ROWS * H;
    for (int e = threadIdx.x; e < (int)(OFF_QKV / 2); e += 512)          // bf16 planes: values around 1
        reinterpret_cast<unsigned short*>(smem)[e] = (unsigned short)(0x3f00 + ((e * 2654435761u) >> 26));
    for (int e = threadIdx.x; e < (int)((SMEM - OFF_QKV) / 4); e += 512)
        reinterpret_cast<float*>(smem + OFF_QKV)[e] = 1e-3f * (float)((e * 2654435761u) >> 22) - 0.5f;
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nsq = ROWS / L;
    float sink = 0.f;
    if (arr == ARR_SPEC_PRIO && w >= 4) __builtin_amdgcn_s_setprio(3);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (arr == ARR_SEQ8 || arr == ARR_G8 || arr == ARR_V8) {
            if (arr != ARR_V8) {
                for (int rep = 0; rep < GREP; ++rep) role_g(xp, wb, qkv, w >> 2, w & 3, 4);
                __syncthreads();
            }
            if (arr != ARR_G8) {
                for (int rep = 0; rep < VREP; ++rep) sink += role_v(qkv, ob, dob, lses, dlt, L, nsq, threadIdx.x, 512);
                __syncthreads();
            }
        } else if (arr == ARR_SPEC || arr == ARR_SPEC_PRIO || arr == ARR_G4 || arr == ARR_V4) {
            if (w < 4) {
                if (arr != ARR_V4)
                    for (int rep = 0; rep < GREP; ++rep) {
                        role_g(xp, wb, qkv, 0, w, 4);
                        role_g(xp, wb, qkv, 1, w, 4);
                    }
            } else if (arr != ARR_G4) {
                for (int rep = 0; rep < VREP; ++rep) sink += role_v(qkv, ob, dob, lses, dlt, L, nsq, threadIdx.x - 256, 256);
            }
            __syncthreads();
        } else {   // stagger: each wave group runs half of G (its row pair) and half of V (its half of the tasks), in crossed order
            const int half_tasks = (nsq * H * L + 1) / 2;
            (void)half_tasks;
            if (w < 4) {
                for (int rep = 0; rep < GREP; ++rep) role_g(xp, wb, qkv, 0, w, 4);
            } else {
                for (int rep = 0; rep < VREP; ++rep) sink += role_v(qkv, ob, dob, lses, dlt, L, nsq, 2 * (threadIdx.x - 256) + 1, 512);
            }
            __syncthreads();
            if (w < 4) {
                for (int rep = 0; rep < VREP; ++rep) sink += role_v(qkv, ob, dob, lses, dlt, L, nsq, 2 * threadIdx.x, 512);
            } else {
                for (int rep = 0; rep < GREP; ++rep) role_g(xp, wb, qkv, 1, w & 3, 4);
            }
            __syncthreads();
        }
    }
    const long long t1 = clock64();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[w] = (t1 - t0) / iters;
    out[(blockIdx.x * 512 + threadIdx.x) % 4096] = sink;
}

template <int GREP, int VREP>
static void run_shapes(const uint4* wb, float* out, long long* cyc, int L) {
    const int iters = 40;
    printf("  one chunk = G x %d (%d MFMAs) + V x %d, L = %d (%d of 512 core tasks per pass)\n", GREP, GREP * 720, VREP, L, (64 / L) * H * L);
    CK(hipFuncSetAttribute((const void*)shapes<GREP, VREP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double us[ARR_N];
    for (int arr = 0; arr < ARR_N; ++arr) {
        shapes<GREP, VREP><<<256, 512, SMEM>>>(wb, out, cyc, L, iters, arr);      // warm
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            shapes<GREP, VREP><<<256, 512, SMEM>>>(wb, out, cyc, L, iters, arr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        long long h[8];
        CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
        us[arr] = 1e3 * best / iters;
        printf("    %-40s %8.2f us per chunk   (wave 0: %lld cycles, wave 4: %lld cycles per chunk)\n", arr_name[arr], us[arr], h[0], h[4]);
    }
    printf("    => spec / seq8 = %.3f, stagger / seq8 = %.3f, (G8 + V8) / seq8 = %.3f, max(G4, V4) / seq8 = %.3f\n", us[ARR_SPEC] / us[ARR_SEQ8],
           us[ARR_STAGGER] / us[ARR_SEQ8], (us[ARR_G8] + us[ARR_V8]) / us[ARR_SEQ8], (us[ARR_G4] > us[ARR_V4] ? us[ARR_G4] : us[ARR_V4]) / us[ARR_SEQ8]);
}

int main() {
    float* out; long long* cyc; uint4* wb;
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 64));
    const size_t wbytes = (size_t)15 * 2 * 3 * 1024;
    CK(hipMalloc(&wb, wbytes));
    {
        std::vector<unsigned short> h(wbytes / 2);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 26));
        CK(hipMemcpy(wb, h.data(), wbytes, hipMemcpyHostToDevice));
    }
    printf("== part 1: pure instruction streams, 256 work-groups x 512 threads (waves w and w + 4 share a SIMD)\n");
    printf(" fp32 MFMA + v_fma_f32\n");        run_pure<0, 0>(out, cyc, "v_mfma_f32_16x16x4_f32", "v_fma_f32");
    printf(" fp32 MFMA + v_pk_fma_f32\n");     run_pure<0, 1>(out, cyc, "v_mfma_f32_16x16x4_f32", "v_pk_fma_f32");
    printf(" bf16 MFMA + v_fma_f32\n");        run_pure<1, 0>(out, cyc, "v_mfma_f32_16x16x32_bf16", "v_fma_f32");
    printf(" bf16 MFMA + v_pk_fma_f32\n");     run_pure<1, 1>(out, cyc, "v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32");
    printf("== part 2: the shapes of attn_bwd3_kernel (bf16x3 projection || VALU backward core)\n");
    run_shapes<4, 2>(wb, out, cyc, 21);
    run_shapes<4, 2>(wb, out, cyc, 11);
    run_shapes<2, 2>(wb, out, cyc, 21);
    return 0;
}
