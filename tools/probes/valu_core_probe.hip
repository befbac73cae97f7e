// Diagnostic probe (not product): attention-backward pass 1 (one lane per (sequence, head, query), loop over keys) as a
// function of waves per SIMD: 512 threads / full key loop vs 1024 threads / key loop split over two lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ROWS = 64, LDQ = 244, LDT = 84, H = 8, DH = 10, I = 80;
struct V10 { float v[DH]; };
#ifndef LDVOL
#define LDVOL
#endif
typedef float vf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ld(V10& x, const float* p) {
#pragma unroll
    for (int c = 0; c < DH; c += 2) { const vf2 t = *reinterpret_cast<const LDVOL vf2*>(p + c); x.v[c] = t.x; x.v[c + 1] = t.y; }
}
__device__ __forceinline__ float dot(const V10& a, const V10& b) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 2) { s0 = fmaf(a.v[c], b.v[c], s0); s1 = fmaf(a.v[c + 1], b.v[c + 1], s1); }
    return s0 + s1;
}
template <int NT, int SPLIT>
__global__ void __launch_bounds__(NT) pass1(float* out, long long* cyc, int L, int iters) {
    extern __shared__ float sm[];
    float* qkv = sm;
    float* dob = qkv + ROWS * LDQ;
    float* ob = dob + ROWS * LDT;
    float* lses = ob + ROWS * LDT;
    for (int e = threadIdx.x; e < ROWS * (LDQ + 2 * LDT + H); e += NT) sm[e] = 0.01f * (float)((e * 2654435761u) >> 20) - 20.f;
    __syncthreads();
    const int nsq = ROWS / L, ntasks = nsq * H * L * SPLIT;
    long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int task = threadIdx.x; task < ntasks; task += NT) {
            const int part = task % SPLIT, tq = task / SPLIT;
            const int i = tq % L, h = (tq / L) % H, sq = tq / (L * H);
            const int row_i = sq * L + i;
            V10 q, go, kv, dq;
            ld(q, qkv + row_i * LDQ + h * DH);
            ld(go, dob + row_i * LDT + h * DH);
            ld(kv, ob + row_i * LDT + h * DH);
            const float delta = dot(go, kv);
#pragma unroll
            for (int c = 0; c < DH; ++c) dq.v[c] = 0.f;
            const float lse = lses[row_i * H + h];
            const float* kbase = qkv + (sq * L) * LDQ + I + h * DH;
            for (int j = part; j < L; j += SPLIT) {
                const float* kp = kbase + j * LDQ;
                ld(kv, kp + I);
                const float dp = dot(go, kv);
                ld(kv, kp);
                const float p = __builtin_amdgcn_exp2f(dot(q, kv) * 0.4f - lse);
                const float w = p * (dp - delta);
#pragma unroll
                for (int c = 0; c < DH; ++c) dq.v[c] = fmaf(w, kv.v[c], dq.v[c]);
            }
            if (SPLIT == 2) {
#pragma unroll
                for (int c = 0; c < DH; ++c) dq.v[c] += __shfl_xor(dq.v[c], 1, 64);
            }
            if (part == 0) {
                float* op = ob + row_i * LDT + h * DH;
#pragma unroll
                for (int c = 0; c < DH; c += 2) *reinterpret_cast<float2*>(op + c) = make_float2(dq.v[c] * 1e-3f, dq.v[c + 1] * 1e-3f);
            }
            acc += dq.v[0];
        }
        __syncthreads();
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / iters;
    out[(blockIdx.x * NT + threadIdx.x) % 4096] = acc;
}
// K and V rows through hand-issued ds_read_b64 (hipcc merges adjacent b64 loads into ds_read2_b64, which the LDS serves at half
// the bytes per clock): loads of the NEXT key are issued before the wait for the current one (lgkmcnt(10) = ten newer loads in flight)
__device__ __forceinline__ void issue10(vf2 (&k)[5], vf2 (&v)[5], const float* kp) {
    const unsigned addr = (unsigned)(unsigned long long)kp;
    asm volatile("ds_read_b64 %0, %1" : "=v"(k[0]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(k[1]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(k[2]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:24" : "=v"(k[3]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(k[4]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:320" : "=v"(v[0]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:328" : "=v"(v[1]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:336" : "=v"(v[2]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:344" : "=v"(v[3]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1 offset:352" : "=v"(v[4]) : "v"(addr));
}
template <int N>
__device__ __forceinline__ void wait10(vf2 (&k)[5], vf2 (&v)[5]) {
    if (N == 10) asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k[0]), "+v"(k[1]), "+v"(k[2]), "+v"(k[3]), "+v"(k[4]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
}
__device__ __forceinline__ float dot5(const V10& a, const vf2 (&b)[5]) {
    vf2 acc = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) { const vf2 x = {a.v[2 * c], a.v[2 * c + 1]}; acc = __builtin_elementwise_fma(x, b[c], acc); }
    return acc.x + acc.y;
}
template <int NT>
__global__ void __launch_bounds__(NT) pass1_asm(float* out, long long* cyc, int L, int iters) {
    extern __shared__ float sm[];
    float* qkv = sm;
    float* dob = qkv + ROWS * LDQ;
    float* ob = dob + ROWS * LDT;
    float* lses = ob + ROWS * LDT;
    for (int e = threadIdx.x; e < ROWS * (LDQ + 2 * LDT + H); e += NT) sm[e] = 0.01f * (float)((e * 2654435761u) >> 20) - 20.f;
    __syncthreads();
    const int nsq = ROWS / L, ntasks = nsq * H * L;
    long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int task = threadIdx.x; task < ntasks; task += NT) {
            const int i = task % L, h = (task / L) % H, sq = task / (L * H);
            const int row_i = sq * L + i;
            V10 q, go, kv;
            ld(q, qkv + row_i * LDQ + h * DH);
            ld(go, dob + row_i * LDT + h * DH);
            ld(kv, ob + row_i * LDT + h * DH);
            const float delta = dot(go, kv);
            vf2 dq[5] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
            const float lse = lses[row_i * H + h];
            const float* kbase = qkv + (sq * L) * LDQ + I + h * DH;
            vf2 k0[5], v0[5], k1[5], v1[5];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue10(k0, v0, kbase);
            for (int j = 0; j < L; j += 2) {
                issue10(k1, v1, kbase + (j + 1 < L ? j + 1 : j) * LDQ);
                wait10<10>(k0, v0);
                {
                    const float dp = dot5(go, v0);
                    const float p = __builtin_amdgcn_exp2f(dot5(q, k0) * 0.4f - lse);
                    const float w = p * (dp - delta);
                    const vf2 w2 = {w, w};
#pragma unroll
                    for (int c = 0; c < 5; ++c) dq[c] = __builtin_elementwise_fma(w2, k0[c], dq[c]);
                }
                if (j + 1 < L) {
                    issue10(k0, v0, kbase + (j + 2 < L ? j + 2 : j + 1) * LDQ);
                    wait10<10>(k1, v1);
                    const float dp = dot5(go, v1);
                    const float p = __builtin_amdgcn_exp2f(dot5(q, k1) * 0.4f - lse);
                    const float w = p * (dp - delta);
                    const vf2 w2 = {w, w};
#pragma unroll
                    for (int c = 0; c < 5; ++c) dq[c] = __builtin_elementwise_fma(w2, k1[c], dq[c]);
                } else {
                    wait10<0>(k1, v1);
                }
            }
            wait10<0>(k0, v0);
            float* op = ob + row_i * LDT + h * DH;
#pragma unroll
            for (int c = 0; c < 5; ++c) *reinterpret_cast<float2*>(op + 2 * c) = make_float2(dq[c].x * 1e-3f, dq[c].y * 1e-3f);
            acc += dq[0].x;
        }
        __syncthreads();
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / iters;
    out[(blockIdx.x * NT + threadIdx.x) % 4096] = acc;
}
template <int A>
__device__ __forceinline__ float qb(float x) {      // value of lane A of this lane's quad (DPP quad_perm broadcast)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), A | (A << 2) | (A << 4) | (A << 6), 0xf, 0xf, true));
}
// acc += (lane A of the quad's x) * y, one v_fmac_f32 with a DPP quad_perm source (hipcc does not fold update_dpp into the FMA)
template <int A>
__device__ __forceinline__ void qfma(float& acc, float x, float y) {
    if (A == 0) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    if (A == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    if (A == 2) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    if (A == 3) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
}
template <int A>
__device__ __forceinline__ void key_step_asm(const V10& q, const V10& go, const V10& kq, const V10& vq, V10& dq, float lse, float delta, bool on) {
    float dp0 = 0.f, dp1 = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 2) {
        qfma<A>(dp0, vq.v[c], go.v[c]);
        qfma<A>(dp1, vq.v[c + 1], go.v[c + 1]);
        qfma<A>(s0, kq.v[c], q.v[c]);
        qfma<A>(s1, kq.v[c + 1], q.v[c + 1]);
    }
    const float p = __builtin_amdgcn_exp2f((s0 + s1) * 0.4f - lse);
    const float w = on ? p * ((dp0 + dp1) - delta) : 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) qfma<A>(dq.v[c], kq.v[c], w);
}
template <int A>
__device__ __forceinline__ void key_step(const V10& q, const V10& go, const V10& kq, const V10& vq, V10& dq, float lse, float delta, bool on) {
    float dp0 = 0.f, dp1 = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 2) {
        dp0 = fmaf(qb<A>(vq.v[c]), go.v[c], dp0);
        dp1 = fmaf(qb<A>(vq.v[c + 1]), go.v[c + 1], dp1);
        s0 = fmaf(qb<A>(kq.v[c]), q.v[c], s0);
        s1 = fmaf(qb<A>(kq.v[c + 1]), q.v[c + 1], s1);
    }
    const float p = __builtin_amdgcn_exp2f((s0 + s1) * 0.4f - lse);
    const float w = on ? p * ((dp0 + dp1) - delta) : 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) dq.v[c] = fmaf(qb<A>(kq.v[c]), w, dq.v[c]);
}
// quads = 4 queries of one (sequence, head); lane a of the quad loads the K / V rows of keys 4 kb + a and the quad shares them by DPP
template <int NT>
__global__ void __launch_bounds__(NT) pass1_quad(float* out, long long* cyc, int L, int iters) {
    extern __shared__ float sm[];
    float* qkv = sm;
    float* dob = qkv + ROWS * LDQ;
    float* ob = dob + ROWS * LDT;
    float* lses = ob + ROWS * LDT;
    for (int e = threadIdx.x; e < ROWS * (LDQ + 2 * LDT + H); e += NT) sm[e] = 0.01f * (float)((e * 2654435761u) >> 20) - 20.f;
    __syncthreads();
    const int nsq = ROWS / L, nqb = (L + 3) / 4, ntasks = nsq * H * nqb * 4;
    long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int task = threadIdx.x; task < ntasks; task += NT) {
            const int a = task & 3, u = task >> 2;
            const int blk = u % nqb, h = (u / nqb) % H, sq = u / (nqb * H);
            const int i = 4 * blk + a, row_i = sq * L + (i < L ? i : L - 1);
            V10 q, go, kv, dq, kq, vq;
            ld(q, qkv + row_i * LDQ + h * DH);
            ld(go, dob + row_i * LDT + h * DH);
            ld(kv, ob + row_i * LDT + h * DH);
            const float delta = dot(go, kv);
#pragma unroll
            for (int c = 0; c < DH; ++c) dq.v[c] = 0.f;
            const float lse = lses[row_i * H + h];
            const float* kbase = qkv + (sq * L) * LDQ + I + h * DH;
            for (int kb = 0; kb < nqb; ++kb) {
                const int jj = 4 * kb + a;
                const float* kp = kbase + (jj < L ? jj : L - 1) * LDQ;
                ld(kq, kp);
                ld(vq, kp + I);
                key_step<0>(q, go, kq, vq, dq, lse, delta, 4 * kb + 0 < L);
                key_step<1>(q, go, kq, vq, dq, lse, delta, 4 * kb + 1 < L);
                key_step<2>(q, go, kq, vq, dq, lse, delta, 4 * kb + 2 < L);
                key_step<3>(q, go, kq, vq, dq, lse, delta, 4 * kb + 3 < L);
            }
            if (i < L) {
                float* op = ob + row_i * LDT + h * DH;
#pragma unroll
                for (int c = 0; c < DH; c += 2) *reinterpret_cast<float2*>(op + c) = make_float2(dq.v[c] * 1e-3f, dq.v[c + 1] * 1e-3f);
            }
            acc += dq.v[0];
        }
        __syncthreads();
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / iters;
    out[(blockIdx.x * NT + threadIdx.x) % 4096] = acc;
}
// quads = 4 queries of one (sequence, head); lane a of the quad loads the K / V rows of keys 4 kb + a and the quad shares them by DPP
template <int NT>
__global__ void __launch_bounds__(NT) pass1_quad_asm(float* out, long long* cyc, int L, int iters) {
    extern __shared__ float sm[];
    float* qkv = sm;
    float* dob = qkv + ROWS * LDQ;
    float* ob = dob + ROWS * LDT;
    float* lses = ob + ROWS * LDT;
    for (int e = threadIdx.x; e < ROWS * (LDQ + 2 * LDT + H); e += NT) sm[e] = 0.01f * (float)((e * 2654435761u) >> 20) - 20.f;
    __syncthreads();
    const int nsq = ROWS / L, nqb = (L + 3) / 4, ntasks = nsq * H * nqb * 4;
    long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (int task = threadIdx.x; task < ntasks; task += NT) {
            const int a = task & 3, u = task >> 2;
            const int blk = u % nqb, h = (u / nqb) % H, sq = u / (nqb * H);
            const int i = 4 * blk + a, row_i = sq * L + (i < L ? i : L - 1);
            V10 q, go, kv, dq, kq, vq;
            ld(q, qkv + row_i * LDQ + h * DH);
            ld(go, dob + row_i * LDT + h * DH);
            ld(kv, ob + row_i * LDT + h * DH);
            const float delta = dot(go, kv);
#pragma unroll
            for (int c = 0; c < DH; ++c) dq.v[c] = 0.f;
            const float lse = lses[row_i * H + h];
            const float* kbase = qkv + (sq * L) * LDQ + I + h * DH;
            for (int kb = 0; kb < nqb; ++kb) {
                const int jj = 4 * kb + a;
                const float* kp = kbase + (jj < L ? jj : L - 1) * LDQ;
                ld(kq, kp);
                ld(vq, kp + I);
                key_step_asm<0>(q, go, kq, vq, dq, lse, delta, 4 * kb + 0 < L);
                key_step_asm<1>(q, go, kq, vq, dq, lse, delta, 4 * kb + 1 < L);
                key_step_asm<2>(q, go, kq, vq, dq, lse, delta, 4 * kb + 2 < L);
                key_step_asm<3>(q, go, kq, vq, dq, lse, delta, 4 * kb + 3 < L);
            }
            if (i < L) {
                float* op = ob + row_i * LDT + h * DH;
#pragma unroll
                for (int c = 0; c < DH; c += 2) *reinterpret_cast<float2*>(op + c) = make_float2(dq.v[c] * 1e-3f, dq.v[c + 1] * 1e-3f);
            }
            acc += dq.v[0];
        }
        __syncthreads();
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = (t1 - t0) / iters;
    out[(blockIdx.x * NT + threadIdx.x) % 4096] = acc;
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 8);
    const size_t smem = (size_t)ROWS * (LDQ + 2 * LDT + H) * 4;
    for (int L : {21, 11}) {
        long long h;
        pass1<512, 1><<<256, 512, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d  512 threads, full loop:        %lld cycles per pass\n", L, h);
        pass1<1024, 1><<<256, 1024, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d 1024 threads, full loop (half idle): %lld\n", L, h);
        pass1<1024, 2><<<256, 1024, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d 1024 threads, keys split over 2 lanes: %lld\n", L, h);
        pass1_asm<512><<<256, 512, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d  512 threads, hand-issued ds_read_b64, double-buffered: %lld\n", L, h);
        pass1_quad<512><<<256, 512, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d  512 threads, DPP quad sharing: %lld\n", L, h);
        pass1_quad_asm<512><<<256, 512, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d  512 threads, DPP quad sharing, asm fmac_dpp: %lld\n", L, h);
        pass1_quad_asm<1024><<<256, 1024, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d 1024 threads, DPP quad sharing, asm fmac_dpp: %lld\n", L, h);
        pass1_quad<1024><<<256, 1024, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d 1024 threads, DPP quad sharing: %lld\n", L, h);
        pass1<512, 2><<<256, 512, smem>>>(out, cyc, L, 200); (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("L=%d  512 threads, keys split over 2 lanes (2 rounds): %lld\n", L, h);
    }
    return 0;
}
