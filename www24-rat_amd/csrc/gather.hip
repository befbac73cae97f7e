// gather.hip — K1: token-grid assembly (embedding gather + label tokens) and its backward scatter.
//
// Forward replaces RAT_m2.forward lines 113-126 of the reference (3x EmbeddingLayer.forward over F
// nn.Embedding tables, 2x label_embedding_layer, 3x torch.concat): ONE pass that reads each embedding row
// once with 16-byte coalesced loads and writes the [B][T][S][d] grid once.  HBM-bound; algorithmic bytes
// per sample = T*F*d*4 (rows) + T*S*d*4 (grid) + T*L*4 (ids)  (SURVEY.md §8d).
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int GATHER_THREADS = 256;
#ifndef GATHER_BWD_ROWS
#define GATHER_BWD_ROWS 8                 // rows per wave of the d = 64 scatter (grid = rows / this, capped)
#endif
#ifndef GB_TRIP
#define GB_TRIP 2                         // rows requested before the first atomic of a trip
#endif
#ifndef RAT_GATHER_ITEMS
#define RAT_GATHER_ITEMS 2
#endif
// independent rows in flight per thread.  Same-box A/B, round 5 (profiles/round5/r5_gather_items_ab.txt; alone, 3 interleaved rounds):
// 2 rows 43.6 us on the 25.6 GB table (0.675 of 8 TB/s) and 74.4 us at N2 (0.80) against 46.2 us (0.636) / 78.4 us (0.76) with 4 rows,
// 1 row the same as 2, 3 and 8 rows slower; inside the N2 training step (behind the optimizer's sweep) 85-88 us with any of them.
constexpr int GATHER_ITEMS = RAT_GATHER_ITEMS;

// vectorised path: d % 4 == 0, one item = one 16-byte piece of one grid row
__global__ void __launch_bounds__(GATHER_THREADS)
gather_fwd_vec_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ label_ids,
                      const RatField* __restrict__ fields, const float* __restrict__ label_table,
                      float* __restrict__ grid, int64_t nrows, int S, int L, int d) {
    const int cpr = d >> 2;                                   // 16-byte pieces per row
    const int64_t nitems = nrows * cpr;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e0 < nitems; e0 += stride * GATHER_ITEMS) {
        const float4* src[GATHER_ITEMS];
        int extra[GATHER_ITEMS];                              // >0: bag field, number of extra ids to sum
        const int32_t* idp[GATHER_ITEMS];
        const float* tab[GATHER_ITEMS];
        int vocab[GATHER_ITEMS];
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) {
            const int64_t e = e0 + (int64_t)u * stride;
            src[u] = nullptr;
            extra[u] = 0;
            idp[u] = nullptr;
            tab[u] = nullptr;
            vocab[u] = 1;
            if (e < nitems) {
                const int64_t row = e / cpr;
                const int piece = (int)(e - row * cpr);
                const int64_t bt = row / S;
                const int s = (int)(row - bt * S);
                if (s == 0) {
                    int lab = label_ids[bt];
                    lab = lab < 0 ? 0 : (lab > 2 ? 2 : lab);  // memory safety only: rat_check_ids REPORTS such labels
                    src[u] = reinterpret_cast<const float4*>(label_table + (int64_t)lab * d) + piece;
                } else {
                    const RatField f = fields[s - 1];
                    const int32_t* ids = idx + bt * L + f.col;
                    int id = ids[0];
                    id = id < 0 ? 0 : (id >= f.vocab ? f.vocab - 1 : id);
                    src[u] = reinterpret_cast<const float4*>(f.table + (int64_t)id * d) + piece;
                    extra[u] = f.ncols - 1;
                    idp[u] = ids;
                    tab[u] = f.table + piece * 4;
                    vocab[u] = f.vocab;
                }
            }
        }
        float4 v[GATHER_ITEMS];
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) v[u] = src[u] ? *src[u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) {
            for (int j = 1; j <= extra[u]; ++j) {             // MaskedSumPooling bag (padding row is all-zero)
                int id = idp[u][j];
                id = id < 0 ? 0 : (id >= vocab[u] ? vocab[u] - 1 : id);
                const float4 w = *reinterpret_cast<const float4*>(tab[u] + (int64_t)id * d);
                v[u].x += w.x; v[u].y += w.y; v[u].z += w.z; v[u].w += w.w;
            }
        }
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) {
            const int64_t e = e0 + (int64_t)u * stride;
            if (e < nitems) reinterpret_cast<float4*>(grid)[e] = v[u];
        }
    }
}

// d = 64 and fewer than 2^31 grid rows: 32-bit index arithmetic (the path above spends two 64-bit divisions per 16-byte piece) and
// branch-free fetches — the label row and the field rows go through the same unconditional id load and row load, so that all
// GATHER_ITEMS rows of a lane are in flight together (a load inside a divergent branch is waited for inside the branch).
// NT: the grid leaves with non-temporal stores.  Same-box A/Bs (round 4, profiles/round4/r4_nt_ab.txt): inside the N2 training step
// (242 MB grid, behind the optimizer's sweep) 91-98 us with the hint against 103 us without; alone on the 25.6 GB table at B = 1024
// (118 MB grid: it fits the 256 MB Infinity Cache, where plain stores are absorbed) 46 us without against 50 us with — so the
// host asks for the hint only when the grid is larger than what the cache can take.
template <bool NT>
__global__ void __launch_bounds__(GATHER_THREADS)
gather_fwd_rows64_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ label_ids,
                         const RatField* __restrict__ fields, const float* __restrict__ label_table,
                         float* __restrict__ grid, unsigned nrows, unsigned S, int L) {
    constexpr int d = 64;
    const unsigned piece = threadIdx.x & 15;                                    // 16 lanes per 256-byte row
    const unsigned slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, nslots = (gridDim.x * blockDim.x) >> 4;
    for (unsigned r0 = slot; r0 < nrows; r0 += GATHER_ITEMS * nslots) {
        const float* tab[GATHER_ITEMS];
        const int32_t* idp[GATHER_ITEMS];
        int vocab[GATHER_ITEMS], extra[GATHER_ITEMS], id[GATHER_ITEMS];
        bool ok[GATHER_ITEMS];
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) {
            const unsigned row = r0 + u * nslots;
            ok[u] = row < nrows;
            const unsigned rr = ok[u] ? row : 0u;
            const unsigned bt = rr / S, s = rr - bt * S;
            const RatField f = fields[s > 0 ? s - 1 : 0];
            const bool lab = s == 0;
            tab[u] = lab ? label_table : f.table;
            vocab[u] = lab ? 3 : f.vocab;
            extra[u] = lab ? 0 : f.ncols - 1;
            idp[u] = lab ? label_ids + bt : idx + (int64_t)bt * L + f.col;
        }
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) id[u] = *idp[u];
        float4 v[GATHER_ITEMS];
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) {
            const int i = id[u] < 0 ? 0 : (id[u] >= vocab[u] ? vocab[u] - 1 : id[u]);   // memory safety only: rat_check_ids REPORTS
            v[u] = *(reinterpret_cast<const float4*>(tab[u] + (int64_t)i * d) + piece);
        }
#pragma unroll
        for (int u = 0; u < GATHER_ITEMS; ++u) {
            for (int j = 1; j <= extra[u]; ++j) {                                // MaskedSumPooling bag (padding row is all-zero)
                int i = idp[u][j];
                i = i < 0 ? 0 : (i >= vocab[u] ? vocab[u] - 1 : i);
                const float4 w = *(reinterpret_cast<const float4*>(tab[u] + (int64_t)i * d) + piece);
                v[u].x += w.x; v[u].y += w.y; v[u].z += w.z; v[u].w += w.w;
            }
            if (ok[u]) {
                float* dst = grid + ((size_t)(r0 + u * nslots) * 16 + piece) * 4;
                if (NT) rat_st4_stream(dst, v[u]);
                else *reinterpret_cast<float4*>(dst) = v[u];
            }
        }
    }
}

// generic path: any d, one item = one float
__global__ void __launch_bounds__(GATHER_THREADS)
gather_fwd_scalar_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ label_ids,
                         const RatField* __restrict__ fields, const float* __restrict__ label_table,
                         float* __restrict__ grid, int64_t nrows, int S, int L, int d) {
    const int64_t nitems = nrows * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nitems; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / d;
        const int c = (int)(e - row * d);
        const int64_t bt = row / S;
        const int s = (int)(row - bt * S);
        float v;
        if (s == 0) {
            int lab = label_ids[bt];
            lab = lab < 0 ? 0 : (lab > 2 ? 2 : lab);
            v = label_table[(int64_t)lab * d + c];
        } else {
            const RatField f = fields[s - 1];
            const int32_t* ids = idx + bt * L + f.col;
            v = 0.f;
            for (int j = 0; j < f.ncols; ++j) {
                int id = ids[j];
                id = id < 0 ? 0 : (id >= f.vocab ? f.vocab - 1 : id);
                v += f.table[(int64_t)id * d + c];
            }
        }
        grid[e] = v;
    }
}

// backward: one lane per grid element; field rows -> fp32 atomics shaped as contiguous row segments
// (a 64-lane wave covers one 256-byte row at d = 64: the full-rate atomic shape on gfx950).
__global__ void __launch_bounds__(GATHER_THREADS)
gather_bwd_fields_kernel(const float* __restrict__ dgrid, const float* __restrict__ dflat,
                         const int32_t* __restrict__ idx, const RatField* __restrict__ gfields,
                         int64_t nbt, int T, int S, int L, int d) {
    const int F = S - 1;
    const int64_t nitems = nbt * F * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nitems; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % d);
        const int64_t r = e / d;
        const int fi = (int)(r % F);
        const int64_t bt = r / F;
        float g = dgrid[((bt * S) + 1 + fi) * d + c];
        if (dflat != nullptr && (bt % T) == 0) g += dflat[((bt / T) * F + fi) * d + c];
        const RatField f = gfields[fi];
        const int32_t* ids = idx + bt * L + f.col;
        for (int j = 0; j < f.ncols; ++j) {
            int id = ids[j];
            id = id < 0 ? 0 : (id >= f.vocab ? f.vocab - 1 : id);
            if (id != f.padding_idx) atomicAdd(f.table + (int64_t)id * d + c, g);
        }
    }
}

// The same for d = 64 (a wave = one 256-byte row): the row index is wave-uniform, so its decomposition into (sample row, field)
// and the id / field-descriptor fetches run on the scalar unit — the element-wise form above spends most of its time in three
// 64-bit integer divisions per element.  GB_TRIP rows per trip: their gradient rows are requested before the atomics of the first.
__global__ void __launch_bounds__(GATHER_THREADS)
gather_bwd_fields64_kernel(const float* __restrict__ dgrid, const float* __restrict__ dflat, const int32_t* __restrict__ idx,
                           const RatField* __restrict__ gfields, int nrows, int T, int S, int L) {
    constexpr int d = 64;
    const int lane = threadIdx.x & 63;
#ifdef RAT_EMU
    const int wave0 = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
#else
    const int wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
#endif
    const int nwaves = (int)(gridDim.x * blockDim.x) >> 6;
    const int F = S - 1;
    auto fetch = [&](int r, int& bt, int& fi) {
        bt = r / F;
        fi = r - bt * F;
        float g = dgrid[((int64_t)bt * S + 1 + fi) * d + lane];
        if (dflat != nullptr) {
            const int b = bt / T;
            if (bt - b * T == 0) g += dflat[((int64_t)b * F + fi) * d + lane];
        }
        return g;
    };
    auto scatter = [&](int bt, int fi, float g) {
        const RatField f = gfields[fi];
        const int32_t* ids = idx + (int64_t)bt * L + f.col;
        for (int j = 0; j < f.ncols; ++j) {
            int id = ids[j];
            id = id < 0 ? 0 : (id >= f.vocab ? f.vocab - 1 : id);
            if (id != f.padding_idx) atomicAdd(f.table + (int64_t)id * d + lane, g);
        }
    };
    int r = wave0;
    for (; r + (GB_TRIP - 1) * nwaves < nrows; r += GB_TRIP * nwaves) {
        int bt[GB_TRIP], fi[GB_TRIP];
        float g[GB_TRIP];
#pragma unroll
        for (int u = 0; u < GB_TRIP; ++u) g[u] = fetch(r + u * nwaves, bt[u], fi[u]);
#pragma unroll
        for (int u = 0; u < GB_TRIP; ++u) scatter(bt[u], fi[u], g[u]);
    }
    for (; r < nrows; r += nwaves) {
        int bt0, fi0;
        const float g0 = fetch(r, bt0, fi0);
        scatter(bt0, fi0, g0);
    }
}

// label-token rows: only 3 destination rows.  Thread = (column, row group) keeps its three label rows in registers over the block's slice
// of the (sample, target / retrieved) rows (four rows in flight), row groups meet in LDS, then 3 d atomics per block.  (The first
// version added every element to LDS with an atomic: 28 us at the north-star shape, all of it contention on 3 d addresses.)
__global__ void __launch_bounds__(GATHER_THREADS)
gather_bwd_label_kernel(const float* __restrict__ dgrid, const int32_t* __restrict__ label_ids,
                        float* __restrict__ dlabel, int64_t nbt, int S, int d) {
    RAT_DYN_SMEM(smem);
    float* red = reinterpret_cast<float*>(smem);             // [groups][3][d]
    const int groups = GATHER_THREADS / d;                   // d <= GATHER_THREADS (checked on the host)
    const int c = threadIdx.x % d, rg = threadIdx.x / d;
    const int64_t per = (nbt + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < nbt ? r0 + per : nbt;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (rg < groups) {
        int64_t bt = r0 + rg;
        for (; bt + 3 * groups < r1; bt += 4 * groups) {
            float g[4];
            int lab[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = dgrid[((bt + (int64_t)u * groups) * S) * d + c];
                lab[u] = label_ids[bt + (int64_t)u * groups];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 += lab[u] <= 0 ? g[u] : 0.f;
                a1 += lab[u] == 1 ? g[u] : 0.f;
                a2 += lab[u] >= 2 ? g[u] : 0.f;
            }
        }
        for (; bt < r1; bt += groups) {
            const float g = dgrid[(bt * S) * d + c];
            const int lab = label_ids[bt];
            a0 += lab <= 0 ? g : 0.f;
            a1 += lab == 1 ? g : 0.f;
            a2 += lab >= 2 ? g : 0.f;
        }
        red[(rg * 3 + 0) * d + c] = a0;
        red[(rg * 3 + 1) * d + c] = a1;
        red[(rg * 3 + 2) * d + c] = a2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * d; i += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < groups; ++k) s += red[k * 3 * d + i];
        if (s != 0.f) atomicAdd(&dlabel[i], s);
    }
}

// deterministic variant of the label-row gradient: every block owns a fixed slice of the (sample, target/retrieved) rows, thread =
// (column, row group) accumulates its three label rows in registers, row groups are combined through LDS in a fixed order and
// the per-block partials [blocks][3][d] are summed in block order by a second launch: bit-reproducible, no atomics.
constexpr int LABEL_BLOCKS = 256;

__global__ void __launch_bounds__(GATHER_THREADS)
label_grad_partial_kernel(const float* __restrict__ dgrid, const int32_t* __restrict__ label_ids, float* __restrict__ part,
                          int64_t nbt, int S, int d) {
    RAT_DYN_SMEM(smem);
    float* red = reinterpret_cast<float*>(smem);              // [groups][3][d]
    const int groups = GATHER_THREADS / d;
    const int c = threadIdx.x % d, rg = threadIdx.x / d;
    const int64_t per = (nbt + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < nbt ? r0 + per : nbt;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (rg < groups)
        for (int64_t bt = r0 + rg; bt < r1; bt += groups) {
            const float g = dgrid[(bt * S) * d + c];
            const int lab = label_ids[bt];
            a0 += lab <= 0 ? g : 0.f;
            a1 += lab == 1 ? g : 0.f;
            a2 += lab >= 2 ? g : 0.f;
        }
    if (rg < groups) {
        red[(rg * 3 + 0) * d + c] = a0;
        red[(rg * 3 + 1) * d + c] = a1;
        red[(rg * 3 + 2) * d + c] = a2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * d; i += blockDim.x) {
        float sacc = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) sacc += red[g2 * 3 * d + i];
        part[(int64_t)blockIdx.x * 3 * d + i] = sacc;
    }
}

__global__ void __launch_bounds__(GATHER_THREADS)
label_grad_final_kernel(const float* __restrict__ part, float* __restrict__ dlabel, int nblocks, int d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * d) return;
    float sacc = 0.f;
    for (int b = 0; b < nblocks; ++b) sacc += part[(int64_t)b * 3 * d + i];
    dlabel[i] += sacc;
}

// nn.Embedding raises IndexError on an id outside [0, vocab); the kernels above clamp for memory safety, so the error would
// be silent.  This pass COUNTS the offenders instead (counts[0]: feature ids, counts[1]: label ids outside {0, 1, 2}); the
// host reads the two counters at its next natural synchronisation point and raises.
__global__ void __launch_bounds__(GATHER_THREADS)
check_ids_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ label_ids, const RatField* __restrict__ fields,
                 int nfields, int64_t nbt, int L, int* __restrict__ counts) {
    int bad_id = 0, bad_lab = 0;
    for (int64_t bt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; bt < nbt; bt += (int64_t)gridDim.x * blockDim.x) {
        const int lab = label_ids[bt];
        bad_lab += (lab < 0 || lab > 2) ? 1 : 0;
        for (int f = 0; f < nfields; ++f) {
            const RatField fd = fields[f];
            for (int j = 0; j < fd.ncols; ++j) {
                const int id = idx[bt * L + fd.col + j];
                bad_id += (id < 0 || id >= fd.vocab) ? 1 : 0;
            }
        }
    }
    if (bad_id) atomicAdd(&counts[0], bad_id);
    if (bad_lab) atomicAdd(&counts[1], bad_lab);
}

int pick_blocks(int64_t nitems, int per_thread) {
    int64_t want = (nitems + (int64_t)GATHER_THREADS * per_thread - 1) / ((int64_t)GATHER_THREADS * per_thread);
    if (want < 1) want = 1;
    if (want > 2048) want = 2048;                             // 256 CUs x 8 blocks, grid-stride the rest.  (A grid that gives every thread
    return (int)want;                                         //  the same number of trips — 1804 blocks instead of 2048 — measured 2-3 % slower.)
}

}  // namespace

extern "C" int rat_gather_fwd(const int32_t* idx, const int32_t* label_ids, const RatField* fields_dev, int nfields,
                              const float* label_table, float* grid, int B, int T, int L, int d, void* stream) {
    RAT_REQUIRE(B > 0 && T > 0 && L > 0 && d > 0 && nfields >= 0, "bad dims");
    RAT_REQUIRE(idx && label_ids && label_table && grid && (fields_dev || nfields == 0), "null pointer");
    const int S = nfields + 1;
    const int64_t nrows = (int64_t)B * T * S;
    const bool vec = (d % 4 == 0) && ((reinterpret_cast<uintptr_t>(grid) | reinterpret_cast<uintptr_t>(label_table)) % 16 == 0);
    if (vec) {
        if (d == 64 && nrows < (int64_t)0x7fffffff) {
            if (nrows * d * 4 > ((int64_t)160 << 20))
                RAT_LAUNCH(gather_fwd_rows64_kernel<true>, pick_blocks(nrows * (d / 4), GATHER_ITEMS), GATHER_THREADS, 0, stream, idx,
                           label_ids, fields_dev, label_table, grid, (unsigned)nrows, (unsigned)S, L);
            else
                RAT_LAUNCH(gather_fwd_rows64_kernel<false>, pick_blocks(nrows * (d / 4), GATHER_ITEMS), GATHER_THREADS, 0, stream, idx,
                           label_ids, fields_dev, label_table, grid, (unsigned)nrows, (unsigned)S, L);
        } else
            RAT_LAUNCH(gather_fwd_vec_kernel, pick_blocks(nrows * (d / 4), GATHER_ITEMS), GATHER_THREADS, 0, stream, idx,
                       label_ids, fields_dev, label_table, grid, nrows, S, L, d);
    } else {
        RAT_LAUNCH(gather_fwd_scalar_kernel, pick_blocks(nrows * d, 4), GATHER_THREADS, 0, stream, idx, label_ids,
                   fields_dev, label_table, grid, nrows, S, L, d);
    }
    return rat_check_launch("rat_gather_fwd");
}

extern "C" int rat_gather_bwd(const float* dgrid, const float* dflat, const int32_t* idx, const int32_t* label_ids,
                              const RatField* grad_fields_dev, int nfields, float* dlabel_table, int B, int T, int L,
                              int d, void* stream) {
    RAT_REQUIRE(B > 0 && T > 0 && L > 0 && d > 0 && nfields >= 0, "bad dims");
    RAT_REQUIRE(dgrid && idx && label_ids, "null pointer");
    const int S = nfields + 1;
    const int64_t nbt = (int64_t)B * T;
    if (nfields > 0) {
        RAT_REQUIRE(grad_fields_dev, "null grad field table");
        if (d == 64 && nbt * nfields < (int64_t)0x7fffffff)
            RAT_LAUNCH(gather_bwd_fields64_kernel, pick_blocks(nbt * nfields * d, GATHER_BWD_ROWS), GATHER_THREADS, 0, stream, dgrid, dflat,
                       idx, grad_fields_dev, (int)(nbt * nfields), T, S, L);
        else
            RAT_LAUNCH(gather_bwd_fields_kernel, pick_blocks(nbt * nfields * d, 4), GATHER_THREADS, 0, stream, dgrid, dflat,
                       idx, grad_fields_dev, nbt, T, S, L, d);
    }
    if (dlabel_table) {
        RAT_REQUIRE(d <= GATHER_THREADS, "embedding_dim above the block size");
        int64_t blocks = (nbt + 15) / 16;                      // at least 16 rows per work-group
        if (blocks > 256) blocks = 256;
        RAT_LAUNCH(gather_bwd_label_kernel, (unsigned)blocks, GATHER_THREADS, (size_t)(GATHER_THREADS / d) * 3 * d * sizeof(float), stream,
                   dgrid, label_ids, dlabel_table, nbt, S, d);
    }
    return rat_check_launch("rat_gather_bwd");
}

extern "C" int rat_check_ids(const int32_t* idx, const int32_t* label_ids, const RatField* fields_dev, int nfields, int B, int T,
                             int L, int32_t* counts, void* stream) {
    RAT_REQUIRE(B > 0 && T > 0 && L > 0 && nfields >= 0 && idx && label_ids && counts && (fields_dev || nfields == 0), "bad args");
    const int64_t nbt = (int64_t)B * T;
    RAT_LAUNCH(check_ids_kernel, pick_blocks(nbt, 1), GATHER_THREADS, 0, stream, idx, label_ids, fields_dev, nfields, nbt, L,
               reinterpret_cast<int*>(counts));
    return rat_check_launch("rat_check_ids");
}

extern "C" size_t rat_label_grad_workspace(int d) { return (size_t)LABEL_BLOCKS * 3 * (size_t)(d > 0 ? d : 1) * sizeof(float); }

extern "C" int rat_label_grad(const float* dgrid, const int32_t* label_ids, float* dlabel_table, float* workspace, int64_t nbt,
                              int S, int d, void* stream) {
    RAT_REQUIRE(dgrid && label_ids && dlabel_table && workspace && nbt > 0 && S > 0 && d > 0 && d <= GATHER_THREADS, "bad args");
    const int groups = GATHER_THREADS / d;
    RAT_LAUNCH(label_grad_partial_kernel, LABEL_BLOCKS, GATHER_THREADS, (size_t)groups * 3 * d * sizeof(float), stream, dgrid,
               label_ids, workspace, nbt, S, d);
    RAT_LAUNCH(label_grad_final_kernel, (3 * d + GATHER_THREADS - 1) / GATHER_THREADS, GATHER_THREADS, 0, stream, workspace,
               dlabel_table, LABEL_BLOCKS, d);
    return rat_check_launch("rat_label_grad");
}
