// retrieval.hip — K6: BM25-style top-K retrieval pre-compute.
//
// Replaces the scoring / top-k / merge core of BM25_topk_retrieval_v4 (fuxictr/datasets/data_utils.py:774-1064):
//     score[b, n] = sum_f (qry[b, f] == db[n, f]) * idf[b, f]          (data_utils.py:1003, float64)
//   with exact-match columns (rat_bm25_topk_grouped; data_utils.py:851-866, 932-938): only the pool rows of the query's GROUP (equal
//   on every exact-match column; the host numbers the groups) are candidates and a candidate scores (BM25 + 1):
//     score[b, n] = (grp_db[n] == grp_qry[b]) ? score[b, n] + 1 : 0
//     per query: the K largest scores, sorted descending, zero scores dropped (index -1), lens = number kept
//                                                                         (padded_topk + sort_results, data_utils.py:786-818)
// The reference materialises [qry_batch x db_chunk x F] tensors chunk by chunk, takes torch.topk per chunk and merges; here
// one work-group scans the whole pool for a tile of queries and keeps the running top-K in registers, so nothing but the
// id columns is ever read and nothing but the [Q, K] results is written.
//
// Integer / HBM-bound work: db ids are stored FIELD-MAJOR ([F][N] int32) so that a wave reads 256 contiguous bytes per
// field; the query ids and IDF weights of the tile are wave-uniform (scalar loads).  Each lane owns rows n = lane, lane +
// 256, ... (four rows per trip, so four independent loads per field are in flight) and inserts a row into its private sorted list only when its score is positive and beats the current K-th
// best (rare after the first few hundred rows).  The 256 private lists are merged by K rounds of a work-group arg-max.
// Ties: torch.topk leaves the order of equal scores unspecified (it differs between its CPU and CUDA kernels and with the
// chunk size); this kernel is deterministic — equal scores keep the LOWER pool index.
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

constexpr int RT_THREADS = 256;
constexpr int RT_FMAX = 32;

struct RetrArgs {
    const int32_t* db_t;     // [F][N]
    const int32_t* qry;      // [Q][F]
    const double* idf;       // [Q][F]
    const int32_t* db_grp;   // [N] exact-match group of every pool row, or null (no exact-match columns)
    const int32_t* qry_grp;  // [Q] group of every query (>= 0)
    double* out_val;         // [Q][K]
    int64_t* out_idx;        // [Q][K]
    int64_t* out_len;        // [Q]
    int64_t N, Q;
    int F, K;
};

__device__ __forceinline__ bool better(double sa, int64_t ia, double sb, int64_t ib) {   // (score desc, index asc)
    return sa > sb || (sa == sb && ia < ib);
}

template <int KMAX, int QT, int RU>
__global__ void __launch_bounds__(RT_THREADS) bm25_topk_kernel(RetrArgs a) {
    __shared__ double red_s[RT_THREADS];
    __shared__ int64_t red_i[RT_THREADS];
    const int tid = threadIdx.x;
    const int64_t ntiles = (a.Q + QT - 1) / QT;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t q0 = tile * QT;
        double val[QT][KMAX], kth[QT];                     // kth = score of the current K-th entry (0 while the list is not full)
        int64_t idx[QT][KMAX];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            kth[t] = 0.0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                val[t][k] = 0.0;
                idx[t][k] = -1;
            }
        }
        // ---- scan: every lane walks its rows in increasing order, so among equal scores the lower index arrives first and a
        //      later row must be STRICTLY better than the current K-th entry to get in
        for (int64_t n0 = tid; n0 < a.N; n0 += (int64_t)RT_THREADS * RU) {
            double s[RU][QT];
#pragma unroll
            for (int u = 0; u < RU; ++u)
#pragma unroll
                for (int t = 0; t < QT; ++t) s[u][t] = 0.0;
            for (int f = 0; f < a.F; ++f) {
                int32_t id[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {                                            // RU independent loads in flight per field
                    const int64_t n = n0 + (int64_t)u * RT_THREADS;
                    id[u] = n < a.N ? a.db_t[(int64_t)f * a.N + n] : -1;
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const int64_t q = q0 + t < a.Q ? q0 + t : a.Q - 1;                    // wave-uniform: scalar loads
                    const int32_t qid = a.qry[q * a.F + f];
                    const double w = a.idf[q * a.F + f];
#pragma unroll
                    for (int u = 0; u < RU; ++u) s[u][t] += (qid == id[u] && n0 + (int64_t)u * RT_THREADS < a.N) ? w : 0.0;
                }
            }
            if (a.db_grp) {                                                               // exact-match gate: (BM25 + 1) inside the group, 0 outside
                int32_t g[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int64_t n = n0 + (int64_t)u * RT_THREADS;
                    g[u] = n < a.N ? a.db_grp[n] : -1;
                }
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const int32_t qg = a.qry_grp[q0 + t < a.Q ? q0 + t : a.Q - 1];
#pragma unroll
                    for (int u = 0; u < RU; ++u) s[u][t] = g[u] == qg ? s[u][t] + 1.0 : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {                                                // rows in increasing order
                const int64_t n = n0 + (int64_t)u * RT_THREADS;
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    if (s[u][t] > kth[t]) {                                               // positive AND strictly better than the K-th
                        double cs = s[u][t];
                        int64_t ci = n;
                        bool placed = false;          // once the newcomer sits the rest only shifts down: a displaced entry is OLDER
                                                      // (lower index) than the equal scores below it and must stay in front of them
#pragma unroll
                        for (int k = 0; k < KMAX; ++k) {
                            if (k < a.K && (placed || cs > val[t][k])) {
                                placed = true;
                                const double ts = val[t][k];
                                const int64_t ti = idx[t][k];
                                val[t][k] = cs;
                                idx[t][k] = ci;
                                cs = ts;
                                ci = ti;
                            }
                            if (k == a.K - 1) kth[t] = val[t][k];
                        }
                    }
                }
            }
        }
        // ---- merge the 256 private lists: K rounds of a work-group arg-max over the lists' heads
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            int64_t kept = 0;
            for (int k = 0; k < a.K; ++k) {
                red_s[tid] = val[t][0];
                red_i[tid] = idx[t][0];
                __syncthreads();
                for (int step = RT_THREADS / 2; step > 0; step >>= 1) {
                    if (tid < step) {
                        const double so = red_s[tid + step];
                        const int64_t io = red_i[tid + step];
                        if (io >= 0 && (red_i[tid] < 0 || better(so, io, red_s[tid], red_i[tid]))) {
                            red_s[tid] = so;
                            red_i[tid] = io;
                        }
                    }
                    __syncthreads();
                }
                const double ws = red_s[0];
                const int64_t wi = red_i[0];
                __syncthreads();
                if (wi >= 0 && idx[t][0] == wi) {                                          // the winner pops its head
#pragma unroll
                    for (int j = 0; j + 1 < KMAX; ++j) {
                        val[t][j] = val[t][j + 1];
                        idx[t][j] = idx[t][j + 1];
                    }
                    val[t][KMAX - 1] = 0.0;
                    idx[t][KMAX - 1] = -1;
                }
                if (tid == 0 && q0 + t < a.Q) {
                    a.out_val[(q0 + t) * a.K + k] = wi >= 0 ? ws : 0.0;
                    a.out_idx[(q0 + t) * a.K + k] = wi;
                }
                kept += wi >= 0 ? 1 : 0;
            }
            if (tid == 0 && q0 + t < a.Q) a.out_len[q0 + t] = kept;
        }
    }
}

}  // namespace

static int bm25_topk_launch(const char* what, const int32_t* db_ids_field_major, const int32_t* db_groups, const int32_t* qry_ids,
                            const double* qry_idf, const int32_t* qry_groups, double* out_values, int64_t* out_indices,
                            int64_t* out_lens, int64_t n_db, int64_t n_qry, int n_fields, int topk, void* stream) {
    RAT_REQUIRE(db_ids_field_major && qry_ids && qry_idf && out_values && out_indices && out_lens, "null pointer");
    RAT_REQUIRE(n_db > 0 && n_qry > 0 && n_fields > 0 && topk > 0, "bad dims");
    RAT_REQUIRE(n_fields <= RT_FMAX, "more than 32 retrieval columns are not supported");
    RAT_REQUIRE(topk <= 32, "topK > 32 is not supported");
    RetrArgs a{};
    a.db_t = db_ids_field_major;
    a.qry = qry_ids;
    a.idf = qry_idf;
    a.db_grp = db_groups;
    a.qry_grp = qry_groups;
    a.out_val = out_values;
    a.out_idx = out_indices;
    a.out_len = out_lens;
    a.N = n_db;
    a.Q = n_qry;
    a.F = n_fields;
    a.K = topk;
    if (topk <= 8) {
        const int64_t tiles = (n_qry + 3) / 4;
        RAT_LAUNCH((bm25_topk_kernel<8, 4, 4>), (unsigned)(tiles < 65536 ? tiles : 65536), RT_THREADS, 0, stream, a);
    } else {
        RAT_LAUNCH((bm25_topk_kernel<32, 1, 4>), (unsigned)(n_qry < 65536 ? n_qry : 65536), RT_THREADS, 0, stream, a);
    }
    return rat_check_launch(what);
}

extern "C" int rat_bm25_topk(const int32_t* db_ids_field_major, const int32_t* qry_ids, const double* qry_idf, double* out_values,
                             int64_t* out_indices, int64_t* out_lens, int64_t n_db, int64_t n_qry, int n_fields, int topk,
                             void* stream) {
    return bm25_topk_launch("rat_bm25_topk", db_ids_field_major, nullptr, qry_ids, qry_idf, nullptr, out_values, out_indices, out_lens,
                            n_db, n_qry, n_fields, topk, stream);
}

extern "C" int rat_bm25_topk_grouped(const int32_t* db_ids_field_major, const int32_t* db_groups, const int32_t* qry_ids,
                                     const double* qry_idf, const int32_t* qry_groups, double* out_values, int64_t* out_indices,
                                     int64_t* out_lens, int64_t n_db, int64_t n_qry, int n_fields, int topk, void* stream) {
    RAT_REQUIRE(db_groups && qry_groups, "null group pointer");
    return bm25_topk_launch("rat_bm25_topk_grouped", db_ids_field_major, db_groups, qry_ids, qry_idf, qry_groups, out_values,
                            out_indices, out_lens, n_db, n_qry, n_fields, topk, stream);
}
