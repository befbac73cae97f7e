// batch.hip — device-side batch assembly: the reference's Dataset.__getitem__ + default_collate for retrieval-augmented
// samples (fuxictr/pytorch/data_generator.py:66-78, 239-241) as ONE gather over HBM-resident arrays.
//
// The reference builds every sample in Python inside DataLoader workers:
//     darray_i = concat(darray[index][None], retr_pool_darray[retr_indices[index]])     # [(1+K), L+1]
//     X_i, y_i = darray_i[..., :-1], darray_i[..., -1]
// then collates float64 tensors and copies them to the device.  Here the encoded query table, the retrieval pool and the
// pre-computed neighbour lists stay resident in HBM (int32 ids, fp32 labels) and a batch is one kernel over the row ids:
//     idx[b][0][:]   = data_ids[rows[b]]                  label_ids[b][0] = 2 (the target's [UNK] label token, RAT_m2.py:116)
//     idx[b][1+k][:] = pool_ids[nbr(b, k)]                label_ids[b][1+k] = (int) pool_labels[nbr(b, k)]
//     y_true[b]      = data_labels[rows[b]]
// nbr(b, k) = retr_indices[rows[b]][k], a NEGATIVE entry counting from the end of the pool exactly like the numpy fancy
// index of the reference does (its -1 padding of short neighbour lists therefore selects the last pool row).
#include "rat_device.h"
#include "../../include/rat_hip.h"

namespace {

struct BatchArgs {
    const int32_t* data_ids;
    const float* data_labels;
    const int32_t* pool_ids;
    const float* pool_labels;
    const int64_t* retr_indices;
    const int64_t* rows;
    int32_t* idx;
    int32_t* label_ids;
    float* y_true;
    int64_t Q, N;
    int B, K, L;
};

// one thread per output id: consecutive threads write consecutive ints of idx (coalesced); the source rows are L ints
// (80 bytes at the north-star config) fetched by L neighbouring lanes
__global__ void __launch_bounds__(256) batch_assemble_kernel(BatchArgs a) {
    const int T = a.K + 1;
    const int64_t total = (int64_t)a.B * T * a.L;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % a.L);
        const int64_t bt = e / a.L;
        const int t = (int)(bt % T);
        const int b = (int)(bt / T);
        const int64_t row = a.rows[b];
        int32_t v;
        if (t == 0) {
            v = a.data_ids[row * a.L + c];
            if (c == 0) {
                a.label_ids[bt] = 2;
                a.y_true[b] = a.data_labels[row];
            }
        } else {
            int64_t nb = a.retr_indices[row * a.K + (t - 1)];
            if (nb < 0) nb += a.N;
            v = a.pool_ids[nb * a.L + c];
            if (c == 0) a.label_ids[bt] = (int32_t)a.pool_labels[nb];
        }
        a.idx[e] = v;
    }
}

}  // namespace

extern "C" int rat_batch_assemble(const int32_t* data_ids, const float* data_labels, const int32_t* pool_ids,
                                  const float* pool_labels, const int64_t* retr_indices, const int64_t* rows, int32_t* idx,
                                  int32_t* label_ids, float* y_true, int64_t Q, int64_t N, int B, int K, int L, void* stream) {
    RAT_REQUIRE(data_ids && data_labels && pool_ids && pool_labels && rows && idx && label_ids && y_true, "null pointer");
    RAT_REQUIRE(K == 0 || retr_indices != nullptr, "null retr_indices");
    RAT_REQUIRE(B > 0 && K >= 0 && L > 0 && Q > 0 && N > 0, "bad dims");
    BatchArgs a{data_ids, data_labels, pool_ids, pool_labels, retr_indices, rows, idx, label_ids, y_true, Q, N, B, K, L};
    const int64_t total = (int64_t)B * (K + 1) * L;
    const int64_t blocks = (total + 255) / 256;
    RAT_LAUNCH(batch_assemble_kernel, (unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, stream, a);
    return rat_check_launch("rat_batch_assemble");
}

// ---- inputs_to_device + the label-token rule for a batch that is ALREADY on the device (ABI v8) --------------------------------------
// base_model.py:125-133 (X stays what it is, y.float()) and RAT_m2.py:110-118 (the target's label token is id 2, a retrieved sample's its
// label): X [B][T][L] of any of four element types -> idx int32; y [B][T] -> label_ids int32 (column 0: 2) and y_true fp32 (= y[:, 0]).
// One launch instead of the five ATen launches the same conversion costs (to(int32), clone, index fill, slice copy, to(float32)), and it
// writes straight into the static input tensors of a captured step.
namespace {
template <class TX, class TY>
__global__ void __launch_bounds__(256)
batch_prepare_kernel(const TX* __restrict__ X, const TY* __restrict__ y, int32_t* __restrict__ idx, int32_t* __restrict__ label_ids,
                     float* __restrict__ y_true, int64_t n_ids, int64_t n_lab, int T) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ids; i += (int64_t)gridDim.x * blockDim.x) {
        idx[i] = (int32_t)X[i];
        if (i < n_lab) {
            const TY v = y[i];
            const bool target = (i % T) == 0;
            label_ids[i] = target ? 2 : (int32_t)v;
            if (target) y_true[i / T] = (float)v;
        }
    }
}
template <class TX>
int prepare_y(int y_type, const void* X, const void* y, int32_t* idx, int32_t* label_ids, float* y_true, int64_t n_ids, int64_t n_lab,
              int T, void* stream) {
    const int64_t n = n_ids > n_lab ? n_ids : n_lab;
    const int64_t blocks = (n + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
    if (y_type == RAT_DTYPE_F32)
        RAT_LAUNCH((batch_prepare_kernel<TX, float>), grid, 256, 0, stream, static_cast<const TX*>(X), static_cast<const float*>(y), idx,
                   label_ids, y_true, n_ids, n_lab, T);
    else if (y_type == RAT_DTYPE_F64)
        RAT_LAUNCH((batch_prepare_kernel<TX, double>), grid, 256, 0, stream, static_cast<const TX*>(X), static_cast<const double*>(y), idx,
                   label_ids, y_true, n_ids, n_lab, T);
    else
        return rat_fail("rat_batch_prepare: y must be fp32 or fp64");
    return rat_check_launch("rat_batch_prepare");
}
}  // namespace

extern "C" int rat_batch_prepare(const void* X, int x_type, const void* y, int y_type, int32_t* idx, int32_t* label_ids, float* y_true,
                                 int B, int T, int L, void* stream) {
    RAT_REQUIRE(X && y && idx && label_ids && y_true, "null pointer");
    RAT_REQUIRE(B > 0 && T > 0 && L > 0, "bad dims");
    const int64_t n_ids = (int64_t)B * T * L, n_lab = (int64_t)B * T;          // (L >= 1: n_ids >= n_lab, every label index is visited)
    switch (x_type) {
        case RAT_DTYPE_I32: return prepare_y<int32_t>(y_type, X, y, idx, label_ids, y_true, n_ids, n_lab, T, stream);
        case RAT_DTYPE_I64: return prepare_y<int64_t>(y_type, X, y, idx, label_ids, y_true, n_ids, n_lab, T, stream);
        case RAT_DTYPE_F32: return prepare_y<float>(y_type, X, y, idx, label_ids, y_true, n_ids, n_lab, T, stream);
        case RAT_DTYPE_F64: return prepare_y<double>(y_type, X, y, idx, label_ids, y_true, n_ids, n_lab, T, stream);
        default: return rat_fail("rat_batch_prepare: X must be int32, int64, fp32 or fp64");
    }
}
