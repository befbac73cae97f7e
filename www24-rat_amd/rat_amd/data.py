"""Batch source for the plugin: the 4-tuple the reference's DataLoader hands to ``forward``
(fuxictr/pytorch/data_generator.py:66-78,239-241), assembled a whole batch at a time.

The reference builds every sample in a Python ``__getitem__`` (``pool[retr_indices[i]]`` + concat) inside 3 worker
processes and collates float64 tensors; this class does the same gather with one vectorised fancy-index per batch and
hands over int32 ids / float32 labels (the model converts once anyway).  On-disk formats: the reference's ``*.h5``
(``data`` = float [N, L+1] with the label last; ``retrieval_{K}_{split}.h5`` with ``indices``/``values``/``lens``) are
read through ``h5py`` when it is importable and otherwise through the HDF5 C library itself (``rat_amd/h5io.py``, ctypes over
``libhdf5``); ``.npz`` files with the same keys are always accepted.  A missing retrieval file is
computed on the device by ``rat_amd.retrieval.precompute_retrieval`` (run_expid.py), like the reference's DataGenerator does
with BM25_topk_retrieval_v4."""
import os

import numpy as np
import torch


def load_array_file(path, keys):
    if path.endswith(".npz"):
        blob = np.load(path)
        return {k: blob[k] for k in keys}
    try:
        import h5py
    except ImportError:
        # no h5py (this image): the HDF5 C library through ctypes (rat_amd/h5io.py) — same files, same keys
        from . import h5io
        try:
            return h5io.read_arrays(path, keys)
        except h5io.Hdf5Unavailable as exc:
            raise RuntimeError("%s is an HDF5 file and neither h5py nor libhdf5 is available (%s); export it to .npz with keys %s"
                               % (path, exc, keys)) from exc
    with h5py.File(path, "r") as hf:
        return {k: hf[k][:] for k in keys}


class RetrievalBatches:
    """Iterable over batches of (X [B,1+K,L], y [B,1+K], retrieved_values [B,K], retrieved_lens [B])."""

    def __init__(self, data, pool, retr_indices, retr_values, retr_lens, batch_size, shuffle=False, seed=0,
                 drop_last=False, shard=(0, 1)):
        """shard = (rank, world): data parallelism over the batch dimension — `batch_size` stays the GLOBAL batch,
        every rank walks the same permutation (same seed) and takes rows [rank*per, (rank+1)*per) of each global batch,
        per = len(batch) // world (a tail batch's len % world left-over samples are dropped so that all ranks hold equal shards and
        the mean-of-means of the loss stays the global mean)."""
        self.shard = (int(shard[0]), int(shard[1]))
        data, pool = np.asarray(data), np.asarray(pool)
        self.ids = np.ascontiguousarray(data[:, :-1].astype(np.int32))
        self.labels = np.ascontiguousarray(data[:, -1].astype(np.float32))
        same = pool is data
        self.pool_ids = self.ids if same else np.ascontiguousarray(pool[:, :-1].astype(np.int32))
        self.pool_labels = self.labels if same else np.ascontiguousarray(pool[:, -1].astype(np.float32))
        self.retr_indices = np.asarray(retr_indices).astype(np.int64)     # -1 selects the last pool row, like numpy does
        self.retr_values = np.asarray(retr_values, dtype=np.float32)
        self.retr_lens = np.asarray(retr_lens, dtype=np.int64)
        assert len(self.ids) == len(self.retr_indices) == len(self.retr_values) == len(self.retr_lens)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self._rng = np.random.RandomState(seed)

    def __len__(self):
        n = len(self.ids)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def to_device(self, device, lib=None):
        """The same dataset, HBM-resident: batches are then assembled by ``rat_batch_assemble`` (same order for the same seed)."""
        # the check RAT_m2.forward makes on every host batch (RAT_m2.py:109 `retrieved_lens.ndim == 1`): a label-wise retrieval
        # file ([Q, 2] lens, [Q, 2K] indices) must be rejected here too, the device batches no longer carry the lens
        assert self.retr_lens.ndim == 1, "RIM does not support label-wise retrieval-enhanced training"
        src = DeviceRetrievalBatches.__new__(DeviceRetrievalBatches)
        src._init_from_arrays(self.ids, self.labels, self.pool_ids, self.pool_labels, self.retr_indices, self.batch_size, device,
                              self.shuffle, self.drop_last, lib)
        src.shard = self.shard
        src._rng = np.random.RandomState()
        src._rng.set_state(self._rng.get_state())
        return src

    def __iter__(self):
        n = len(self.ids)
        order = self._rng.permutation(n) if self.shuffle else np.arange(n)
        for b in range(len(self)):
            rows = shard_rows(order[b * self.batch_size:(b + 1) * self.batch_size], self.shard, getattr(self, "keep_all", False))
            if len(rows) == 0:
                continue
            ridx = self.retr_indices[rows]                                              # [B, K]
            X = np.concatenate([self.ids[rows][:, None, :], self.pool_ids[ridx]], axis=1)
            y = np.concatenate([self.labels[rows][:, None], self.pool_labels[ridx]], axis=1)
            yield (torch.from_numpy(X), torch.from_numpy(y), torch.from_numpy(self.retr_values[rows]),
                   torch.from_numpy(self.retr_lens[rows]))


def shard_rows(rows, shard, keep_all=False):
    """rows of one GLOBAL batch -> this rank's equal share (see RetrievalBatches.__init__).  keep_all (evaluation): nothing is dropped —
    rank r takes rows [r c, (r + 1) c) with c = ceil(len / world), the last ranks may get fewer (or none)."""
    rank, world = shard
    if world == 1:
        return rows
    per = -(-len(rows) // world) if keep_all else len(rows) // world
    return rows[rank * per:(rank + 1) * per]


class DeviceBatch:
    """A batch already assembled on the device (ids int32 [B,1+K,L], label-token ids int32 [B,1+K], y_true fp32 [B]):
    ``RAT_m2`` consumes it as is — no host tensors, no float64 -> int32 conversion, no H2D copy."""

    __slots__ = ("idx", "label_ids", "y_true")

    def __init__(self, idx, label_ids, y_true):
        self.idx, self.label_ids, self.y_true = idx, label_ids, y_true

    def __len__(self):
        return int(self.idx.shape[0])


class DeviceRetrievalBatches:
    """HBM-resident replacement of the reference's Dataset + DataLoader for retrieval-augmented training
    (fuxictr/pytorch/data_generator.py:66-78, 239-241): the encoded query table, the retrieval pool and the neighbour
    lists are uploaded ONCE (int32 ids, fp32 labels, int64 neighbour indices); every batch is one ``rat_batch_assemble``
    launch over the batch's row ids.  Yields ``DeviceBatch`` objects.  Same ordering rule as ``RetrievalBatches`` (a numpy
    permutation per epoch when ``shuffle``), so both sources produce identical batches for the same seed."""

    def __init__(self, data, pool, retr_indices, batch_size, device, shuffle=False, seed=0, drop_last=False, lib=None,
                 retr_lens=None, shard=(0, 1)):
        self.shard = (int(shard[0]), int(shard[1]))         # (rank, world): see RetrievalBatches
        if retr_lens is not None:                       # same rejection of label-wise retrieval files as the host path
            assert np.asarray(retr_lens).ndim == 1, "RIM does not support label-wise retrieval-enhanced training"
        data, pool_arr = np.asarray(data), np.asarray(pool)
        same = pool is data
        ids, labels = data[:, :-1].astype(np.int32), data[:, -1].astype(np.float32)
        self._init_from_arrays(ids, labels, ids if same else pool_arr[:, :-1].astype(np.int32),
                               labels if same else pool_arr[:, -1].astype(np.float32), np.asarray(retr_indices).astype(np.int64),
                               batch_size, device, shuffle, drop_last, lib)
        self._rng = np.random.RandomState(seed)

    def _init_from_arrays(self, ids, labels, pool_ids, pool_labels, retr_indices, batch_size, device, shuffle, drop_last, lib):
        from . import ops
        self._ops, self._lib = ops, lib
        dev = torch.device(device)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.data_ids, self.data_labels = up(ids), up(labels)
        self.pool_ids = self.data_ids if pool_ids is ids else up(pool_ids)
        self.pool_labels = self.data_labels if pool_labels is labels else up(pool_labels)
        self.retr_indices = up(retr_indices)
        assert self.retr_indices.shape[0] == self.data_ids.shape[0]
        self.n = int(self.data_ids.shape[0])
        self.batch_size, self.shuffle, self.drop_last, self.device = int(batch_size), shuffle, drop_last, dev

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def assemble(self, rows):
        """rows: int64 tensor (device) of query-row ids -> DeviceBatch"""
        idx, label_ids, y_true = self._ops.batch_assemble(self.data_ids, self.data_labels, self.pool_ids, self.pool_labels,
                                                          self.retr_indices, rows.contiguous(), lib=self._lib)
        return DeviceBatch(idx, label_ids, y_true)

    def __iter__(self):
        order = self._rng.permutation(self.n) if self.shuffle else np.arange(self.n)
        order_dev = torch.from_numpy(order.astype(np.int64)).to(self.device)      # one small upload per epoch
        for b in range(len(self)):
            rows = shard_rows(order_dev[b * self.batch_size:(b + 1) * self.batch_size], getattr(self, "shard", (0, 1)),
                              getattr(self, "keep_all", False))
            if len(rows) == 0:
                continue
            yield self.assemble(rows)


def batches_from_files(data_path, retrieval_path, batch_size, pool_path=None, **kw):
    data = load_array_file(data_path, ["data"])["data"]
    pool = data if pool_path is None else load_array_file(pool_path, ["data"])["data"]
    r = load_array_file(retrieval_path, ["indices", "values", "lens"])
    return RetrievalBatches(data, pool, r["indices"], r["values"], r["lens"], batch_size, **kw)


def synthetic_split(feature_map, n_rows, topk, seed, label_rule=True):
    """A self-contained synthetic dataset with learnable structure (labels depend on the ids), for end-to-end runs
    when no real data is present: returns (data [N, L+1] float64, retr_indices, retr_values, retr_lens)."""
    rs = np.random.RandomState(seed)
    cols, score = [], np.zeros(n_rows)
    for j, spec in enumerate(feature_map.feature_specs.values()):
        ncols = len(spec["index"]) if isinstance(spec["index"], (list, tuple)) else 1
        v = spec["vocab_size"]
        ids = rs.randint(0, v - 1 if ncols > 1 else v, size=(n_rows, ncols))
        wts = np.random.RandomState(1000 + j).standard_normal(v)
        score += wts[ids].sum(axis=1)
        cols.append(ids)
    X = np.concatenate(cols, axis=1).astype(np.float64)
    prob = 1.0 / (1.0 + np.exp(-score / np.sqrt(len(cols)))) if label_rule else np.full(n_rows, 0.5)
    y = (rs.rand(n_rows) < prob).astype(np.float64)
    data = np.concatenate([X, y[:, None]], axis=1)
    # stand-in for BM25: neighbours that share the first column's id when possible, else random rows
    order = np.argsort(X[:, 0], kind="mergesort")
    pos = np.empty(n_rows, dtype=np.int64)
    pos[order] = np.arange(n_rows)
    offs = np.arange(1, topk + 1)[None, :]
    indices = order[(pos[:, None] + offs) % n_rows]
    values = rs.rand(n_rows, topk)
    lens = np.full(n_rows, topk, dtype=np.int64)
    return data, indices, values, lens
