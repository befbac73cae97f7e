"""Top-K retrieval pre-compute on the device — the drop-in for ``BM25_topk_retrieval_v4``
(fuxictr/datasets/data_utils.py:774-1064), which the reference's DataGenerator calls once per split to build
``retrieval_{K}_{split}.h5`` (fuxictr/pytorch/data_generator.py:114-215).

Same signature, same ``(values, indices, lens)`` named tuple of numpy arrays.  The host side keeps only what is host work in
the reference too (the per-column IDF table: value counts of the pool, data_utils.py:873-880, and mapping the query ids to
their weights, :843-847); scoring, top-k and the merge over the pool run in ONE kernel (rat_bm25_topk, csrc/retrieval.hip)
that scans the pool once per tile of four queries — ``db_chunk_size`` is accepted and ignored (nothing is materialised),
``qry_batch_size`` bounds the result buffers per launch.

``exact_match_col_indices`` (data_utils.py:851-866; the shipped configs use ``exact_match_cols: []``): the pool rows equal to the
query on those columns are its only candidates.  The host numbers the distinct keys (one ``np.unique`` over pool and queries — the
reference's pandas group-by) and, as the reference does batch by batch, either lists the members directly when no group of the
batch is larger than topK (value 1.0, pool order, no scoring: data_utils.py:911-917) or scores ``BM25 + 1`` inside the group —
on the device that is the same single pool scan with one more compare per row (``rat_bm25_topk_grouped``), not the
reference's padded ``[B, E]`` member lists.
"""
import ctypes
from collections import namedtuple

import numpy as np
import torch

from ._lib import get_lib

ResultsNamedTuple = namedtuple("ResultsNameTuple", ["values", "indices", "lens"])


def idf_tables(db_np_data):
    """per column: (sorted distinct ids, log(N / count)) — data_utils.py:873-880."""
    n = len(db_np_data)
    tables = []
    for c in range(db_np_data.shape[1]):
        vals, counts = np.unique(db_np_data[:, c], return_counts=True)
        tables.append((vals, np.log(n / counts)))
    return tables


def map_data_to_idf(np_data, tables):
    """ids of ONE query batch -> IDF weight, 0 for ids the pool column does not hold (map_data_to_IDF_v1,
    data_utils.py:843-847).  Kept bug-compatible: the reference's ``np.vectorize(lambda x: stats.get(x, 0))`` takes its
    output dtype from the first element, so a batch whose FIRST row carries an unseen id in a column gets that whole column
    as int64 — every weight truncated toward zero.  The published ``retrieval_*.h5`` files embody this."""
    out = np.zeros(np_data.shape, dtype=np.float64)
    for c, (vals, idf) in enumerate(tables):
        pos = np.minimum(np.searchsorted(vals, np_data[:, c]), len(vals) - 1)
        hit = vals[pos] == np_data[:, c]
        col = np.where(hit, idf[pos], 0.0)
        if len(np_data) and not hit[0]:
            col = col.astype(np.int64).astype(np.float64)
        out[:, c] = col
    return out


def _as_int32(a, what):
    a = np.asarray(a)
    if a.size and (a.min() < np.iinfo(np.int32).min or a.max() > np.iinfo(np.int32).max):
        raise ValueError("%s ids do not fit int32" % what)
    return a.astype(np.int32)


class ExactMatchGroups:
    """The reference's ``db_df.groupby(cols).groups`` + ``get_indexer`` (data_utils.py:851-859) as flat arrays: a code per distinct
    key of the exact-match columns, the pool rows sorted by code (stable: ascending row inside a group), and per query the code of
    its key or -1 when no pool row carries it."""

    def __init__(self, db_keys, qry_keys):
        n_db = len(db_keys)
        _, inv = np.unique(np.concatenate([db_keys, qry_keys], axis=0), axis=0, return_inverse=True)
        inv = inv.reshape(-1).astype(np.int64)
        self.db_code = inv[:n_db]
        self.size = np.bincount(self.db_code, minlength=int(inv.max()) + 1 if len(inv) else 0)
        self.start = np.concatenate([[0], np.cumsum(self.size)[:-1]]).astype(np.int64)
        self.order = np.argsort(self.db_code, kind="stable")
        q_code = inv[n_db:]
        self.qry_code = np.where(self.size[q_code] > 0, q_code, -1) if len(q_code) else q_code

    def members(self, codes, width, last):
        """[len(codes), width] pool rows of each group in ascending order, -1 padded; a group larger than `width` keeps its LAST
        `width` rows when `last` (pad_sequences' default truncating='pre', data_utils.py:903-905)"""
        cnt = np.minimum(self.size[codes], width)
        first = self.start[codes] + (self.size[codes] - cnt if last else 0)
        pos = first[:, None] + np.arange(width)[None, :]
        keep = np.arange(width)[None, :] < cnt[:, None]
        return np.where(keep, self.order[np.minimum(pos, len(self.order) - 1)], -1), cnt


def BM25_topk_retrieval_v4(db_np_data, qry_np_data, exact_match_col_indices=None, qry_batch_size=None, db_chunk_size=None,
                           device="cuda:0", topK=10, enable_clean=False, lib=None, **kwargs):
    db_np_data, qry_np_data = np.asarray(db_np_data), np.asarray(qry_np_data)
    assert db_np_data.ndim == 2 and qry_np_data.ndim == 2 and db_np_data.shape[1] == qry_np_data.shape[1]
    groups = None
    if exact_match_col_indices:
        exm = np.zeros(db_np_data.shape[1], dtype=bool)
        exm[list(exact_match_col_indices)] = True
        groups = ExactMatchGroups(db_np_data[:, exm], qry_np_data[:, exm])
        db_np_data, qry_np_data = db_np_data[:, ~exm], qry_np_data[:, ~exm]           # BM25 over the remaining columns (:860-862)
    lib = lib or get_lib()
    dev = torch.device(device)
    n_db, nf = db_np_data.shape
    n_qry = len(qry_np_data)
    values = np.zeros((n_qry, topK), dtype=np.float64)
    indices = np.full((n_qry, topK), -1, dtype=np.int64)
    lens = np.zeros(n_qry, dtype=np.int64)
    if n_qry == 0 or n_db == 0:
        return ResultsNamedTuple(values, indices, lens)
    assert groups is not None or nf > 0, "detected empty query tensor input"          # data_utils.py:1040
    tables = idf_tables(db_np_data)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
    db_t = db_g = None                                                                # uploaded when the first batch needs scoring
    step = n_qry if qry_batch_size is None else int(qry_batch_size)
    for q0 in range(0, n_qry, step):
        rows = np.arange(q0, min(q0 + step, n_qry))
        if groups is not None:
            rows = rows[groups.qry_code[rows] != -1]                                  # no candidate at all: (0, -1, len 0) stays
            if len(rows) == 0:
                continue
            codes = groups.qry_code[rows]
            if nf == 0 or groups.size[codes].max() <= topK:                           # every group of the batch fits: no scoring
                idx, cnt = groups.members(codes, topK, last=True)
                indices[rows], lens[rows], values[rows] = idx, cnt, (idx != -1).astype(np.float64)
                continue
        q_np = qry_np_data[rows]
        if db_t is None:
            db_t = torch.from_numpy(np.ascontiguousarray(_as_int32(db_np_data, "pool").T)).to(dev)      # [F][N] field-major
            if groups is not None:
                db_g = torch.from_numpy(_as_int32(groups.db_code, "group")).to(dev)
        q_ids = torch.from_numpy(_as_int32(q_np, "query")).contiguous().to(dev)
        q_idf = torch.from_numpy(map_data_to_idf(q_np, tables)).contiguous().to(dev)
        b = len(q_np)
        out_v = torch.empty((b, topK), dtype=torch.float64, device=dev)
        out_i = torch.empty((b, topK), dtype=torch.int64, device=dev)
        out_l = torch.empty((b,), dtype=torch.int64, device=dev)
        outs = (ctypes.c_void_p(out_v.data_ptr()), ctypes.c_void_p(out_i.data_ptr()), ctypes.c_void_p(out_l.data_ptr()))
        if groups is None:
            lib.call("rat_bm25_topk", ctypes.c_void_p(db_t.data_ptr()), ctypes.c_void_p(q_ids.data_ptr()),
                     ctypes.c_void_p(q_idf.data_ptr()), *outs, n_db, b, nf, int(topK), stream)
        else:
            q_g = torch.from_numpy(_as_int32(groups.qry_code[rows], "group")).to(dev)
            lib.call("rat_bm25_topk_grouped", ctypes.c_void_p(db_t.data_ptr()), ctypes.c_void_p(db_g.data_ptr()),
                     ctypes.c_void_p(q_ids.data_ptr()), ctypes.c_void_p(q_idf.data_ptr()), ctypes.c_void_p(q_g.data_ptr()), *outs,
                     n_db, b, nf, int(topK), stream)
        values[rows] = out_v.cpu().numpy()
        indices[rows] = out_i.cpu().numpy()
        lens[rows] = out_l.cpu().numpy()
    return ResultsNamedTuple(values, indices, lens)


# ------------------------------------------------------------------------------------------------------------------
def used_col_indices(feature_map, retrieval_configs):
    """h5_generator (fuxictr/datasets/data_utils.py:1193-1205): columns of the encoded array the retrieval compares."""
    return [feature_map.feature_specs[col]["index"] for col in retrieval_configs["used_cols"]]


def exact_match_col_indices(retrieval_configs):
    """h5_generator (data_utils.py:1199-1205): positions of ``exact_match_cols`` INSIDE ``used_cols``, None when there are none."""
    cols = retrieval_configs.get("exact_match_cols") or []
    return [retrieval_configs["used_cols"].index(c) for c in cols] or None


def precompute_retrieval(data_array, retrieval_configs, col_indices, pool_array=None, device="cuda:0", lib=None):
    """What DataGenerator does when ``retrieval_{K}_{split}.h5`` does not exist yet (fuxictr/pytorch/data_generator.py:114-212).

    data_array: the encoded split [Q, L+1] (label last).  pool_array None = pool "self": ``<X>-fold`` retrieval — every fold of
    the split queries the other folds (data_generator.py:115-176); otherwise the split queries the separate pool
    (:177-212).  ``label_wise`` retrieves top-K among the positives and top-K among the negatives (indices / values [Q, 2K],
    lens [Q, 2]).  Returns (indices, values, lens) exactly as the reference stores them — including its index arithmetic on
    padded entries: in the fold / label-wise paths a ``-1`` result indexes the LAST element of the index map
    (``fold_db_indices[-1]``), which is what the reference writes to disk."""
    import re
    exm = retrieval_configs.get("exact_match_col_indices")
    if exm is None and retrieval_configs.get("exact_match_cols"):
        exm = exact_match_col_indices(retrieval_configs)
    kw = dict(qry_batch_size=retrieval_configs.get("qry_batch_size"), topK=retrieval_configs["topK"], device=device, lib=lib,
              exact_match_col_indices=exm)
    label_wise = bool(retrieval_configs.get("label_wise", False))

    def retrieve(db, qry):
        return BM25_topk_retrieval_v4(db_np_data=db, qry_np_data=qry, **kw)

    def one(db_data, db_labels, qry_data, index_map):
        """-> (indices, values, lens) of one query set against one pool; index_map translates pool rows to global rows"""
        if label_wise:
            parts = []
            for sel in (np.nonzero(db_labels)[0], np.nonzero(1 - db_labels)[0]):
                r = retrieve(db_data[sel], qry_data)
                idx = sel[r.indices]
                parts.append((idx if index_map is None else index_map[idx], r.values, r.lens))
            return (np.concatenate([parts[0][0], parts[1][0]], axis=-1), np.concatenate([parts[0][1], parts[1][1]], axis=-1),
                    np.stack([parts[0][2], parts[1][2]], axis=-1))
        r = retrieve(db_data, qry_data)
        return (r.indices if index_map is None else index_map[r.indices]), r.values, r.lens

    if pool_array is None:
        ids = data_array[:, col_indices].astype(int)
        labels = data_array[:, -1].astype(int) if label_wise else None
        m = re.match(r"\d+-fold", retrieval_configs["split_type"])
        assert m is not None, "pool 'self' needs split_type '<X>-fold'"
        fold_num = int(m.group().split("-")[0])
        fold_size = int(np.ceil(len(ids) / fold_num))
        out = []
        for fi in range(fold_num):
            lo, hi = fi * fold_size, (fi + 1) * fold_size
            qry = ids[lo:hi]
            if len(qry) == 0:
                continue
            db = np.concatenate([ids[:lo], ids[hi:]], axis=0)
            index_map = np.concatenate([np.arange(lo), np.arange(min(hi, len(ids)), len(ids))], axis=0)
            db_labels = np.concatenate([labels[:lo], labels[hi:]], axis=0) if label_wise else None
            out.append(one(db, db_labels, qry, index_map))
        return tuple(np.concatenate([o[j] for o in out]) for j in range(3))
    db = pool_array[:, col_indices].astype(int)
    qry = data_array[:, col_indices].astype(int)
    return one(db, pool_array[:, -1].astype(int) if label_wise else None, qry, None)
