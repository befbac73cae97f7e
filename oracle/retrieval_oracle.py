"""CPU oracle for the BM25-style top-K retrieval pre-compute.  TEST INFRASTRUCTURE ONLY (imported by tests/ only).

A plain-numpy restatement of what BM25_topk_retrieval_v4 (fuxictr/datasets/data_utils.py:774-1064) computes on the
``exact_match_col_indices`` = None/[] path of the shipped dataset configs (configs/datasets/movielenslatest_x1.yaml:58-75).

Parity status: PINNED.  ``tests/golden/make_golden_retrieval.py`` runs the real reference function (CPU device, chunked and
unchunked) on seeded inputs and commits its outputs as ``tests/golden/retrieval.npz``; ``tests/test_retrieval.py`` checks this
file against them.  One documented freedom: the reference takes ``torch.topk`` per database chunk and again over the merged
candidates, and torch leaves the order of EQUAL scores unspecified — so indices are compared up to ties
(``assert_topk_equivalent``): every returned (value, index) pair must be a true pair of the score matrix, the value rows
must be identical, and wherever a value is unique in its query's score row the index must match exactly.
"""
import numpy as np


def idf_tables(db):
    """Per pool column: (sorted distinct ids, log(N / count)) — data_utils.py:873-880 (value_counts + np.log(N / counts))."""
    n = len(db)
    out = []
    for c in range(db.shape[1]):
        vals, counts = np.unique(db[:, c], return_counts=True)
        out.append((vals, np.log(n / counts)))
    return out


def map_idf(qry, tables):
    """ONE query batch's ids -> their IDF weight, 0 for ids the pool column does not contain (map_data_to_IDF_v1,
    data_utils.py:843-847).  The reference maps with ``np.vectorize(lambda x: stats.get(x, 0))``, and np.vectorize takes
    its output dtype from the FIRST element: when the first row of the batch holds an id the pool has never seen, the lambda
    returns the int 0, the whole column becomes int64 and every weight in it is truncated toward zero.  Restated as is —
    the published retrieval files were produced that way."""
    out = np.zeros(qry.shape, dtype=np.float64)
    for c, (vals, idf) in enumerate(tables):
        pos = np.searchsorted(vals, qry[:, c])
        pos_c = np.minimum(pos, len(vals) - 1)
        hit = vals[pos_c] == qry[:, c]
        col = np.where(hit, idf[pos_c], 0.0)
        if len(qry) and not hit[0]:
            col = col.astype(np.int64).astype(np.float64)
        out[:, c] = col
    return out


def scores(db, qry, qry_batch_size=None):
    """[Q, N] float64: sum over columns (ascending) of (qry == db) * idf(qry) — data_utils.py:1003; the IDF mapping runs per
    query batch (data_utils.py:893,926), which matters because of the dtype rule in map_idf."""
    tables = idf_tables(db)
    step = len(qry) if qry_batch_size is None else qry_batch_size
    w = np.concatenate([map_idf(qry[i:i + step], tables) for i in range(0, len(qry), step)], axis=0) if len(qry) else np.zeros(qry.shape)
    s = np.zeros((len(qry), len(db)), dtype=np.float64)
    for f in range(db.shape[1]):
        s += (qry[:, f][:, None] == db[:, f][None, :]) * w[:, f][:, None]
    return s


def topk(db, qry, k, qry_batch_size=None):
    """(values [Q,K] f64 descending, indices [Q,K] i64, lens [Q] i64); zero scores dropped (index -1, value 0), equal scores
    in ascending pool-index order — padded_topk + sort_results (data_utils.py:786-818) with a deterministic tie order."""
    s = scores(db, qry, qry_batch_size)
    q, n = s.shape
    values = np.zeros((q, k), dtype=np.float64)
    indices = np.full((q, k), -1, dtype=np.int64)
    lens = np.zeros(q, dtype=np.int64)
    for b in range(q):
        order = np.lexsort((np.arange(n), -s[b]))            # primary: score descending, secondary: index ascending
        keep = [j for j in order[:k] if s[b, j] > 0]
        lens[b] = len(keep)
        values[b, :len(keep)] = s[b, keep]
        indices[b, :len(keep)] = keep
    return values, indices, lens


def assert_topk_equivalent(score_matrix, got, want, atol=1e-12):
    """got / want: (values, indices, lens).  Identical value rows and lens; indices identical up to ties (see module doc)."""
    gv, gi, gl = [np.asarray(x) for x in got]
    wv, wi, wl = [np.asarray(x) for x in want]
    assert gv.shape == wv.shape and gi.shape == wi.shape
    np.testing.assert_allclose(gv, wv, rtol=0, atol=atol)
    np.testing.assert_array_equal(gl.reshape(-1), wl.reshape(-1))
    for b in range(gv.shape[0]):
        for k in range(gv.shape[1]):
            for idx, val in ((gi[b, k], gv[b, k]), (wi[b, k], wv[b, k])):
                if idx >= 0:
                    assert abs(score_matrix[b, idx] - val) <= atol, (b, k, idx, val, score_matrix[b, idx])
                else:
                    assert val == 0
            if gi[b, k] != wi[b, k]:                       # only legitimate when that score occurs more than once in the row
                assert gi[b, k] >= 0 and wi[b, k] >= 0, (b, k, gi[b, k], wi[b, k])
                assert (np.abs(score_matrix[b] - gv[b, k]) <= atol).sum() > 1, (b, k, gi[b, k], wi[b, k])
        used = gi[b][gi[b] >= 0]
        assert len(set(used.tolist())) == len(used), "a pool row was returned twice"
