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


# ----------------------------------------------------------------------------------------------------------------------
def precompute(data, cfg, pool=None):
    """DataGenerator's pre-retrieval branch (fuxictr/pytorch/data_generator.py:106-215) restated on top of ``topk``: pool "self"
    = <X>-fold retrieval (every fold queries the others), otherwise a separate pool; ``label_wise`` = top-K among the pool's
    positives and among its negatives.  Padded (-1) entries go through the reference's index arithmetic unchanged:
    ``index_map[-1]`` is the LAST element of the map.  Returns (indices, values, lens, score) where score[b] maps a GLOBAL pool row
    to its score for query b per label half (NaN where the row is not a candidate) — used to compare indices up to ties."""
    import re
    cols, k, qb, lw = cfg["used_col_indices"], cfg["topK"], cfg.get("qry_batch_size"), bool(cfg.get("label_wise"))
    exm = cfg.get("exact_match_col_indices")

    def one(db, db_labels, qry, index_map, n_global):
        halves = [np.nonzero(db_labels)[0], np.nonzero(1 - db_labels)[0]] if lw else [np.arange(len(db))]
        idx_parts, val_parts, len_parts, score_parts = [], [], [], []
        for sel in halves:
            if exm:
                v, i, ln, sub_scores = topk_exact(db[sel], qry, exm, k, qb)
            else:
                v, i, ln = topk(db[sel], qry, k, qb)
                sub_scores = scores(db[sel], qry, qb)
            g = sel[i]                                            # -1 -> sel[-1], as in the reference
            g = g if index_map is None else index_map[g]
            if not lw and index_map is None:
                g = i                                             # data_generator.py:210: the raw result, -1 kept
            sc = np.full((len(qry), n_global), np.nan)
            cand = sel if index_map is None else index_map[sel]
            sc[:, cand] = sub_scores
            idx_parts.append(g), val_parts.append(v), len_parts.append(ln), score_parts.append(sc)
        if lw:
            return np.concatenate(idx_parts, -1), np.concatenate(val_parts, -1), np.stack(len_parts, -1), score_parts
        return idx_parts[0], val_parts[0], len_parts[0], score_parts

    if pool is None:
        ids = data[:, cols].astype(int)
        labels = data[:, -1].astype(int)
        fold_num = int(re.match(r"\d+-fold", cfg["split_type"]).group().split("-")[0])
        fold_size = int(np.ceil(len(ids) / fold_num))
        outs = []
        for fi in range(fold_num):
            lo, hi = fi * fold_size, (fi + 1) * fold_size
            if len(ids[lo:hi]) == 0:
                continue
            db = np.concatenate([ids[:lo], ids[hi:]], 0)
            index_map = np.concatenate([np.arange(lo), np.arange(min(hi, len(ids)), len(ids))], 0)
            outs.append(one(db, np.concatenate([labels[:lo], labels[hi:]], 0), ids[lo:hi], index_map, len(ids)))
        nh = len(outs[0][3])
        return (np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]), np.concatenate([o[2] for o in outs]),
                [np.concatenate([o[3][h] for o in outs], 0) for h in range(nh)])
    return one(pool[:, cols].astype(int), pool[:, -1].astype(int), data[:, cols].astype(int), None, len(pool))


def assert_driver_equivalent(got, want, score_halves, k, atol=1e-12):
    """got / want: (indices, values, lens) as stored on disk; values and lens identical, indices identical up to ties: every
    positive-valued entry must point at a candidate row with exactly that score, padded entries must be identical."""
    gi, gv, gl = [np.asarray(x) for x in got]
    wi, wv, wl = [np.asarray(x) for x in want]
    np.testing.assert_allclose(gv, wv, rtol=0, atol=atol)
    np.testing.assert_array_equal(gl, wl)
    assert gi.shape == wi.shape
    for h, sc in enumerate(score_halves):
        sl = slice(h * k, (h + 1) * k)
        for b in range(gi.shape[0]):
            for idx_row, val_row in ((gi[b, sl], gv[b, sl]), (wi[b, sl], wv[b, sl])):
                for idx, val in zip(idx_row, val_row):
                    if val > 0:
                        assert abs(sc[b, idx] - val) <= atol, (h, b, idx, val, sc[b, idx])
            pad = gv[b, sl] == 0
            np.testing.assert_array_equal(gi[b, sl][pad], wi[b, sl][pad])


# ----------------------------------------------------------------------------------------------------------------------
# exact_match_col_indices (data_utils.py:851-866, 895-911, 921-938, 1038-1048): candidates of a query are the pool rows that
# agree with it on ALL exact-match columns; BM25 runs over the remaining columns only.
#
# Parity status: PINNED against the reference's function run in the build container (tests/golden/make_golden_retrieval.py),
# with ONE stand-in: the reference pads the group member lists with tensorflow.keras `pad_sequences` (data_utils.py:34, :903);
# tensorflow is absent from the image (and unpinned by the reference — it ships no requirements file), so the generator
# restates its published behaviour (padding='post', truncating='pre' by default: a list longer than maxlen keeps its LAST
# maxlen entries) — see `pad_post` below, which is that restatement on this side.
def exact_match_groups(db, qry, exm_cols):
    """(group id per pool row, group id per query or -1 when no pool row carries the query's key) —
    `db_df.groupby(cols).groups` / `get_indexer` (data_utils.py:852-859).  Groups are numbered in ascending key order."""
    keys, db_grp = np.unique(db[:, exm_cols], axis=0, return_inverse=True)
    lut = {tuple(k.tolist()): g for g, k in enumerate(keys)}
    qry_grp = np.array([lut.get(tuple(r.tolist()), -1) for r in qry[:, exm_cols]], dtype=np.int64).reshape(-1)
    return db_grp.reshape(-1).astype(np.int64), qry_grp


def pad_post(seqs, maxlen, value=-1):
    """keras pad_sequences(padding='post', truncating='pre', maxlen=maxlen or the longest)"""
    width = max(len(s) for s in seqs) if maxlen is None else maxlen
    out = np.full((len(seqs), width), value, dtype=np.int64)
    for i, s in enumerate(seqs):
        t = np.asarray(s)[-width:] if len(s) else np.asarray(s)
        out[i, :len(t)] = t
    return out


def topk_exact(db, qry, exm_cols, k, qry_batch_size=None):
    """-> (values, indices, lens, score_matrix [Q, N]).  Per query batch (the batching IS part of the result):
      * queries whose key has no pool row keep (0, -1, len 0)                                         (data_utils.py:1047-1050)
      * if no group of the batch is larger than K: every member in ascending pool order, value 1.0       (:911-917, 1038-1044)
        (with no remaining column the groups are first cut to their LAST K members, pad_sequences maxlen=K, :903-905)
      * otherwise score = (BM25 over the remaining columns + 1) for group members, top-K of that          (:918-1037)
    score_matrix holds the value each (query, pool row) pair would be returned with (0 = not a candidate)."""
    exm_cols = list(exm_cols)
    rest = [c for c in range(db.shape[1]) if c not in exm_cols]
    db_grp, qry_grp = exact_match_groups(db, qry, exm_cols)
    members = [np.nonzero(db_grp == g)[0] for g in range(db_grp.max() + 1 if len(db_grp) else 0)]
    db_r, qry_r = db[:, rest], qry[:, rest]
    tables = idf_tables(db_r)
    q, n = len(qry), len(db)
    values = np.zeros((q, k), dtype=np.float64)
    indices = np.full((q, k), -1, dtype=np.int64)
    lens = np.zeros(q, dtype=np.int64)
    score = np.zeros((q, n), dtype=np.float64)
    step = q if qry_batch_size is None else qry_batch_size
    for q0 in range(0, q, step):
        rows = np.arange(q0, min(q0 + step, q))
        rows = rows[qry_grp[rows] != -1]
        if len(rows) == 0:
            continue
        padded = pad_post([members[qry_grp[b]] for b in rows], k if not rest else None)
        if padded.shape[1] <= k:
            for j, b in enumerate(rows):
                m = padded[j][padded[j] != -1]
                lens[b] = len(m)
                indices[b, :len(m)] = m
                values[b, :len(m)] = 1.0
                score[b, m] = 1.0
            continue
        w = map_idf(qry_r[rows], tables)
        for j, b in enumerate(rows):
            m = members[qry_grp[b]]
            s = np.ones(len(m), dtype=np.float64)
            bm = np.zeros(len(m), dtype=np.float64)
            for f in range(len(rest)):
                bm += (qry_r[b, f] == db_r[m, f]) * w[j, f]
            s = (bm + 1.0) * s
            score[b, m] = s
            order = np.lexsort((m, -s))[:k]
            lens[b] = len(order)
            indices[b, :len(order)] = m[order]
            values[b, :len(order)] = s[order]
    return values, indices, lens, score
