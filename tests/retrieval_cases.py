"""Seeded inputs shared by tests/golden/make_golden_retrieval.py (reference side) and tests/test_retrieval.py."""
import numpy as np

CASES = {
    # three id columns like MovieLens-Tag (user, item, tag): small vocabularies -> many equal scores (ties everywhere)
    "mltag_like": dict(n_db=400, n_qry=37, vocab=[23, 17, 9], topk=5, seed=3, unseen=True),
    # more columns, larger vocabularies: mostly distinct scores
    "wide": dict(n_db=700, n_qry=21, vocab=[300, 250, 40, 12, 500, 8], topk=8, seed=4, unseen=True),
    # topK larger than the pool: padded with -1 / 0 (padded_topk's first branch)
    "tiny_pool": dict(n_db=6, n_qry=9, vocab=[4, 3], topk=10, seed=5, unseen=False),
    # a column that holds ONE value in the whole pool: its IDF is log(1) = 0, matches on it alone score 0 and are dropped
    "constant_column": dict(n_db=120, n_qry=15, vocab=[1, 30, 6], topk=12, seed=6, unseen=True),
}


def make_case(case):
    rs = np.random.RandomState(case["seed"])
    db = np.stack([rs.randint(0, v, size=case["n_db"]) for v in case["vocab"]], axis=1).astype(np.int64)
    qry = np.stack([rs.randint(0, v, size=case["n_qry"]) for v in case["vocab"]], axis=1).astype(np.int64)
    if case["unseen"]:                                   # ids the pool has never seen (valid / test rows): IDF weight 0
        qry[0, :] = [v + 5 for v in case["vocab"]]
        qry[1, 0] = case["vocab"][0] + 7
    return db, qry


def run_variants(case):
    """the reference's execution modes.  The query batching is part of the result (the IDF mapping's dtype rule looks at the
    first row of every batch, see oracle/retrieval_oracle.py:map_idf); the pool chunking only reorders ties."""
    return {"whole": dict(), "chunked": dict(qry_batch_size=8, db_chunk_size=50),
            "rechunked": dict(qry_batch_size=8, db_chunk_size=64, enable_clean=True)}
