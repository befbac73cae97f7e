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


# ---- exact_match_col_indices (data_utils.py:851-866): candidates = pool rows equal to the query on the exact-match columns
EXM_CASES = {
    # one exact-match column, groups larger than topK: BM25 + 1 over the other columns inside the group
    "one_col": dict(n_db=300, n_qry=29, vocab=[6, 17, 9, 4], exm=[0], topk=5, seed=31, unseen=True),
    # two exact-match columns (not adjacent), small groups mixed with large ones
    "two_cols": dict(n_db=500, n_qry=33, vocab=[5, 40, 6, 7], exm=[0, 2], topk=6, seed=32, unseen=True),
    # every group fits in topK: members in pool order with value 1.0, no scoring at all
    "groups_fit": dict(n_db=90, n_qry=17, vocab=[30, 5, 4], exm=[0], topk=12, seed=33, unseen=True),
    # all columns exact-match: nothing left to score, groups cut to their LAST topK members
    "all_cols": dict(n_db=200, n_qry=25, vocab=[4, 3], exm=[0, 1], topk=7, seed=34, unseen=False),
}


def make_exm_case(case):
    return make_case(case)


def run_exm_variants(case):
    """the query batching decides, batch by batch, between the "every group fits" shortcut and scoring"""
    return {"whole": dict(), "chunked": dict(qry_batch_size=8, db_chunk_size=50), "small_batches": dict(qry_batch_size=3)}


# ---- the DataGenerator-level driver (fuxictr/pytorch/data_generator.py:106-215): fold / separate-pool / label-wise retrieval
DRIVER_CASES = {
    "fold3_self": dict(n=50, n_pool=None, vocab=[9, 7, 5], topk=4, split_type="3-fold", label_wise=False, seed=21, qry_batch_size=16),
    "fold4_self_labelwise": dict(n=61, n_pool=None, vocab=[6, 5, 4], topk=3, split_type="4-fold", label_wise=True, seed=22, qry_batch_size=None),
    "separate_pool": dict(n=23, n_pool=40, vocab=[8, 6], topk=5, split_type="random", label_wise=False, seed=23, qry_batch_size=10),
    "separate_pool_labelwise": dict(n=19, n_pool=35, vocab=[5, 4, 3], topk=2, split_type="sequential", label_wise=True, seed=24, qry_batch_size=None),
    "fold3_self_exact": dict(n=80, n_pool=None, vocab=[4, 7, 5], topk=3, split_type="3-fold", label_wise=False, seed=25, qry_batch_size=16, exm=[0]),
    "separate_pool_labelwise_exact": dict(n=21, n_pool=70, vocab=[3, 4, 5], topk=2, split_type="random", label_wise=True, seed=26, qry_batch_size=None,
                                          exm=[0]),
}


def make_driver_case(case):
    """-> (data [n, F+1] float64 with the label last, pool or None, retrieval_configs as h5_generator prepares them)"""
    rs = np.random.RandomState(case["seed"])

    def table(n):
        ids = np.stack([rs.randint(0, v, size=n) for v in case["vocab"]], axis=1)
        return np.concatenate([ids, rs.randint(0, 2, size=(n, 1))], axis=1).astype(np.float64)
    data = table(case["n"])
    pool = None if case["n_pool"] is None else table(case["n_pool"])
    cfg = dict(pre_retrieval=True, split_type=case["split_type"], label_wise=case["label_wise"], topK=case["topk"],
               used_col_indices=list(range(len(case["vocab"]))), exact_match_col_indices=case.get("exm"),
               qry_batch_size=case["qry_batch_size"], db_chunk_size=17, device="cpu", enable_clean=False)
    return data, pool, cfg
