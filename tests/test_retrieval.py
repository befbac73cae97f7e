"""Top-K retrieval pre-compute: oracle pinned to the reference's outputs, HIP kernel against the oracle.

CPU: oracle vs tests/golden/retrieval.npz (real reference, three execution modes), and the kernel source through the
host-emulation build.  GPU (-m gpu): the HIP kernel through the C ABI on the golden cases, and a larger pool checked through
size-independent properties (values are true scores, rows sorted, no better candidate left out)."""
import os
import sys

import numpy as np
import pytest
import torch

import retrieval_cases as rc
from oracle import retrieval_oracle as ro

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "retrieval.npz")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


def gold_of(name, tag):
    g = np.load(GOLD)
    return g["%s/%s/values" % (name, tag)], g["%s/%s/indices" % (name, tag)], g["%s/%s/lens" % (name, tag)]


@pytest.mark.parametrize("name", list(rc.CASES))
def test_oracle_matches_reference(name):
    case = rc.CASES[name]
    db, qry = rc.make_case(case)
    for tag, kw in rc.run_variants(case).items():
        qb = kw.get("qry_batch_size")
        ro.assert_topk_equivalent(ro.scores(db, qry, qb), ro.topk(db, qry, case["topk"], qb), gold_of(name, tag))


def test_db_chunking_changes_only_tie_order():
    """same query batching, different pool chunking: the value rows and lens of the reference runs are identical"""
    for name, case in rc.CASES.items():
        a, b = gold_of(name, "chunked"), gold_of(name, "rechunked")
        np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(a[2], b[2])


def check_product(name, device, lib):
    from rat_amd import retrieval
    case = rc.CASES[name]
    db, qry = rc.make_case(case)
    for tag, kw in rc.run_variants(case).items():
        qb = kw.get("qry_batch_size")
        want = ro.topk(db, qry, case["topk"], qb)
        got = retrieval.BM25_topk_retrieval_v4(db_np_data=db, qry_np_data=qry, device=device, topK=case["topk"], lib=lib, **kw)
        np.testing.assert_array_equal(got.indices, want[1])                    # same deterministic tie order as the oracle
        np.testing.assert_allclose(got.values, want[0], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(got.lens, want[2])
        ro.assert_topk_equivalent(ro.scores(db, qry, qb), got, gold_of(name, tag))   # and equivalent to the reference's own output


# the emulator runs one OS thread per GPU thread: two cases on the CPU (ties everywhere; topK > pool), all four on the GPU
@pytest.mark.parametrize("name", ["mltag_like", "tiny_pool"])
def test_kernel_emulated(name):
    import build_emu
    import rat_amd._lib as L
    check_product(name, "cpu", L.RatLib(build_emu.build()))


def check_same_lane_ties(device, lib):
    """Rows 0, 256 and 512 are scanned by the SAME lane (256 lanes, row = lane + 256 j): two equal scores arrive first, then a
    better one.  The displaced entries must keep their arrival order in the lane's private list, or the merge sees row 256 before
    row 0 (and with a short topK loses row 0 altogether)."""
    from rat_amd import retrieval
    for topk in (2, 3, 9):                                  # 9 > 8: the one-query-per-work-group instantiation
        db = np.full((1024, 2), 7, dtype=np.int64)
        db[:, 0] = np.arange(1024) % 5 + 10                 # column 0: five values; column 1: everything 7 except the marked rows
        for r in (0, 256, 300, 512, 700):
            db[r, 1] = 3
        db[512, 0], db[700, 0] = 99, 98                     # rare ids: rows 512 / 700 match the queries on BOTH columns
        db[[0, 256, 300], 0] = 50
        qry = np.array([[99, 3], [98, 3]], dtype=np.int64)
        want = ro.topk(db, qry, topk)
        assert want[1][0][:2].tolist() == [512, 0] and (topk < 3 or want[1][0][2] == 256)
        got = retrieval.BM25_topk_retrieval_v4(db, qry, device=device, topK=topk, lib=lib)
        np.testing.assert_array_equal(got.indices, want[1])
        np.testing.assert_allclose(got.values, want[0], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(got.lens, want[2])


def test_same_lane_ties_keep_pool_order_emulated():
    import build_emu
    import rat_amd._lib as L
    check_same_lane_ties("cpu", L.RatLib(build_emu.build()))


@pytest.mark.gpu
def test_same_lane_ties_keep_pool_order_gpu():
    import rat_amd._lib as L
    check_same_lane_ties("cuda:0", L.get_lib())


# ---- exact_match_col_indices: group filter + (BM25 + 1) inside the group (data_utils.py:851-866, 895-938, 1038-1050)
def gold_exm(name, tag):
    return gold_of("exm/" + name, tag)


@pytest.mark.parametrize("name", list(rc.EXM_CASES))
def test_exact_match_oracle_matches_reference(name):
    case = rc.EXM_CASES[name]
    db, qry = rc.make_exm_case(case)
    took_shortcut, scored = False, False
    for tag, kw in rc.run_exm_variants(case).items():
        v, i, ln, sc = ro.topk_exact(db, qry, case["exm"], case["topk"], kw.get("qry_batch_size"))
        ro.assert_topk_equivalent(sc, (v, i, ln), gold_exm(name, tag))
        took_shortcut |= bool(((v == 1.0) | (v == 0.0)).all())
        scored |= bool((v > 1.0).any())
    assert scored == (name in ("one_col", "two_cols")) and (took_shortcut or scored)


def check_exact_match_product(name, device, lib):
    from rat_amd import retrieval
    case = rc.EXM_CASES[name]
    db, qry = rc.make_exm_case(case)
    for tag, kw in rc.run_exm_variants(case).items():
        wv, wi, wl, sc = ro.topk_exact(db, qry, case["exm"], case["topk"], kw.get("qry_batch_size"))
        got = retrieval.BM25_topk_retrieval_v4(db_np_data=db, qry_np_data=qry, exact_match_col_indices=list(case["exm"]), device=device,
                                               topK=case["topk"], lib=lib, **kw)
        np.testing.assert_array_equal(got.indices, wi)                         # same deterministic tie order as the oracle
        np.testing.assert_allclose(got.values, wv, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(got.lens, wl)
        ro.assert_topk_equivalent(sc, got, gold_exm(name, tag))                # and equivalent to the reference's own output


@pytest.mark.parametrize("name", ["one_col", "all_cols"])
def test_exact_match_emulated(name):
    import build_emu
    import rat_amd._lib as L
    check_exact_match_product(name, "cpu", L.RatLib(build_emu.build()))


def test_exact_match_col_indices_follow_h5_generator():
    from rat_amd import retrieval
    cfg = {"used_cols": ["user_id", "item_id", "tag_id"], "exact_match_cols": ["tag_id", "user_id"]}
    assert retrieval.exact_match_col_indices(cfg) == [2, 0]                    # positions inside used_cols, in exact_match_cols order
    assert retrieval.exact_match_col_indices({"used_cols": ["a"], "exact_match_cols": []}) is None


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(rc.EXM_CASES))
def test_exact_match_gpu(name):
    import rat_amd._lib as L
    check_exact_match_product(name, "cuda:0", L.get_lib())


@pytest.mark.gpu
def test_exact_match_large_pool_properties_gpu():
    """150k-row pool, 2 exact-match columns of 5 used, 400 queries: results of a query sample against the oracle, and for every
    query: all returned rows belong to its group, values sorted, len = min(group size, K)."""
    import rat_amd._lib as L
    from rat_amd import retrieval
    rs = np.random.RandomState(12)
    vocab = [40, 3000, 25, 200, 50]
    exm, topk = [0, 2], 10
    db = np.stack([rs.randint(0, v, size=150_000) for v in vocab], axis=1).astype(np.int64)
    qry = np.stack([rs.randint(0, v, size=400) for v in vocab], axis=1).astype(np.int64)
    qry[5, 0] = 99                                                              # a key the pool does not hold
    got = retrieval.BM25_topk_retrieval_v4(db, qry, exact_match_col_indices=exm, device="cuda:0", topK=topk, qry_batch_size=128,
                                           lib=L.get_lib())
    assert (np.diff(got.values, axis=1) <= 0).all() and got.lens[5] == 0 and (got.indices[5] == -1).all()
    for b in range(len(qry)):
        idx = got.indices[b][got.indices[b] >= 0]
        assert (db[idx][:, exm] == qry[b, exm]).all()
        assert got.lens[b] == min(topk, int((db[:, exm] == qry[b, exm]).all(axis=1).sum())) == len(idx)
    wv, wi, wl, _ = ro.topk_exact(db, qry, exm, topk, 128)
    np.testing.assert_array_equal(got.indices, wi)
    np.testing.assert_allclose(got.values, wv, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got.lens, wl)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(rc.CASES))
def test_kernel_gpu(name):
    import rat_amd._lib as L
    check_product(name, "cuda:0", L.get_lib())


@pytest.mark.gpu
@pytest.mark.parametrize("topk", [5, 20])
def test_large_pool_properties_gpu(topk):
    """200k-row pool x 512 queries: every returned value is the true score of its index, rows are sorted, and no pool row
    outside the result beats the K-th entry (checked against the full score matrix of a query sample)."""
    import rat_amd._lib as L
    from rat_amd import retrieval
    rs = np.random.RandomState(9)
    vocab = [5000, 3000, 200, 50]
    db = np.stack([rs.randint(0, v, size=200_000) for v in vocab], axis=1).astype(np.int64)
    qry = np.stack([rs.randint(0, v, size=512) for v in vocab], axis=1).astype(np.int64)
    got = retrieval.BM25_topk_retrieval_v4(db, qry, device="cuda:0", topK=topk, qry_batch_size=200, lib=L.get_lib())
    assert (np.diff(got.values, axis=1) <= 0).all()
    sample = rs.permutation(512)[:48]
    s = ro.scores(db, qry[sample])
    want = ro.topk(db, qry[sample], topk)
    np.testing.assert_array_equal(got.indices[sample], want[1])
    np.testing.assert_allclose(got.values[sample], want[0], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got.lens[sample], want[2])
    ro.assert_topk_equivalent(s, (got.values[sample], got.indices[sample], got.lens[sample]), want)


@pytest.mark.gpu
@pytest.mark.parametrize("topk", [5, 20])
def test_heavy_ties_keep_pool_order_gpu(topk):
    """tiny vocabularies over a 60k-row pool: almost every score is shared by thousands of rows, so the result is decided by the tie
    order alone (lower pool index first) — in every lane's private list and in the merge"""
    import rat_amd._lib as L
    from rat_amd import retrieval
    rs = np.random.RandomState(17)
    vocab = [3, 4, 2, 5]
    db = np.stack([rs.randint(0, v, size=60_000) for v in vocab], axis=1).astype(np.int64)
    qry = np.stack([rs.randint(0, v, size=64) for v in vocab], axis=1).astype(np.int64)
    got = retrieval.BM25_topk_retrieval_v4(db, qry, device="cuda:0", topK=topk, lib=L.get_lib())
    want = ro.topk(db, qry, topk)
    np.testing.assert_array_equal(got.indices, want[1])
    np.testing.assert_allclose(got.values, want[0], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got.lens, want[2])
    got = retrieval.BM25_topk_retrieval_v4(db, qry, exact_match_col_indices=[1], device="cuda:0", topK=topk, lib=L.get_lib())
    wv, wi, wl, _ = ro.topk_exact(db, qry, [1], topk)
    np.testing.assert_array_equal(got.indices, wi)
    np.testing.assert_allclose(got.values, wv, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(got.lens, wl)


# ---- the driver: fold / separate-pool / label-wise pre-retrieval as DataGenerator runs it (data_generator.py:106-215)
def gold_driver(name):
    g = np.load(GOLD)
    return g["driver/%s/indices" % name], g["driver/%s/values" % name], g["driver/%s/lens" % name]


@pytest.mark.parametrize("name", list(rc.DRIVER_CASES))
def test_driver_oracle_matches_reference(name):
    case = rc.DRIVER_CASES[name]
    data, pool, cfg = rc.make_driver_case(case)
    idx, val, lens, sc = ro.precompute(data, cfg, pool)
    ro.assert_driver_equivalent((idx, val, lens), gold_driver(name), sc, case["topk"])


def check_driver_product(name, device, lib):
    from rat_amd import retrieval
    case = rc.DRIVER_CASES[name]
    data, pool, cfg = rc.make_driver_case(case)
    oi, ov, ol, sc = ro.precompute(data, cfg, pool)
    gi, gv, gl = retrieval.precompute_retrieval(data, cfg, cfg["used_col_indices"], pool_array=pool, device=device, lib=lib)
    np.testing.assert_array_equal(gi, oi)                       # same deterministic tie order as the oracle
    np.testing.assert_allclose(gv, ov, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(gl, ol)
    ro.assert_driver_equivalent((gi, gv, gl), gold_driver(name), sc, case["topk"])


@pytest.mark.parametrize("name", ["fold3_self", "separate_pool_labelwise"])
def test_driver_emulated(name):
    import build_emu
    import rat_amd._lib as L
    check_driver_product(name, "cpu", L.RatLib(build_emu.build()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(rc.DRIVER_CASES))
def test_driver_gpu(name):
    import rat_amd._lib as L
    check_driver_product(name, "cuda:0", L.get_lib())
