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


def test_exact_match_columns_are_rejected():
    from rat_amd import retrieval
    with pytest.raises(NotImplementedError):
        retrieval.BM25_topk_retrieval_v4(np.zeros((3, 2), dtype=np.int64), np.zeros((2, 2), dtype=np.int64), exact_match_col_indices=[0])


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
