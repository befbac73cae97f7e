#!/usr/bin/env python3
"""Diagnostic: throughput of the top-K retrieval kernel (rat_bm25_topk) on a MovieLens-Tag-sized problem.

    python tools/retrieval_bench.py [--n-db 1400000] [--n-qry 20000] [--fields 3] [--topk 5]

Reports (query, pool row) pairs scored per second and the algorithmic read rate of the kernel: one pass over the pool's id
columns (n_db x fields x 4 B) per tile of 4 queries (topK <= 8) — SURVEY §8f rank 3.  The reference's CPU path for the same
problem can be timed with --reference-sample N (needs nothing from /root/reference: it times the ORACLE restatement)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "www24-rat_amd")):
    sys.path.insert(0, p)
from rat_amd import retrieval  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-db", type=int, default=1_400_000)
    ap.add_argument("--n-qry", type=int, default=20_000)
    ap.add_argument("--fields", type=int, default=3)
    ap.add_argument("--topk", type=int, default=5)
    ap.add_argument("--oracle-sample", type=int, default=0, help="also time the numpy oracle on this many queries (CPU)")
    args = ap.parse_args()
    rs = np.random.RandomState(0)
    vocab = [17_000, 23_000, 49_000, 300, 40, 12][: args.fields] + [1000] * max(0, args.fields - 6)
    db = np.stack([rs.randint(0, v, size=args.n_db) for v in vocab], axis=1).astype(np.int64)
    qry = np.stack([rs.randint(0, v, size=args.n_qry) for v in vocab], axis=1).astype(np.int64)
    from rat_amd._lib import get_lib
    lib = get_lib()
    inner, events = lib.call, []

    def timed_call(name, *a):                     # HIP events around the kernel launch, on the stream it is launched on
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        inner(name, *a)
        e.record()
        events.append((s, e))
    retrieval.BM25_topk_retrieval_v4(db, qry[:512], device="cuda:0", topK=args.topk)          # warm-up (library load, clocks)
    torch.cuda.synchronize()
    lib.call = timed_call
    t0 = time.perf_counter()
    res = retrieval.BM25_topk_retrieval_v4(db, qry, device="cuda:0", topK=args.topk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.call = inner
    kt = sum(s.elapsed_time(e) for s, e in events) * 1e-3
    pairs = args.n_db * args.n_qry
    tile = 4 if args.topk <= 8 else 1
    bytes_read = args.n_db * args.fields * 4 * ((args.n_qry + tile - 1) // tile)
    print("rat_bm25_topk  n_db %d  n_qry %d  F %d  K %d : %.3f s end to end (host IDF mapping + H2D + kernel + D2H)" %
          (args.n_db, args.n_qry, args.fields, args.topk, dt))
    print("   kernel alone %.4f s: %.1f G pairs/s, %.1f GB/s algorithmic pool reads (one pass over the id columns per tile of %d "
          "queries; %.1f %% of 8 TB/s); end to end %.1f G pairs/s; mean lens %.2f" %
          (kt, pairs / kt / 1e9, bytes_read / kt / 1e9, tile, 100 * bytes_read / kt / 8e12, pairs / dt / 1e9, float(res.lens.mean())))
    if args.oracle_sample:
        from oracle import retrieval_oracle as ro
        t0 = time.perf_counter()
        ro.topk(db, qry[: args.oracle_sample], args.topk)
        dt_o = time.perf_counter() - t0
        print("   numpy oracle on %d queries: %.2f s -> %.4f G pairs/s" % (args.oracle_sample, dt_o, args.n_db * args.oracle_sample / dt_o / 1e9))


if __name__ == "__main__":
    main()
