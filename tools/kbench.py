#!/usr/bin/env python3
"""Diagnostic: time single C-ABI kernels at the north-star shapes (B=4096, T=11, S=21, d=64, h=8, dh=10, H=128).

    python tools/kbench.py [ffn_fwd ffn_bwd attn_fwd attn_bwd] [--reps 10] [--check]

Never used by the product or the tests; bench.py is the contract benchmark."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "www24-rat_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from rat_amd import ops  # noqa: E402

PEAK = 157.3


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["ffn_fwd", "ffn_bwd", "attn_fwd", "attn_bwd"])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--arith", default="f32", choices=["f32", "bf16x3", "both"])
    ap.add_argument("--gather-rows", type=int, default=50_000, help="gather: rows per field (2500000 with --gather-F 40 = configs[3]'s 25.6 GB table)")
    ap.add_argument("--gather-F", type=int, default=20)
    ap.add_argument("--T", type=int, default=11, help="samples per batch row (retrieved + 1): the cross-sample sequence length")
    ap.add_argument("--S", type=int, default=21, help="tokens per sample (fields + 1): the intra-sample sequence length")
    args = ap.parse_args()
    dev = "cuda"
    B, T, S, d, heads, dh, H = args.B, args.T, args.S, 64, 8, 10, 128
    I = heads * dh
    tok = B * T * S
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    if "gather" in args.which:
        from types import SimpleNamespace
        F, vocab = args.gather_F, args.gather_rows
        S = F + 1
        dy = torch.randn(B, T, S, d, generator=g).to(dev)
        table = torch.empty(F * vocab, d, device=dev).normal_(0.0, 0.01)
        fields = [SimpleNamespace(col=i, ncols=1, vocab=vocab, padding_idx=None) for i in range(F)]
        tabs = [table[i * vocab:(i + 1) * vocab] for i in range(F)]
        gtable = torch.zeros_like(table)
        ftab = ops.field_table(fields, tabs, dev)
        gftab = ops.field_table(fields, [gtable[i * vocab:(i + 1) * vocab] for i in range(F)], dev)
        label_tab = rn(3, d)
        idxs = [torch.randint(0, vocab, (B, T, F), generator=g).to(torch.int32).to(dev) for _ in range(4)]
        labels = torch.randint(0, 2, (B, T), generator=g).to(torch.int32).to(dev)
        it = [0]

        def fwd():
            it[0] += 1
            ops.gather_fwd(idxs[it[0] % 4], labels, ftab, F, label_tab, B, T, F, d)
        ms = timeit(fwd, args.reps)
        nb = B * (T * F * d * 4 + T * S * d * 4 + T * F * 4)
        print("gather_fwd %.4f ms  %.0f GB/s  (%.1f %% of 8 TB/s)" % (ms, nb / ms / 1e6, 100 * nb / ms / 1e6 / 8000))
        dlab = torch.zeros(3, d, device=dev)
        dflat = rn(B, F * d)

        def bwd():
            it[0] += 1
            ops.gather_bwd(dy, dflat, idxs[it[0] % 4], labels, gftab, F, dlab, B, T, F, d)
        ms = timeit(bwd, args.reps)
        nb = B * (T * S * d * 4 + 2 * T * F * d * 4 + T * F * 4)
        print("gather_bwd %.4f ms  %.0f GB/s  (%.1f %% of 8 TB/s)" % (ms, nb / ms / 1e6, 100 * nb / ms / 1e6 / 8000))
        del table, gtable, dy
        S = args.S
    x = torch.randn(B, T, S, d, generator=g).to(dev)
    dy = torch.randn(B, T, S, d, generator=g).to(dev)
    if "merge" in args.which:
        # the table-gradient merge of a data-parallel step at the strong-scaling rank shape (8 ranks x 512 samples of N2): every rank sorts
        # and reduces the union of all lists gathered at capacity (round 3) against the owner's share only (round 4, 1 / world of it)
        world, Bl, F, rows_total = 8, 512, 20, 1_000_000
        cap = min(Bl * T * F, rows_total)
        per = -(-rows_total // world)

        def one_list(seed):
            gg = torch.Generator().manual_seed(seed)
            r = torch.unique(torch.randint(0, rows_total, (Bl * T * F,), generator=gg)).to(torch.int32)
            return r

        lists = [one_list(100 + k) for k in range(world)]
        # (a) all lists at capacity
        rows_a = torch.zeros(world, cap, dtype=torch.int32)
        counts_a = torch.tensor([len(r) for r in lists], dtype=torch.int32)
        for k, r in enumerate(lists):
            rows_a[k, :len(r)] = r
        rows_a, counts_a = rows_a.to(dev), counts_a.to(dev)
        grads_a = torch.randn(world * cap, d, device=dev)
        # (b) what owner 0 receives: the rows of its range from every list, packed in rank order
        recv = torch.cat([r[r < per] for r in lists])
        cap_b = -(-len(recv) // 4096) * 4096
        rows_b = torch.zeros(cap_b, dtype=torch.int32)
        rows_b[:len(recv)] = recv
        rows_b, count_b = rows_b.to(dev), torch.tensor([len(recv)], dtype=torch.int32, device=dev)
        grads_b = torch.randn(cap_b, d, device=dev)
        for label, rows, counts, c, w, grads in (("all-gather at capacity: union of %d lists" % world, rows_a, counts_a, cap, world, grads_a),
                                                 ("owner-partitioned: one owner's share", rows_b, count_b, cap_b, 1, grads_b)):
            n = c * w
            out_rows = torch.empty(min(n, rows_total), dtype=torch.int32, device=dev)
            out_grads = torch.empty((min(n, rows_total), d), device=dev)
            plan = [None]

            def run():
                plan[0] = ops.sparse_plan_rows(rows.reshape(-1), counts, c, w, rows_total, plan=plan[0])
                ops.sparse_reduce_rows(plan[0], grads, c, w, d, out_rows, out_grads)
            ms = timeit(run, args.reps)
            print("merge %-48s n = %7d pairs  %.4f ms per table family" % (label, n, ms))
    if any(w.startswith("ffn") for w in args.which):
        w1, b1, w2, b2 = rn(H, d, sc=d ** -0.5), rn(H, sc=0.1), rn(d, H, sc=H ** -0.5), rn(d, sc=0.1)
        y = torch.empty_like(x)
        for arith in (["f32", "bf16x3"] if args.arith == "both" else [args.arith]):
            if "ffn_fwd" in args.which:
                ms = timeit(lambda: ops.ffn_fwd(x, w1, b1, w2, b2, d, H, out=y, arith=arith), args.reps)
                fl = tok * 4 * d * H
                print("ffn_fwd %-6s %.4f ms  %.1f TFLOP/s  (%.1f %% of %.1f)" % (arith, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / PEAK, PEAK))
            if "ffn_bwd" in args.which:
                gs = [torch.zeros_like(t) for t in (w1, b1, w2, b2)]
                ws = [None]

                def run():
                    _, ws[0] = ops.ffn_bwd(x, dy, w1, b1, w2, b2, gs[0], gs[1], gs[2], gs[3], d, H, workspace=ws[0], arith=arith)
                ms = timeit(run, args.reps)
                fl = 2 * tok * 4 * d * H
                print("ffn_bwd %-6s %.4f ms  %.1f TFLOP/s  (%.1f %%)" % (arith, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / PEAK))
    if "core" in args.which:
        # RAT_m0's joint attention core at the headline shape: B sequences of T * S tokens (rat_attn_core_*_map's contiguous form)
        Lc = T * S
        qkv = rn(B * Lc, 3 * I)
        dout = rn(B * Lc, I)
        o, lse = ops.attn_core_fwd(qkv, B, Lc, heads, dh)
        fl = B * Lc * 4 * I * Lc
        ms = timeit(lambda: ops.attn_core_fwd(qkv, B, Lc, heads, dh), args.reps)
        print("core_fwd L%-3d %.4f ms  %.1f TFLOP/s  (%.1f %%)" % (Lc, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / PEAK))
        ms = timeit(lambda: ops.attn_core_bwd(qkv, o, lse, dout, B, Lc, heads, dh), args.reps)
        print("core_bwd L%-3d %.4f ms  %.1f TFLOP/s  (%.1f %%)" % (Lc, ms, 2 * fl / ms / 1e9, 100 * 2 * fl / ms / 1e9 / PEAK))
        del qkv, dout, o, lse
    if any(w.startswith("attn") for w in args.which):
        ln_g, ln_b = 1 + rn(d, sc=0.1), rn(d, sc=0.1)
        w_qkv, w_out, b_out = rn(3 * I, d, sc=d ** -0.5), rn(d, I, sc=I ** -0.5), rn(d, sc=0.1)
        params = ops.attn_params(ln_g, ln_b, w_qkv, w_out, b_out)
        gsl = [torch.zeros_like(t) for t in (ln_g, ln_b, w_qkv, w_out, b_out)]
        grads = ops.attn_params(*gsl)
        for arith in (["f32", "bf16x3"] if args.arith == "both" else [args.arith]):
          for mode, smap, L in (("intra", ops.intra_map(B, T, S), S), ("cross", ops.cross_map(B, T, S), T)):
            y = torch.empty_like(x)
            fl = tok * (8 * d * I + 4 * I * L)
            y, o_save, lse = ops.attn_fwd(x, params, smap, d, heads, dh, save=True, out=y, arith=arith)
            if "attn_fwd" in args.which:
                ms = timeit(lambda: ops.attn_fwd(x, params, smap, d, heads, dh, save=True, out=y, arith=arith), args.reps)
                print("attn_fwd %-6s L%-2d %.4f ms  %.1f TFLOP/s  (%.1f %%)" % (arith, L, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / PEAK))
            if "attn_bwd" in args.which:
                ws = [None]

                def run():
                    _, ws[0] = ops.attn_bwd(x, dy, o_save, lse, params, grads, smap, d, heads, dh, workspace=ws[0], arith=arith)
                ms = timeit(run, args.reps)
                print("attn_bwd %-6s L%-2d %.4f ms  %.1f TFLOP/s  (%.1f %%)" % (arith, L, ms, 2 * fl / ms / 1e9, 100 * 2 * fl / ms / 1e9 / PEAK))


if __name__ == "__main__" and "sgemm" not in sys.argv:
    main()


def sgemm_bench(reps=10, arith="f32"):
    """DNN-head GEMM shapes of the north-star config (B=4096, 1280 -> 400 -> 400 -> 400 -> 1)."""
    dev = "cuda"
    B = int(os.environ.get("KBENCH_B", "4096"))
    shapes = []
    dims = [1280, 400, 400, 400]
    for li in range(3):
        K, N = dims[li], dims[li + 1]
        shapes.append(("fwd L%d" % li, 0, 1, B, N, K, (B, K), (N, K)))
        shapes.append(("wgrad L%d" % li, 1, 0, N, K, B, (B, N), (B, K)))
        shapes.append(("dgrad L%d" % li, 0, 0, B, K, N, (B, N), (N, K)))
    shapes.append(("fwd out", 0, 1, B, 1, 400, (B, 400), (1, 400)))
    shapes.append(("wgrad out", 1, 0, 1, 400, B, (B, 1), (B, 400)))
    shapes.append(("dgrad out", 0, 0, B, 400, 1, (B, 1), (1, 400)))
    tot = 0.0
    for name, ta, tb, M, N, K, sa, sb in shapes:
        A = torch.randn(*sa, device=dev)
        Bm = torch.randn(*sb, device=dev)
        C = torch.empty(M, N, device=dev)
        ms = timeit(lambda: ops.sgemm(ta, tb, M, N, K, A, sa[1], Bm, sb[1], C, N, arith=arith), reps)
        tot += ms
        print("sgemm %-6s %-10s M=%5d N=%5d K=%5d  %.4f ms  %.1f TFLOP/s" % (arith, name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
    print("sgemm %s total %.3f ms" % (arith, tot))


if __name__ == "__main__" and "sgemm" in sys.argv:
    sgemm_bench(arith="f32")
    sgemm_bench(arith="bf16x3")
