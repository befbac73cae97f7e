"""Diagnostic: column-strip BatchNorm launches against the two-launch kernels (HIP events, 50 repetitions each)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "www24-rat_amd"))
from rat_amd import ops
from rat_amd._lib import get_lib

lib = get_lib()
dev = "cuda"


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N in (400, 1280):
    for M in (256, 512, 1024, 2048, 4096, 8192):
        z = torch.randn(M, N, device=dev)
        da = torch.randn(M, N, device=dev)
        g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
        dg, db, dbl = (torch.zeros(N, device=dev) for _ in range(3))
        a, sm, sr = ops.bn_act_fwd_strip(z, g, b, rm, rv, True, True, lib=lib)
        t_f = timeit(lambda: ops.bn_act_fwd_strip(z, g, b, rm, rv, True, True, lib=lib))
        t_f0 = timeit(lambda: ops.bn_relu_fwd(z, g, b, rm, rv, True, True, lib=lib))
        t_b = timeit(lambda: ops.bn_act_bwd_strip(z, a, da, g, sm, sr, dg, db, dbl, True, lib=lib))
        def old():
            dz = ops.bn_relu_bwd(z, a, da, g, sm, sr, dg, db, True, lib=lib)
            ops.colsum(dz, N, dbl, M, N, lib=lib)
        t_b0 = timeit(old)
        print("M %5d N %5d  fwd strip %6.1f us  two-launch %6.1f us | bwd strip %6.1f us  three-launch + colsum %6.1f us" % (M, N, t_f, t_f0, t_b, t_b0))
