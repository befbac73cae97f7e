#!/usr/bin/env python3
"""Build librat_hip.so: hipcc --offload-arch=gfx950 over csrc/*.hip (cross-compiles without a GPU).

    python www24-rat_amd/build.py [--force] [--verbose]

The library is built IN-TREE (www24-rat_amd/lib/librat_hip.so) so that it travels with the repo snapshot to
the GPU box; it is git-ignored.
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librat_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
# per-file scheduler strategy, chosen by same-box A/B (tools/ab_bench.sh): the max-ILP strategy is 1-2.5 % faster on the FFN
# kernels and 6-15 % slower on the attention kernels
FILE_FLAGS = {"ffn.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "rat_hip.h")]:
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    h.update((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    return h.hexdigest()


def build(force=False, verbose=False, prof=False):
    """prof=True builds the DIAGNOSTIC variant librat_hip_prof.so (-DRAT_PROF: in-kernel phase stamps; used only by
    tools/phase_profile.py, never by the product or the tests)."""
    os.makedirs(LIBDIR, exist_ok=True)
    lib = os.path.join(LIBDIR, "librat_hip_prof.so") if prof else LIB
    flags = FLAGS + (["-DRAT_PROF"] if prof else [])
    stamp = lib.replace(".so", ".digest")
    dig = _digest() + ("-prof" if prof else "")
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return lib
    objdir = os.path.join(LIBDIR, "obj_prof" if prof else "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        cmd = [HIPCC] + flags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose:
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, prof="--prof" in sys.argv))
