"""Build tests/emu/librat_emu.so: the SAME kernel sources as librat_hip.so, compiled for the host against
hip_emu.h (one OS thread per GPU thread).  Test infrastructure only — lets the CPU test-suite exercise the
kernels' index arithmetic, LDS layouts and MFMA lane maps without a GPU."""
import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "www24-rat_amd", "csrc")
LIB = os.path.join(HERE, "librat_emu.so")


def build(force=False):
    # RAT_EMU_CXXFLAGS (e.g. "-DRAT_FWD_BOUND"): an experiment's compile-time variant of the kernels -> its own library file
    extra = os.environ.get("RAT_EMU_CXXFLAGS", "").split()
    if extra:
        return _build(True, extra, os.path.join(HERE, "librat_emu_variant.so"), os.path.join(HERE, "obj_variant"))
    return _build(force, [], LIB, os.path.join(HERE, "obj"))


def _build(force, extra, LIB, objdir):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "hip_emu.h"), os.path.join(HERE, "hip_emu.cpp"),
                                                                   os.path.join(ROOT, "include", "rat_hip.h")]
    h = hashlib.sha256()
    for p in deps:
        h.update(open(p, "rb").read())
    stamp = LIB.replace(".so", ".digest")
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return LIB
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs + [os.path.join(HERE, "hip_emu.cpp")]:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = ["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-DRAT_EMU", "-I", HERE, "-I", CSRC, "-x", "c++", "-c", s, "-o", o,
               "-Wno-attributes", "-ffp-contract=off"] + extra
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("emu compile failed for %s:\n%s" % (s, out))
    r = subprocess.run(["g++", "-shared", "-o", LIB] + objs + ["-lpthread", "-latomic"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emu link failed:\n" + r.stdout + r.stderr)
    open(stamp, "w").write(h.hexdigest())
    return LIB


if __name__ == "__main__":
    print(build(force=True))
