#!/usr/bin/env python3
"""Diagnostic: build www24-rat_amd/lib/librat_<name>.so with SEVERAL sources recompiled under extra flags (same-box A/Bs).
    python tools/variant_multi.py ntboth gather.hip:-DRAT_GATHER_NT optim.hip:-DRAT_OPT_NT
Every other object comes from the regular build (lib/obj)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "www24-rat_amd")
sys.path.insert(0, PKG)
import build as rb  # noqa: E402


def main():
    name, specs = sys.argv[1], sys.argv[2:]
    rb.build()
    objdir = os.path.join(rb.LIBDIR, "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    replaced = {}
    for spec in specs:
        src, _, flags = spec.partition(":")
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [rb.HIPCC] + rb.FLAGS + rb.FILE_FLAGS.get(src, []) + flags.split(",") + ["-c", os.path.join(rb.CSRC, src), "-o", obj]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        replaced[src.replace(".hip", ".o")] = obj
    objs = [replaced.get(os.path.basename(o), o) for o in sorted(os.path.join(rb.LIBDIR, "obj", f) for f in os.listdir(os.path.join(rb.LIBDIR, "obj")))]
    out = os.path.join(rb.LIBDIR, "librat_%s.so" % name)
    subprocess.run([rb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    print(out)


if __name__ == "__main__":
    main()
