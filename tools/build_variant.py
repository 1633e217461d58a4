#!/usr/bin/env python3
"""Build a VARIANT of libbf_hip.so beside the product library, for A/B runs on one GPU box (select it with BF_LIB_PATH=<path>):
    python tools/build_variant.py <name> [--packed] [--base <variant> | --base product] [--only a.hip,b.hip] [-DMACRO ...]
(--base: only the sources named by --only [default tsdf.hip] are compiled, the other objects are taken from that variant's build, or from the product's)
writes bundlefusion_amd/lib/variants/libbf_hip_<name>.so.  --packed drops the library's `-packed-fp32-ops` feature switch (the compiler's default code)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bundlefusion_amd import build as b


def main():
    name = sys.argv[1]
    argv = sys.argv[2:]
    base = None
    if "--base" in argv:
        i = argv.index("--base"); base = argv[i + 1]; del argv[i:i + 2]
    only = ["tsdf.hip"]
    if "--only" in argv:
        i = argv.index("--only"); only = argv[i + 1].split(","); del argv[i:i + 2]
    extra = [a for a in argv if a != "--packed"]
    flags = [f for f in b.HIP_FLAGS if f != "-shared"]
    if "--packed" in sys.argv:
        i = flags.index("-packed-fp32-ops")
        del flags[i - 3:i + 1]
    out_dir = os.path.join(b.LIB_DIR, "variants")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    srcs = b._sources(b.CSRC, (".hip", ".cpp"))
    procs, objs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        if base and os.path.basename(src) not in only:
            objs.append(os.path.join(b.LIB_DIR, "obj", os.path.basename(src) + ".o") if base == "product" else os.path.join(out_dir, "obj_" + base, os.path.basename(src) + ".o"))
            continue
        objs.append(obj)
        procs.append((src, subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + extra + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit("hipcc failed for %s:\n%s" % (src, out.decode()))
    lib = os.path.join(out_dir, "libbf_hip_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-o", lib])
    print(lib)


if __name__ == "__main__":
    main()
