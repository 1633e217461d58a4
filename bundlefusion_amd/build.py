"""Build recipes: the gfx950 shared library (product) and the CPU oracle (test infrastructure).

`hipcc --offload-arch=gfx950` cross-compiles without a GPU; outputs stay in-tree so they
travel to the GPU box with the repo snapshot (they are git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bundlefusion_amd", "csrc")
LIB_DIR = os.path.join(ROOT, "bundlefusion_amd", "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbf_hip.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # op-by-op IEEE arithmetic, bit-comparable with the oracle
    "-fvisibility=hidden", "-Wno-unused-value", "-Wno-unused-result",
    # No packed-FP32 instructions (v_pk_add/mul/fma_f32): the compiler pads the gfx950 forwarding hazard behind such an instruction with s_nop ONLY when the
    # instruction's first source has op_sel_hi set; where that source is a broadcast scalar (op_sel_hi:[0,1]) the very next instruction may read the result with no
    # wait state - and on the device, with other kernels running, lanes 48-63 then now and then read the previous value (round 5: the batched voxel update's
    # run-to-run differences sat exactly on the three such places of k_update_batch_apx, all in the first voxel pair's depth; every .hip file has some: tools/
    # pk_hazard_scan.py, profiles/r05_determinism.md).  The two halves as two scalar instructions are the same IEEE operations bit for bit; the voxel update
    # is 7 % longer in instructions and 12 registers smaller.  (The host pass prints "not a recognized feature for this target": expected.)
    "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
]


def _sources(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def build_lib(force=False, verbose=False):
    """Compile every HIP/C++ source under csrc/ into bundlefusion_amd/lib/libbf_hip.so."""
    srcs = _sources(CSRC, (".hip", ".cpp"))
    deps = srcs + _sources(CSRC, (".h",)) + _sources(os.path.join(ROOT, "include"), (".h",))
    if not force and not _stale(LIB_PATH, deps):
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + [d for d in deps if d.endswith(".h")]):
            cmd = [hipcc] + [f for f in HIP_FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-o", LIB_PATH]      # zlib: depth frames of .sens files
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    return LIB_PATH


def build_probe(force=False):
    """tools/probe/libbf_probe.so: memory-system probe kernels for the measurement tools (tools/hbm_block_probe.py, tools/pmc_calibrate.py) - NOT part of the product library."""
    src = os.path.join(ROOT, "tools", "probe", "probe.hip")
    out = os.path.join(ROOT, "tools", "probe", "libbf_probe.so")
    if force or _stale(out, [src]):
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", src, "-o", out],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("probe build failed:\n" + r.stdout.decode())
    # tools/probe/bf_hazards: the per-mechanism hazard reproducers of round 6 (an executable; profiles/r06_determinism.md)
    hsrc, hexe = os.path.join(ROOT, "tools", "probe", "hazards.hip"), os.path.join(ROOT, "tools", "probe", "bf_hazards")
    if force or _stale(hexe, [hsrc, os.path.join(ROOT, "tools", "probe", "window_gen.h")]):
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", hsrc, "-o", hexe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("hazard probe build failed:\n" + r.stdout.decode())
    return out


def build_examples(force=False):
    """bundlefusion_amd/lib/class_surface_bench: the reference's frame loop against include/bundlefusion/bundlefusion.hpp (plain g++, no HIP headers), run by bench.py"""
    src = os.path.join(ROOT, "examples", "class_surface_bench.cpp")
    exe = os.path.join(LIB_DIR, "class_surface_bench")
    if force or _stale(exe, [src, os.path.join(ROOT, "include", "bundlefusion", "bundlefusion.hpp"), LIB_PATH]):
        r = subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-L", LIB_DIR, "-lbf_hip", "-Wl,-rpath,$ORIGIN", "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("class_surface_bench build failed:\n" + r.stdout.decode())
    return exe


def build_oracle(force=False):
    """Compile the CPU oracle (oracle/*.cpp) into oracle/_build/liboracle.so via its Makefile."""
    args = ["make", "-s", "-C", ORACLE_DIR]
    if force:
        args.append("-B")
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout.decode())
    return ORACLE_LIB


def build_ref():
    """Compile oracle/_ref/libbfref.so — the reference's own device code for the pinned stages, built for the host from
    /root/reference by oracle/ref/Makefile (test infrastructure).  Where the reference is absent (the GPU box) the prebuilt
    library that travelled with the snapshot is kept; returns None when there is neither."""
    ref_lib = os.path.join(ORACLE_DIR, "_ref", "libbfref.so")
    if os.path.isdir("/root/reference/FriedLiver/Source"):
        r = subprocess.run(["make", "-s", "-C", os.path.join(ORACLE_DIR, "ref")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("oracle/_ref build failed:\n" + r.stdout.decode())
    return ref_lib if os.path.exists(ref_lib) else None


if __name__ == "__main__":
    print(build_lib(force="-f" in sys.argv, verbose=True))
    print(build_probe(force="-f" in sys.argv))
    print(build_examples(force="-f" in sys.argv))
    print(build_oracle(force="-f" in sys.argv))
    print(build_ref())
