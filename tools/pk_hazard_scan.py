#!/usr/bin/env python3
"""Compile every csrc/*.hip to gfx950 assembly and list the packed-FP32 instructions (v_pk_add/mul/fma_f32) whose result the NEXT instruction reads with no wait state
in between - the pattern behind the batched voxel update's run-to-run differences in round 5 (profiles/r05_determinism.md).  With the library's build flags
(bundlefusion_amd/build.py: packed FP32 off) there is no packed instruction at all; `--packed` compiles without that flag and shows what the compiler emits otherwise.
    python tools/pk_hazard_scan.py [--packed]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bundlefusion_amd.build import HIP_FLAGS


def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm):
    found, npk = [], len(re.findall(r"\n\s*v_pk_(?:add|mul|fma)_f32", asm))
    for m in re.finditer(r"\n(_Z\S+):[^\n]*\n(.*?)s_endpgm", asm, re.S):
        ins = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
        for k in range(1, len(ins)):
            p, c = ins[k - 1], ins[k]
            if not re.match(r"v_pk_(add|mul|fma)_f32", p) or not c.startswith("v_"):
                continue
            dst = regs(p.split(None, 1)[1].split(",")[0])
            src = set()
            for t in c.split(None, 1)[1].split(",")[1:]:
                if t.strip():
                    src |= regs(t.strip().split()[0])
            if dst & src:
                found.append((m.group(1), p, c))
    return npk, found


def main():
    packed = "--packed" in sys.argv
    flags = [f for f in HIP_FLAGS if f != "-shared"]
    if packed:
        i = flags.index("-packed-fp32-ops")
        del flags[i - 3:i + 1]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src_dir = os.path.join(ROOT, "bundlefusion_amd", "csrc")
    total = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(src_dir)):
            if not f.endswith(".hip"):
                continue
            out = os.path.join(tmp, f + ".s")
            r = subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"), "-I" + src_dir, os.path.join(src_dir, f), "-o", out], capture_output=True, text=True)
            if r.returncode != 0:
                print(f, "compile failed:", r.stderr[-500:]); continue
            npk, found = scan(open(out).read())
            total += len(found)
            print("%-14s packed FP32 instructions %5d, results read by the next instruction without a wait state %3d" % (f, npk, len(found)))
            for k, p, c in found[:3]:
                print("      %s:  %s  ->  %s" % (k[:40], p[:64], c[:64]))
    print("total", total)
    return 0 if (packed or total == 0) else 1


if __name__ == "__main__":
    sys.exit(main())
