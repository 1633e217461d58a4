#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of one csrc/*.hip file, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python tools/kernel_resources.py bundlefusion_amd/csrc/tsdf.hip [name-filter] > profiles/rNN_resources_tsdf.md

Waves per SIMD follow MI355X_MICROARCH.md "Register files": allocation granule 8, min(8, floor(512 / alloc)).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(n):
    r = subprocess.run(["c++filt", n], stdout=subprocess.PIPE)
    s = r.stdout.decode().strip()
    s = s.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", re.sub(r"^void ", "", s))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
           "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: +([A-Za-z][A-Za-z /\[\]]*?): +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}; rows.append(cur)
        elif cur is not None:
            cur[k] = v
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B | LDS B | waves/SIMD (hipcc) | waves/SIMD by registers |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = demangle(r["name"])
        if flt and flt not in name:
            continue
        v = int(r.get("VGPRs", 0)); a = int(r.get("AGPRs", 0))
        alloc = ((v + a + 7) // 8) * 8
        wv = min(8, 512 // alloc) if alloc else 8
        print("| `%s` | %d | %d | %s | %s | %s | %s | %d |" % (name, v, a, r.get("TotalSGPRs", r.get("SGPRs", "?")), r.get("ScratchSize [bytes/lane]", "0"),
                                                        r.get("LDS Size [bytes/block]", "0"), r.get("Occupancy [waves/SIMD]", "?"), wv))


if __name__ == "__main__":
    main()
