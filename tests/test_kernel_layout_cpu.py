"""CPU check of a layout rule the hardware taught in round 5: no scalar load of kernel arguments in the fast voxel-update kernels may straddle a 64-byte
line.  (A straddling s_load_dwordx8 of a run-time-indexed operator record gave lanes 48-63 of the first dependent vector instructions a stale value now and then -
run-to-run different voxels in the frame loop.)  The kernels are compiled to gfx950 assembly here (hipcc cross-compiles without a GPU) and every s_load is checked:
loads relative to the kernel-argument base (s[0:1], 64-byte aligned: .kernarg_segment_align) by their offset, loads relative to a computed record base (the batch's
operator records, 64-byte aligned by their type) likewise."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_fast_update_kernels_have_no_scalar_load_across_a_64_byte_line(tmp_path):
    out = tmp_path / "tsdf.s"
    from bundlefusion_amd.build import HIP_FLAGS
    cmd = [HIPCC] + [f for f in HIP_FLAGS if f != "-shared"] + ["--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "bundlefusion_amd", "csrc"),
           os.path.join(ROOT, "bundlefusion_amd", "csrc", "tsdf.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    checked = 0
    for m in re.finditer(r"\n(_ZN\S*?(k_update_apx|k_update_batch_apx)\S*):[^\n]*\n(.*?)s_endpgm", asm, re.S):
        name, body = m.group(1), m.group(3)
        loads = re.findall(r"s_load_dword(?:x(\d+))?\s+\S+,\s*s\[\d+:\d+\],\s*(0x[0-9a-f]+|\d+)", body)
        assert loads, name
        for n, off in loads:
            n, off = int(n or 1), int(off, 0)
            assert (off % 64) + 4 * n <= 64, "%s: s_load_dwordx%d at offset 0x%x straddles a 64-byte line" % (name, n, off)
        checked += 1
        align = re.search(r"\.amdhsa_kernel %s.*?\.end_amdhsa_kernel" % re.escape(name), asm, re.S)
        assert align is not None
    assert checked >= 8, "expected the six k_update_apx and the two k_update_batch_apx instantiations, found %d" % checked
    # the argument segment itself must be aligned to the line: the metadata carries the maximum alignment of the arguments
    entries = re.split(r"\n  - \.a", asm[asm.index("amdhsa.kernels:"):])
    seen = 0
    for e in entries:
        nm = re.search(r"\.name:\s+(\S+)", e)
        if nm and ("k_update_batch_apx" in nm.group(1) or "k_update_apx" in nm.group(1)):
            seg = re.search(r"\.kernarg_segment_align:\s*(\d+)", e)
            assert seg and int(seg.group(1)) >= 64, (nm.group(1), seg and seg.group(1))
            seen += 1
    assert seen >= 8


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_no_packed_fp32_result_is_read_without_a_wait_state():
    """tools/pk_hazard_scan.py over every csrc/*.hip with the library's build flags: no v_pk_add/mul/fma_f32 whose result the next instruction reads (with the
    flags of bundlefusion_amd/build.py there is no packed FP32 instruction at all).  The other half of DESIGN.md 7.2: the compiler pads that forwarding hazard
    only for some operand encodings, and the batched voxel update's run-to-run differences sat on the unpadded ones."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pk_hazard_scan.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "total 0" in r.stdout and r.stdout.count("packed FP32 instructions     0") >= 9, r.stdout
