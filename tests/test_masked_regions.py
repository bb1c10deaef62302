"""The guard for heavy code under a partial EXEC mask (tools/check_masked_regions.py; csrc/Makefile runs it on every build
of nmpc_block_kernel.hip and fails the build on a hit).  Here: the scan passes on the assembly of the library that is in the
tree, and it does catch a spill when one is planted.  CPU-only: hipcc cross-compiles without a GPU."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alore_legged_manipulator_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def block_kernel_assembly(td):
    built = os.path.join(CSRC, "build", "nmpc_block_kernel-hip-amdgcn-amd-amdhsa-gfx950.s")
    obj = os.path.join(CSRC, "build", "nmpc_block_kernel.o")
    src = os.path.join(CSRC, "nmpc_block_kernel.hip")
    if os.path.exists(built) and os.path.exists(obj) and os.path.getmtime(built) >= os.path.getmtime(src):
        return built            # what the object in the tree was assembled from
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("FLAGS"):
            flags = line.split(":=", 1)[1].split()
    assert flags, "FLAGS line of the csrc Makefile not found"
    flags = [f.replace("$(ARCH)", "gfx950") for f in flags if f != "-fPIC"]
    out = os.path.join(td, "blk.s")
    subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True, timeout=900)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_spill_traffic_inside_the_masked_backward_sweep():
    from check_masked_regions import check
    with tempfile.TemporaryDirectory() as td:
        path = block_kernel_assembly(td)
        assert check(path) == []
        # the scan is not vacuous: a reload planted behind the first exec save of a sweep window is found
        lines = open(path).read().splitlines()
        inside, planted = False, False
        for i, ln in enumerate(lines):
            if re.match(r"^_ZN4nmpc16rti_block_kernelILi16ELi2E\S+:", ln):
                inside = True
            if inside and "s_setprio 3" in ln:
                for k in range(i, len(lines)):
                    if "s_and_saveexec_b64" in lines[k]:
                        lines.insert(k + 1, "\tv_accvgpr_read_b32 v1, a0")
                        planted = True
                        break
                break
        assert planted
        bad = os.path.join(td, "planted.s")
        open(bad, "w").write("\n".join(lines))
        assert any("register-file traffic" in f for f in check(bad))


def test_the_makefile_runs_the_scan():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "check_masked_regions.py $(if $(STRICT_MASK_CHECK),--strict) $(BLK_ASM) || (rm -f $@; exit 1)" in mk
