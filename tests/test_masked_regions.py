"""Build-time guard for the one place where the stage-block RTI kernel runs heavy code under a partial EXEC mask.

Values that are live ACROSS a divergent region in the lanes that sit it out are only safe if the register allocator
puts no spill / reload / AGPR copy of them inside the region (such a copy executes for the active lanes only; the
matching one outside restores garbage in the others -- seen in ltv_mpc.hip, whose lane-by-lane sweeps therefore run
unpredicated).  rti_block_kernel<16, 2, ...> keeps the masked form of its backward sweep (a tenth fewer instructions);
this test compiles the kernel file to gfx950 assembly and fails if any register-file traffic of that kind appears
between the exec-mask save and restore of the sweep (bracketed by s_setprio 3 / s_setprio 0).  CPU-only: hipcc
cross-compiles without a GPU."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alore_legged_manipulator_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
BAD = re.compile(r"\b(v_accvgpr_(read|write|mov)|scratch_(load|store)|buffer_(load|store)_dword\S*\s.*\boffen\b|v_readlane_b32|v_writelane_b32)")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_spill_traffic_inside_the_masked_backward_sweep():
    flags = None
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("FLAGS"):
            flags = line.split(":=", 1)[1].split()
    assert flags, "FLAGS line of the csrc Makefile not found"
    flags = [f.replace("$(ARCH)", "gfx950") for f in flags if f != "-fPIC"]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "blk.s")
        subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", os.path.join(CSRC, "nmpc_block_kernel.hip"), "-o", out],
                       check=True, capture_output=True, timeout=900)
        text = open(out).read().splitlines()
    kernels = {}
    name = None
    for ln in text:
        m = re.match(r"^(_ZN4nmpc16rti_block_kernel\S+):", ln)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name is not None:
            kernels[name].append(ln)
            if "s_endpgm" in ln:
                name = None
    masked = {k: v for k, v in kernels.items() if "ILi16ELi2E" in k}
    assert len(masked) == 3, sorted(kernels)          # diag / no-diag / stamped instantiations of (16, 2)
    for k, lines in masked.items():
        windows, cur = [], None
        for ln in lines:
            if "s_setprio 3" in ln:
                cur = []
            elif "s_setprio 0" in ln and cur is not None:
                windows.append(cur)
                cur = None
            elif cur is not None:
                cur.append(ln)
        assert windows, k
        for w in windows:
            idx_save = [i for i, ln in enumerate(w) if "s_and_saveexec_b64" in ln]
            idx_rest = [i for i, ln in enumerate(w) if re.search(r"s_or_b64 exec, exec", ln)]
            assert idx_save and idx_rest, k
            body = w[idx_save[0]:idx_rest[-1] + 1]
            assert sum(1 for ln in body if re.search(r"\bv_(fma|fmac|mul|add)_f32", ln)) > 150, "the Riccati steps are not inside the window?"
            bad = [ln.strip() for ln in body if BAD.search(ln)]
            assert not bad, (k, bad[:5])
