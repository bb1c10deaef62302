"""Build-time guard for the one place where the stage-block RTI kernel runs heavy code under a partial EXEC mask.

Values that are live ACROSS a divergent region in the lanes that sit it out are only safe if the register allocator
puts no spill / reload / AGPR copy of them inside the region (such a copy executes for the active lanes only; the
matching one outside restores garbage in the others -- seen in ltv_mpc.hip, whose lane-by-lane sweeps therefore run
unpredicated).  rti_block_kernel<16, 2, ...> and <32, 1, ...> keep the masked form of its backward sweep (a tenth fewer instructions);
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
    masked = {k: v for k, v in kernels.items() if "ILi16ELi2E" in k or "ILi32ELi1E" in k}
    assert len(masked) == 10, sorted(kernels)         # diag / no-diag, each also as the single-iteration build, and the stamped one, of (16, 2) and (32, 1)
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
            # lines under a saved exec mask, in layout order (the compiler may rotate the loop so that a restore precedes
            # its save in layout: then everything after the save up to the end of the window is the masked body)
            assert any("s_and_saveexec_b64" in ln for ln in w) and any(re.search(r"s_or_b64 exec, exec", ln) for ln in w), k
            body, depth, at_label = [], 0, {}
            for ln in w:
                lab = re.match(r"^(\.LBB\S+):", ln)
                br = re.search(r"\bs_c?branch\S*\s+(\.LBB\S+)", ln)
                if lab and lab.group(1) in at_label:          # reached by a branch seen earlier: exec is what it was there
                    depth = min(depth, at_label[lab.group(1)])
                if br:
                    at_label[br.group(1)] = min(depth, at_label.get(br.group(1), depth))
                if re.search(r"s_(and|andn2)_saveexec_b64", ln):
                    depth += 1 if "s_and_saveexec" in ln else 0   # andn2 flips to the else side at the same depth
                elif re.search(r"s_or_b64 exec, exec", ln):
                    depth = max(0, depth - 1)
                elif depth > 0:
                    body.append(ln)
            need = 150 if "ILi16ELi2E" in k else 75   # two stage steps / one
            assert sum(1 for ln in body if re.search(r"\bv_(fma|fmac|mul|add)_f32", ln)) > need, "the Riccati steps are not inside the window?"
            bad = [ln.strip() for ln in body if BAD.search(ln)]
            assert not bad, (k, bad[:5])
