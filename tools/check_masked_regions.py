#!/usr/bin/env python3
"""Build-time guard for the one place where the stage-block RTI kernel runs heavy code under a partial EXEC mask.

Values that are live ACROSS a divergent region in the lanes that sit it out are only safe if the register allocator
puts no spill / reload / AGPR copy of them inside the region (such a copy executes for the active lanes only; the
matching one outside restores garbage in the others -- seen in ltv_mpc.hip, whose lane-by-lane sweeps therefore run
unpredicated).  rti_block_kernel<16, 2, ...> and <32, 1, ...> keep the masked form of the backward sweep (a tenth fewer
instructions).  This script reads the gfx950 assembly the build has just produced for nmpc_block_kernel.hip
(csrc/Makefile: -save-temps=obj) and fails if any register-file traffic of that kind appears between the exec-mask save and
restore of the sweep (bracketed by s_setprio 3 / s_setprio 0).  The Makefile runs it after compiling the file and deletes
the object on failure, so a compiler that allocates differently cannot ship the masked form unnoticed;
tests/test_masked_regions.py runs it again on the same file.

usage: check_masked_regions.py [--strict] <nmpc_block_kernel gfx950 .s file>"""
import re
import sys

# (v_readlane_b32 / v_writelane_b32 -- the spill and reload of a SCALAR register into a lane of a vector register -- ignore EXEC by
# definition (CDNA ISA: "ignores exec mask"), so they are safe under a partial mask and are not findings)
BAD = re.compile(r"\b(v_accvgpr_(read|write|mov)|scratch_(load|store)|buffer_(load|store)_dword\S*\s.*\boffen\b)")


def check(path):
    """returns a list of findings (empty = the masked regions are clean)"""
    text = open(path).read().splitlines()
    kernels, name = {}, None
    for ln in text:
        m = re.match(r"^(_ZN4nmpc(?:16rti_block_kernel|24rti_block_sampler_kernel)\S+):", ln)  # the second kind includes the same body (nmpc_block_body.inc)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name is not None:
            kernels[name].append(ln)
            if ln.startswith(".Lfunc_end"):  # not the first s_endpgm: a kernel may leave early on another path (the sampler's workgroups)
                name = None
    masked = {k: v for k, v in kernels.items() if "ILi16ELi2E" in k or "ILi32ELi1E" in k}
    if not masked:  # whatever set of (16, 2) / (32, 1) instantiations the file holds is checked; none at all means the scan found nothing
        return [f"no instantiation of (16, 2) or (32, 1) found among {sorted(kernels)}"]
    findings = []
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
        if not windows:
            findings.append(f"{k}: no s_setprio 3 .. s_setprio 0 window")
            continue
        for w in windows:
            # lines under a saved exec mask, in layout order (the compiler may rotate the loop so that a restore precedes
            # its save in layout: then everything after the save up to the end of the window is the masked body)
            if not (any("s_and_saveexec_b64" in ln for ln in w) and any(re.search(r"s_or_b64 exec, exec", ln) for ln in w)):
                findings.append(f"{k}: no exec save / restore inside the sweep window")
                continue
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
            # multiply-adds inside the window (a packed instruction counts for its two)
            if sum((2 if "v_pk_" in ln else 1) for ln in body if re.search(r"\bv_(pk_)?(fma|fmac|mul|add)_f32", ln)) <= need:
                findings.append(f"{k}: the Riccati steps are not inside the masked window?")
            bad = [ln.strip() for ln in body if BAD.search(ln)]
            if bad:
                findings.append(f"{k}: register-file traffic under a partial EXEC mask: {bad[:5]}")
    return findings


def is_hazard(finding):
    """the finding the scan exists for (spill / reload traffic under a partial EXEC mask: wrong results); the others say that the
    scan did not recognise the shape of the code -- another compiler version lays the sweep out differently -- and nothing about
    the code itself"""
    return "register-file traffic" in finding


if __name__ == "__main__":
    # exit status 1 (the Makefile then deletes the object): on a hazard always; on a shape the scan does not recognise only with
    # --strict (csrc/Makefile: STRICT_MASK_CHECK=1, what tests/test_masked_regions.py asserts), otherwise that is a warning
    args = [a for a in sys.argv[1:] if a != "--strict"]
    strict = "--strict" in sys.argv[1:]
    if len(args) != 1:
        sys.exit(__doc__)
    f = check(args[0])
    for line in f:
        print("check_masked_regions:", ("" if (strict or is_hazard(line)) else "warning: ") + line, file=sys.stderr)
    sys.exit(1 if any(strict or is_hazard(line) for line in f) else 0)
