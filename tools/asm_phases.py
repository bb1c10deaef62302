#!/usr/bin/env python3
"""Static instruction table of one kernel of a gfx950 .s file, per basic block and per source phase.

The stage-block RTI kernel can be compiled with -DALORE_PHASE_MARKERS: every phase boundary of the source then leaves a
comment `; @phase <name>` in the assembly (an empty `asm volatile`; it constrains nothing but other volatile asm, so the
instruction totals are those of the shipped build to a few instructions -- compare the TOTAL line of both).  This script
splits the kernel into basic blocks, counts the instructions of every block by class and attributes a block to the
phase whose marker precedes it in layout order.  Loop trip counts are not known statically: give them with
--trips label=count (a block label as printed, e.g. .LBB26_14=4) to get a dynamic estimate; unlisted blocks count once
(or --default-trips for blocks of a named phase: phase:pred=3).

    python tools/asm_phases.py build/nmpc_block_kernel-hip-amdgcn-amd-amdhsa-gfx950.s ILi4ELi5ELb1ELb0ELb1ELb1E [--blocks]
"""
import argparse
import collections
import re
import sys

CLASSES = ("valu", "pk", "dpp", "accvgpr", "mov", "trans", "salu", "lds", "vmem", "scratch", "wait", "other")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_")


def classify(ins):
    """-> list of classes an instruction counts in (valu is the total of the vector ALU classes)"""
    op = ins.split()[0]
    if op.startswith("v_accvgpr"):
        return ["valu", "accvgpr"]
    if op.startswith("v_"):
        out = ["valu"]
        if op.startswith("v_pk_"):
            out.append("pk")
        if "dpp" in ins or "quad_perm" in ins or "row_" in ins or "wave_sh" in ins:
            out.append("dpp")
        if op in ("v_mov_b32_e32", "v_mov_b32_e64", "v_mov_b32", "v_pk_mov_b32", "v_mov_b64_e32", "v_mov_b64"):
            out.append("mov")
        if TRANS.match(op):
            out.append("trans")
        return out
    if op.startswith("scratch_"):
        return ["scratch"]
    if op.startswith(("global_", "buffer_", "flat_")):
        return ["vmem"]
    if op.startswith("ds_"):
        return ["lds"]
    if op == "s_waitcnt" or op.startswith("s_waitcnt"):
        return ["wait"]
    if op.startswith("s_"):
        return ["salu"]
    return ["other"]


def kernel_body(path, key):
    lines, name, body = open(path).read().splitlines(), None, []
    for ln in lines:
        m = re.match(r"^(_Z\S+):", ln)
        if name is None:
            if m and key in m.group(1):
                name = m.group(1)
            continue
        body.append(ln)
        if re.match(r"^\s+s_endpgm", ln) and False:
            break
        if ln.startswith(".Lfunc_end"):
            break
    if name is None:
        sys.exit(f"no kernel matching {key}")
    return name, body


def blocks_of(body):
    """-> list of dict(label, phase, counts, insts, term)"""
    out, cur, phase = [], None, "entry"

    def new(label):
        nonlocal cur
        cur = dict(label=label, phase=phase, counts=collections.Counter(), n=0, term="", markers=[])
        out.append(cur)

    new("<entry>")
    for ln in body:
        s = ln.strip()
        m = re.match(r"^(\.LBB\S+):", ln)
        if m:
            new(m.group(1))
            continue
        pm = re.match(r"^;\s*@phase\s+(\S+)", s)
        if pm:
            phase = pm.group(1)
            cur["markers"].append(phase)
            if cur["n"] == 0:
                cur["phase"] = phase
            else:  # a marker in the middle of a block: split so that the rest belongs to the new phase
                lab = cur["label"] + "+"
                new(lab)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        ins = s.split(";")[0].strip()
        if not ins:
            continue
        for c in classify(ins):
            cur["counts"][c] += 1
        cur["n"] += 1
        if re.match(r"^s_c?branch|^s_endpgm|^s_setpc", ins):
            cur["term"] = ins
    return [b for b in out if b["n"] > 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("key")
    ap.add_argument("--blocks", action="store_true", help="print every basic block")
    ap.add_argument("--trips", nargs="*", default=[], help="label=count or phase:<name>=count")
    a = ap.parse_args()
    name, body = kernel_body(a.asm, a.key)
    blocks = blocks_of(body)
    trips_l, trips_p = {}, {}
    for t in a.trips:
        k, v = t.split("=")
        if k.startswith("phase:"):
            trips_p[k[6:]] = float(v)
        else:
            trips_l[k] = float(v)
    print(f"# {name}")
    hdr = f"{'':28s}" + "".join(f"{c:>8s}" for c in CLASSES)
    if a.blocks:
        print(hdr)
        for b in blocks:
            t = trips_l.get(b["label"], trips_p.get(b["phase"], 1.0))
            print(f"{b['label'][:16]:16s}{b['phase'][:11]:>11s} " + "".join(f"{b['counts'][c]:8d}" for c in CLASSES) +
                  f"  x{t:g}  {b['term']}")
    per = collections.OrderedDict()
    dyn = collections.OrderedDict()
    for b in blocks:
        t = trips_l.get(b["label"], trips_p.get(b["phase"], 1.0))
        per.setdefault(b["phase"], collections.Counter()).update(b["counts"])
        d = dyn.setdefault(b["phase"], collections.Counter())
        for c, v in b["counts"].items():
            d[c] += v * t
    print("# static instructions per phase")
    print(hdr)
    tot = collections.Counter()
    for ph, c in per.items():
        print(f"{ph[:27]:28s}" + "".join(f"{c[k]:8d}" for k in CLASSES))
        tot.update(c)
    print(f"{'TOTAL':28s}" + "".join(f"{tot[k]:8d}" for k in CLASSES))
    if a.trips:
        print("# dynamic estimate per wavefront (trip counts as given)")
        print(hdr)
        tot = collections.Counter()
        for ph, c in dyn.items():
            print(f"{ph[:27]:28s}" + "".join(f"{c[k]:8.0f}" for k in CLASSES))
            tot.update(c)
        print(f"{'TOTAL':28s}" + "".join(f"{tot[k]:8.0f}" for k in CLASSES))


if __name__ == "__main__":
    main()
