#!/usr/bin/env python3
"""Compile ONE instantiation of nmpc::rti_block_kernel (seconds instead of the minute the whole file takes) and print its
resource usage and static instruction totals: the inner loop of kernel experiments.

    python tools/blk_fast.py "4, 5, true, false, true, true, false, true" [extra hipcc flags ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "alore_legged_manipulator_amd", "csrc", "nmpc_block_kernel.hip")
OUT = "/tmp/blk/fast"


def main():
    inst = sys.argv[1]
    extra = sys.argv[2:]
    s = open(SRC).read()
    head = s[:s.index("hipError_t launch_rti_block_group(")]
    head = head.replace('#include "nmpc_kernels.h"', f'#include "{ROOT}/alore_legged_manipulator_amd/csrc/nmpc_kernels.h"')
    head = head.replace('#include "nmpc_core.h"', f'#include "{ROOT}/alore_legged_manipulator_amd/csrc/nmpc_core.h"')
    tail = ("hipError_t launch_rti_block_group(const RtiParams& p, const RtiGroup& grp, const LaunchGeom& g, hipStream_t s)\n{\n"
            f"    const void* fn = (const void*)rti_block_kernel<{inst}>;\n"
            "    void* args[] = {const_cast<RtiParams*>(&p), const_cast<RtiGroup*>(&grp)};\n"
            "    return hipLaunchKernel(fn, dim3(g.grid), dim3(64), args, g.lds_bytes, s);\n}\n} // namespace nmpc\n")
    os.makedirs(OUT, exist_ok=True)
    open(os.path.join(OUT, "fast.hip"), "w").write(head + tail)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
           "-fPIC", "-save-temps=obj", "-c", os.path.join(OUT, "fast.hip"), "-o", os.path.join(OUT, "fast.o")] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OUT)
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    asm = os.path.join(OUT, "fast-hip-amdgcn-amd-amdhsa-gfx950.s")
    txt = open(asm).read()
    body = txt[txt.index("_ZN4nmpc16rti_block_kernel"):]
    vals = dict(re.findall(r"; (TotalNumSgprs|NumVgprs|NumAgprs|ScratchSize|codeLenInByte): *(\d+)", body))
    print(" ".join(f"{k}={v}" for k, v in vals.items()))
    sys.stdout.flush()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_phases.py"), asm, "rti_block_kernel"])


if __name__ == "__main__":
    main()
