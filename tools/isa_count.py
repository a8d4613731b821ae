#!/usr/bin/env python3
"""Static instruction mix of the kernels in a gfx950 assembly file (hipcc --cuda-device-only -S): per kernel the number of VALU,
SALU, VMEM, LDS instructions and the most frequent VALU opcodes.  The hot kernels are bound by VALU issue (profiles/README.md),
so for straight-line kernels (the constraint kernels) the static VALU count is a usable proxy for their time when no GPU is at
hand; loops are counted once.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S distaff_amd/csrc/kernels_air_sd4.hip -o /tmp/air.s
    python tools/isa_count.py /tmp/air.s [--top 12]
"""
import collections
import re
import subprocess
import sys


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
    except OSError:
        return name


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 10
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = kernels.setdefault(m.group(1), collections.Counter())
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        m = re.match(r"^\t([a-z_0-9]+)", line)
        if m and not m.group(1).startswith("."):
            cur[m.group(1)] += 1
    for name, ops in kernels.items():
        if not ops:
            continue
        valu = sum(v for k, v in ops.items() if k.startswith("v_"))
        salu = sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith(("s_nop", "s_waitcnt", "s_load", "s_buffer")))
        nops = ops.get("s_nop", 0)
        vmem = sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
        lds = sum(v for k, v in ops.items() if k.startswith("ds_"))
        mads = ops.get("v_mad_u64_u32", 0)
        print("%s\n  VALU %d (v_mad_u64_u32 %d)  SALU %d  s_nop %d  VMEM %d  LDS %d  scratch %d" % (
            demangle(name), valu, mads, salu, nops, vmem, lds, sum(v for k, v in ops.items() if k.startswith("scratch_"))))
        print("  " + ", ".join("%s %d" % kv for kv in sorted(((k, v) for k, v in ops.items() if k.startswith("v_")), key=lambda kv: -kv[1])[:top]))


if __name__ == "__main__":
    main()
