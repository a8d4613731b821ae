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


def count_file(path):
    """-> {demangled kernel name: {"valu": n, "mad64": n, "s_nop": n}} for every kernel of a gfx950 assembly file"""
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
    out = {}
    for name, ops in kernels.items():
        if ops:
            out[demangle(name)] = {"valu": sum(v for k, v in ops.items() if k.startswith("v_")), "mad64": ops.get("v_mad_u64_u32", 0), "s_nop": ops.get("s_nop", 0)}
    return out


def air_key(demangled):
    """'void air_kernel<2, 1, 4, 8, 0, 240u, 6>(AirArgs)' -> 'air_kernel<2,1,4,8,0,240,6>' (the name the library's kernel statistics use)"""
    m = re.search(r"air_kernel<([^>]*)>", demangled)
    if not m:
        return None
    args = [a.strip() for a in m.group(1).split(",")]
    args = ["1" if a == "true" else "0" if a == "false" else re.sub(r"^(?:\(unsigned int\))?(\d+)u?$", r"\1", a) for a in args]
    return "air_kernel<" + ",".join(args) + ">"


def build_air_table(out_path, hipcc="/opt/rocm/bin/hipcc"):
    """Static per-point instruction counts of the constraint-kernel instances of the CURRENT sources (straight-line kernels: static =
    executed), written next to the shared library so that bench.py can price the constraint evaluation in multiply-adds."""
    import concurrent.futures
    import json
    import os
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    units = ["kernels_air_sd4", "kernels_air_small"]
    table = {}
    with tempfile.TemporaryDirectory() as tmp:
        def one(u):
            asm = os.path.join(tmp, u + ".s")
            subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                                   os.path.join(root, "distaff_amd", "csrc", u + ".hip"), "-o", asm], stderr=subprocess.DEVNULL)
            return count_file(asm)
        with concurrent.futures.ThreadPoolExecutor(2) as ex:
            for res in ex.map(one, units):
                for name, cnt in res.items():
                    k = air_key(name)
                    if k:
                        table[k] = cnt
    with open(out_path, "w") as fh:
        json.dump({"note": "static VALU / v_mad_u64_u32 instructions per evaluation point of each constraint-kernel instance (tools/isa_count.py)", "kernels": table}, fh, indent=1)
    return table


def main():
    if sys.argv[1] == "--air-table":
        print(build_air_table(sys.argv[2]))
        return
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
