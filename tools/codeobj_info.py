#!/usr/bin/env python3
"""What the gfx950 code objects inside a HIP shared library or executable say about every kernel: bytes of machine code (.text symbol
size), VGPRs / AGPRs / SGPRs, spilled registers, scratch (private segment) and LDS bytes -- read from the ELF symbol table and the
AMDGPU metadata note of each bundled code object (llvm-objdump --offloading, llvm-readelf).  No GPU needed.

A CU pair of gfx950 shares a 64 KiB instruction cache; a straight-line kernel larger than that streams its code from L2 once per
wavefront (tools/icache_probe measures what that costs).  build() (__graft_entry__.py) writes this table into
distaff_amd/_build_info.json and fails when a kernel of the product's bench path is above the limit or spills.

    python tools/codeobj_info.py distaff_amd/libdistaff_hip.so [--min-bytes 8192]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    if not names:
        return {}
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names).encode(), stdout=subprocess.PIPE).stdout.decode().split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def short_name(demangled):
    """'void air_kernel<2, 1, 4, 8, 0, 240u, 6>(AirArgs)' -> 'air_kernel<2,1,4,8,0,240,6>' (the names the library's kernel statistics use)"""
    s = re.sub(r"^void\s+", "", demangled)
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):                       # strip the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    s = s[:cut].replace(" ", "")
    s = re.sub(r"\(unsignedint\)(\d+)", r"\1", s)
    s = re.sub(r"(\d+)u\b", r"\1", s)                # unsigned template arguments print as 240u
    return s.replace("true", "1").replace("false", "0")


def kernels_of(path):
    """-> {short kernel name: {"code_bytes", "vgpr", "agpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch_bytes", "lds_bytes"}}"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "obj")
        shutil.copy(path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            co = os.path.join(tmp, f)
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], stdout=subprocess.PIPE, check=True).stdout.decode()
            sizes = {}
            for line in syms.splitlines():
                p = line.split()
                if len(p) >= 8 and p[3] == "FUNC":
                    sizes[p[7]] = int(p[2])
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, check=True).stdout.decode()
            # the metadata note is YAML: one block per kernel, fields at a fixed indentation
            for block in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
                block = ".agpr_count:" + block
                fld = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
                sym = fld.get("name")
                if not sym:
                    continue
                sym = sym.strip("'\"")
                out[sym] = {"code_bytes": sizes.get(sym, 0), "vgpr": int(fld.get("vgpr_count", 0)), "agpr": int(fld.get("agpr_count", 0)),
                            "sgpr": int(fld.get("sgpr_count", 0)), "vgpr_spill": int(fld.get("vgpr_spill_count", 0)),
                            "sgpr_spill": int(fld.get("sgpr_spill_count", 0)), "scratch_bytes": int(fld.get("private_segment_fixed_size", 0)),
                            "lds_bytes": int(fld.get("group_segment_fixed_size", 0))}
    names = demangle(list(out))
    return {short_name(names[s]): v for s, v in out.items()}


def main():
    path = sys.argv[1]
    min_bytes = int(sys.argv[sys.argv.index("--min-bytes") + 1]) if "--min-bytes" in sys.argv else 0
    ks = kernels_of(path)
    print("%-64s %9s %5s %5s %6s %8s %6s" % ("kernel", "code B", "vgpr", "sgpr", "spill", "scratch", "lds"))
    for name, k in sorted(ks.items(), key=lambda kv: -kv[1]["code_bytes"]):
        if k["code_bytes"] >= min_bytes:
            print("%-64s %9d %5d %5d %6d %8d %6d" % (name[:64], k["code_bytes"], k["vgpr"], k["sgpr"], k["vgpr_spill"] + k["sgpr_spill"], k["scratch_bytes"], k["lds_bytes"]))


if __name__ == "__main__":
    main()
