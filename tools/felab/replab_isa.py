#!/usr/bin/env python3
"""Static instruction mix of the main loop of replab's three rate kernels (two butterflies per iteration), from the ISA hipcc -save-temps
leaves in _build/: python tools/felab/replab_isa.py tools/felab/_build/replab-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
names = {"0": "4 x 32 canonical (product: fe_mul_tw + fe_addsub)", "1": "FP64 3 x 43 (fd_mul_tw + 6 v_add_f64)", "2": "5 x 26 integers (fr_mul_tw + lazy add / sub)"}
for kind in "012":
    m = re.search(r"^_Z11rate_kernelILi%s.*?:\n(.*?)\n\s*s_endpgm" % kind, text, flags=re.S | re.M)
    body = m.group(1)
    # the main loop = the largest basic block that ends in a backward s_cbranch to its own label
    best = None
    for lm in re.finditer(r"^(\.LBB\d+_\d+):.*?\n(.*?)s_cbranch_\w+ \1\b", body, flags=re.S | re.M):
        blk = lm.group(2)
        if best is None or len(blk) > len(best):
            best = blk
    ins = [ln.split()[0] for ln in best.splitlines() if ln.strip() and not ln.strip().startswith((";", ".", "//")) and re.match(r"\s+[vs]_", ln)]
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    mads = sum(v for k, v in c.items() if k.startswith(("v_mad_u64_u32", "v_mad_i64_i32")))
    fma = sum(v for k, v in c.items() if k.startswith("v_fma_f64"))
    f64 = sum(v for k, v in c.items() if k.endswith("_f64"))
    print("%-52s loop body: %4d VALU (%3d v_mad_*64_*32, %3d v_fma_f64, %3d FP64 in all), %3d SALU, s_nop %d  -> %.1f VALU per butterfly"
          % (names[kind], valu, mads, fma, f64, sum(v for k, v in c.items() if k.startswith("s_") and k != "s_nop"), c.get("s_nop", 0), valu / 2.0))
    top = ", ".join("%s %d" % kv for kv in c.most_common(8))
    print("      " + top)
