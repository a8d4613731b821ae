#!/usr/bin/env python3
"""Device time of the two transform passes of TraceTable::extend (trace_table.rs:143-169) on random columns, for whatever library
DISTAFF_HIP_LIB names (laboratory builds of tools/r6_lde_lab.sh produce wrong extensions on purpose: nothing is checked here).
usage: python tools/r6_lde_lab.py <name> <log_n> [repetitions] [random|zeros|ones]   ->   one line: name log_n pass_a_ms pass_b_ms leaves_ms per commit
`zeros` / `ones`: constant columns instead of random ones -- the same instructions on data that toggles nothing: if the passes run faster on
them, their time depends on the DATA (power / clocks), not only on the instruction stream."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import distaff_amd as D

name, log_n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W = 20
data = sys.argv[4] if len(sys.argv) > 4 else "random"
rng = np.random.default_rng(1)
cols = rng.integers(0, 2**63, size=(W, 1 << log_n, 2), dtype=np.uint64)
if data == "zeros":
    cols[:] = 0
elif data == "ones":
    cols[:] = 0; cols[:, :, 0] = 1
ctx = D.Context(log_n, W, 1, 0)
ctx.upload(cols)
ctx.commit_trace()
ctx.set_profiling(2); ctx.kernel_stats(reset=True)
for _ in range(reps):
    ctx.upload(cols)            # coset 0 of the extension is the trace buffer: a commit interpolates in place
    ctx.commit_trace()
st = ctx.kernel_stats(reset=True)
ph = ctx.phase_ms()
print(name, log_n, " ".join("%.3f" % (st.get(k, {"ms": 0})["ms"] / reps) for k in ("ntt_pass_a", "ntt_pass_b", "trace_leaves_kernel")), "lde_phase %.3f" % ph[0], flush=True)
ctx.close()
