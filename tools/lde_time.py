#!/usr/bin/env python3
"""Times steps 1-2 (interpolation, LDE, leaf hashing, tree) on random columns.  usage: python tools/lde_time.py log_n log_blowup [W]
DISTAFF_NTT=reg|lds selects the NTT kernel family."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import distaff_amd as D

log_n, log_b = int(sys.argv[1]), int(sys.argv[2])
W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(1)
cols = rng.integers(0, 2**63, size=(W, 1 << log_n, 2), dtype=np.uint64)
ctx = D.Context(log_n, W, 1, 0, log_blowup=log_b)
ctx.upload(cols)
ctx.commit_trace()
ctx.set_profiling(True); ctx.kernel_stats(reset=True)
ctx.commit_trace()
st = ctx.kernel_stats(reset=True)
print(os.environ.get("DISTAFF_NTT", "auto"), "2^%d x %d, blowup %d:" % (log_n, W, 1 << log_b),
      {k: (round(v["ms"], 1), round(v["bytes"] / max(v["ms"], 1e-9) / 1e6)) for k, v in st.items() if v["ms"] > 0.5}, "(ms, GB/s algorithmic)")
ctx.close()
