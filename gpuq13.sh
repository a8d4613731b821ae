cat > /tmp/t20.py <<'PY'
import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import distaff_amd as D
log_n = 20
W = 20
rng = np.random.default_rng(1)
cols = rng.integers(0, 2**63, size=(W, 1 << log_n, 2), dtype=np.uint64)
ctx = D.Context(log_n, W, 1, 0)
ctx.upload(cols)
ctx.commit_trace()
ctx.set_profiling(True); ctx.kernel_stats(reset=True)
ctx.commit_trace()
st = ctx.kernel_stats(reset=True)
print(os.environ.get("DISTAFF_NTT_DEBUG", "0"), {k: round(v["ms"], 2) for k, v in st.items() if v["ms"] > 0.5})
PY
for d in 0 1 2 3; do DISTAFF_NTT_DEBUG=$d python /tmp/t20.py 2>/dev/null; done
