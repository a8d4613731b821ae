cat > /tmp/t24.py <<'PY'
import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import distaff_amd as D
log_n = int(sys.argv[1])
W = 16
rng = np.random.default_rng(1)
cols = rng.integers(0, 2**63, size=(W, 1 << log_n, 2), dtype=np.uint64)
ctx = D.Context(log_n, W, 0, 0, log_blowup=4)
ctx.upload(cols)
ctx.commit_trace()
ctx.set_profiling(True); ctx.kernel_stats(reset=True)
t = time.time(); ctx.commit_trace(); dt = time.time() - t
st = ctx.kernel_stats(reset=True)
print(os.environ.get("DISTAFF_NTT_LDS", "reg"), log_n, round(dt * 1e3, 1), {k: round(v["ms"], 1) for k, v in st.items() if v["ms"] > 1})
PY
for ln in 23 24; do
python /tmp/t24.py $ln
DISTAFF_NTT_LDS=1 python /tmp/t24.py $ln
done
