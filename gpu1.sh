set -x
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -15
python -m pytest tests -m gpu -x -q 2>&1 | tail -30
python - <<'PY' 2>&1 | tail -20
import distaff_amd as D
ctx = D.Context(10, 20, 1, 0)
for lanes in (1<<20, 1<<22):
    for it in (64, 256):
        ms = ctx.bench_mulmod(lanes, it)
        print("mulmod bench lanes=%d iters=%d: %.3f ms -> %.1f G mulmod/s" % (lanes, it, ms, lanes*it*4/ms/1e6))
PY
