import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import distaff_amd as D
cols, ph, result = D.fibonacci_trace(20)
ctx = D.Context(20, 20, 1, 0); ctx.upload(cols)
for prof in (False, True, False, True):
    ctx.set_profiling(prof); ctx.kernel_stats(reset=True)
    ctx.prove([1, 0], [result]); ctx.kernel_stats(reset=True)
    t = time.perf_counter()
    for _ in range(5):
        ctx.prove([1, 0], [result])
    dt = (time.perf_counter() - t) / 5
    ctx.kernel_stats(reset=True)
    print("profiling", prof, round(dt * 1e3, 3), "ms")
