set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python - <<'PY' 2>&1 | tail -40
import time, numpy as np
import distaff_amd as D, oracle as O
for log_n in (12, 16, 18, 20):
    t0=time.time(); cols, ph, res = D.fibonacci_trace(log_n); tg=time.time()-t0
    ctx = D.Context(log_n, 20, 1, 0)
    ctx.upload(cols)
    for rep in range(3):
        t0=time.time(); proof = ctx.prove([1,0],[res]); dt=time.time()-t0
        print("log_n=%d rep=%d prove %.1f ms  phases %s" % (log_n, rep, dt*1e3, ["%.1f"%x for x in ctx.phase_ms()]), flush=True)
        ctx.upload(cols)
    ok, err = O.verify(proof, ph, [1,0], [res])
    print("log_n=%d tracegen %.1fs proof %d bytes verify=%s %s cells/s=%.3e" % (log_n, tg, len(proof), ok, err, 20*(1<<log_n)/dt), flush=True)
    ctx.close()
PY
