#!/usr/bin/env python3
"""Where the cycles of a transform pass go.  rocprofv3's thread trace cannot be decoded in this image (the decoder
library is absent: profiles/r6_att_probe.txt), so the passes carry their own instrument in a LABORATORY build (-DNTT_LAB_STAMPS,
tools/ab_variant.sh stamps): every wavefront of 256 sampled workgroups notes the shader clock (s_memtime) at the boundaries of the phases of
a tile -- load issue | barrier | wait for the loads + LDS fill | barrier | five butterfly rounds | read-out -- for its first two tiles.

  on the GPU box:   DISTAFF_HIP_LIB=gpurun_tmp_libs/stamps/distaff_amd/libdistaff_hip.so python tools/r6_pass_stamps.py run <log_n> <out.json>
  anywhere:         python tools/r6_pass_stamps.py isa            static VALU / LDS / VMEM instructions between the stamps (from the same build's assembly)
  anywhere:         python tools/r6_pass_stamps.py table <out.json> [<isa.json>]      the table of profiles/r6_pass_stamps.md
"""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ORDER = [8, 9, 10, 11, 12, 0, 1, 2, 3, 4, 13]          # stamp indices in program order
PHASES = ["issue loads", "barrier 1 (previous tile left LDS)", "wait for loads + fill LDS", "barrier 2", "round 1", "round 2", "round 3", "round 4", "round 5",
          "read-out: table load, multiply, store (pass B: stores from round 5)"]


def run(log_n, out_path):
    import distaff_amd as D
    lib = D.load()
    rng = np.random.default_rng(1)
    cols = rng.integers(0, 2**63, size=(20, 1 << log_n, 2), dtype=np.uint64)
    ctx = D.Context(log_n, 20, 1, 0)
    ctx.upload(cols)
    ctx.commit_trace()
    count = 2 * 256 * 16 * 2 * 16
    buf = np.zeros(count, dtype=np.uint64)
    assert lib.dst_lab_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(count), 1) == 0
    ctx.upload(cols)
    ctx.set_profiling(2); ctx.kernel_stats(reset=True)
    ctx.commit_trace()
    st = ctx.kernel_stats(reset=True)
    assert lib.dst_lab_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(count), 0) == 0
    a = buf.reshape(2, 256, 16, 2, 16).astype(np.int64)
    out = {"log_n": log_n, "kernel_ms": {k: v["ms"] for k, v in st.items() if k.startswith("ntt_pass")}, "passes": {}}
    for pi, name in enumerate(("ntt_pass_a", "ntt_pass_b")):
        x = a[pi]
        ok = (x[..., 8] != 0) & (x[..., 13] != 0)              # waves that wrote (the last launch of the pass had that workgroup)
        seq = x[..., ORDER]
        d = np.diff(seq, axis=-1)                               # [wg, wave, tile, phase]
        rec = {"waves_sampled": int(ok.sum()), "phases": {}}
        for t in range(2):
            sel = ok[:, :, t]
            if not sel.any():
                continue
            dd = d[:, :, t, :][sel]
            tot = (seq[:, :, t, -1] - seq[:, :, t, 0])[sel]
            real = (x[:, :, t, 15] - x[:, :, t, 14])[sel]                 # the same tile by the constant 100 MHz counter
            rec["tile%d" % t] = {"total_cycles_mean": float(tot.mean()), "total_cycles_median": float(np.median(tot)),
                                 "tile_us_mean": float(real.mean()) / 100.0, "s_memtime_MHz": float(tot.mean() / (real.mean() / 100.0)),
                                 "phase_cycles_mean": [float(v) for v in dd.mean(axis=0)], "phase_cycles_median": [float(v) for v in np.median(dd, axis=0)],
                                 "phase_cycles_p10": [float(v) for v in np.percentile(dd, 10, axis=0)], "phase_cycles_p90": [float(v) for v in np.percentile(dd, 90, axis=0)]}
        # spread between the waves of ONE workgroup at the barriers: how long the first wave to arrive waits for the last
        both = ok[:, :, 0].all(axis=1)
        if both.any():
            arr = x[both][:, :, 0, :]
            rec["wave_spread_at"] = {"barrier 1": float((arr[:, :, 9].max(axis=1) - arr[:, :, 9].min(axis=1)).mean()),
                                     "barrier 2": float((arr[:, :, 11].max(axis=1) - arr[:, :, 11].min(axis=1)).mean()),
                                     "end of tile": float((arr[:, :, 13].max(axis=1) - arr[:, :, 13].min(axis=1)).mean())}
        out["passes"][name] = rec
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out)[:600])
    ctx.close()


def isa():
    """static instruction counts between consecutive s_memtime of the fixed 1024 x 4 instances in the stamps build's assembly"""
    s_path = "/tmp/r6_stamps_ntt.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-DNTT_LAB_STAMPS=1",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "distaff_amd", "csrc", "kernels_ntt.hip"), "-o", s_path])
    out = {}
    cur, segs = None, None
    for line in open(s_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()
            cur = name if re.search(r"ntt_pass_[ab]<1024, 8, (false|true), 10, 2, 0>", name) else None
            if cur:
                segs = out.setdefault(cur, [dict(valu=0, mad64=0, lds=0, vmem=0, salu=0, waitcnt=0, barrier=0, branch=0)])
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^\t([a-z_0-9]+)", line)
        if not m:
            continue
        op = m.group(1)
        if op == "s_memtime":
            segs.append(dict(valu=0, mad64=0, lds=0, vmem=0, salu=0, waitcnt=0, barrier=0, branch=0))
            continue
        g = segs[-1]
        if op.startswith("v_"):
            g["valu"] += 1
            g["mad64"] += op == "v_mad_u64_u32"
        elif op.startswith("ds_"):
            g["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            g["vmem"] += 1
        elif op == "s_waitcnt":
            g["waitcnt"] += 1
        elif op == "s_barrier":
            g["barrier"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            g["branch"] += 1
        elif op.startswith("s_"):
            g["salu"] += 1
    json.dump(out, sys.stdout, indent=1)


def table(path, isa_path=None):
    d = json.load(open(path))
    isa_d = json.load(open(isa_path)) if isa_path else {}
    print("n = 2^%d; kernel time of the instrumented commit: %s" % (d["log_n"], {k: round(v, 2) for k, v in d["kernel_ms"].items()}))
    for name, rec in d["passes"].items():
        print("\n### %s -- %d waves sampled; spread between the first and the last wave of a workgroup (cycles): %s" % (name, rec["waves_sampled"], {k: round(v) for k, v in rec.get("wave_spread_at", {}).items()}))
        for t in ("tile0", "tile1"):
            if t not in rec:
                continue
            r = rec[t]
            tot = r["total_cycles_mean"]
            print("\n%s: %.0f s_memtime ticks per tile and wave (median %.0f) = %.2f us by the 100 MHz counter: a tick is 1 / %.0f MHz\n" % (t, tot, r["total_cycles_median"], r.get("tile_us_mean", 0), r.get("s_memtime_MHz", 0)))
            print("| phase | mean cycles | share | median | p10 | p90 |")
            print("|---|---|---|---|---|---|")
            for i, ph in enumerate(PHASES):
                print("| %s | %.0f | %.1f %% | %.0f | %.0f | %.0f |" % (ph, r["phase_cycles_mean"][i], 100 * r["phase_cycles_mean"][i] / tot, r["phase_cycles_median"][i], r["phase_cycles_p10"][i], r["phase_cycles_p90"][i]))
    if isa_d:
        print("\nstatic instructions between consecutive stamps (same build, hipcc -S):")
        for k, segs in isa_d.items():
            print(k)
            for i, g in enumerate(segs):
                print("  segment %2d: %s" % (i, g))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), sys.argv[3])
    elif sys.argv[1] == "isa":
        isa()
    elif sys.argv[1] == "table":
        table(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
