#!/usr/bin/env python3
"""Benchmark of the MI355X STARK prover path (BASELINE.json: prover wall-time + trace-cells/sec on a 2^20-step trace).

A "step" is one complete `stark::prove` (LDE + trace Merkle + constraint evaluation + combination + constraint LDE/Merkle +
DEEP composition + FRI + proof-of-work + openings, default ProofOptions) over one Fibonacci execution trace that is already
resident in HBM (`value`); a second timed region starts from the trace in pinned host memory, the upload overlapped with the
extension (`prover_ms_incl_upload`, the PCIe-inclusive figure).  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import numpy as np
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
W_FIB = 20                     # registers of the Fibonacci trace


def phase_algorithmic_bytes(n, W, B):
    """minimum unavoidable HBM bytes of each prover phase, counted ONCE per phase (E = 16-byte elements)."""
    E, N = 16, n * B
    return {
        "lde": W * n * E + W * N * E,
        "trace_merkle": W * N * E + N * 32 + N * 32,
        "constraint_eval": (N // 4) * W * E + 3 * 8 * n * E,
        "combine": 3 * 8 * n * E + 8 * n * E,
        "constraint_lde_merkle": 8 * n * E + N * E + (N // 2) * 32,
        "deep_composition": W * n * E + 8 * n * E + N * E,
        "fri": int(4 / 3 * (N * E + N * 20)),
    }


HOST_ONLY_SOURCES = ("api.hip", "shard.hip", "comm.hip", "comm.h", "host_util.h", "host_proof.h", "host_vm.h")        # no device code: an edit there does not change what a counter summary measured


def csrc_digest():
    """digest of the KERNEL sources (same rule as __graft_entry__._sources_digest): ties a rocprofv3 summary to the code it measured.  Every
    file of distaff_amd/csrc that can contribute device code; the host-side translation units and headers are left out."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "distaff_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "distaff_amd", "csrc", "*.hip"))):
        if os.path.basename(f) in HOST_ONLY_SOURCES:
            continue
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(log_n, blowup=32, queries=50):
    """Times the CPU oracle (single-threaded restatement of the reference algorithm) on a bounded sample of the workload."""
    import oracle as O
    t = O.fibonacci_trace(1 << log_n)
    p = O.Prover.from_trace(t, 1, ext=blowup, num_queries=queries)
    t0 = time.time()
    p.prove()
    dt = time.time() - t0
    return {"value": (1 << log_n) * t.width / dt, "unit": "trace-cells/s", "cores": 1, "kind": "port",
            "sample": "one full prove() of a 2^%d-step Fibonacci trace (same program and ProofOptions), %.1f s, oracle/liboracle.so -O3, 1 of %d host cores"
                      % (log_n, dt, os.cpu_count() or 1),
            "prove_ms": dt * 1e3, "phase_ms": [round(x, 1) for x in p.phase_ms], "reference_published": REFERENCE_PUBLISHED,
            "same_size_reference": same_size_cpu_leg(),
            "same_size_note": "a 2^%d sample beside a 2^%d headline: the CPU's cost per cell grows with n (log n transforms, caches), so this rate flatters the CPU at the headline size; "
                              "same_size_reference quotes the committed leg at the headline size (one full prove() at 2^20 takes minutes: run once per round with --cpu-log-n 20, not in the driver's bench)" % (log_n, int(os.environ.get("BENCH_LOG_N", "20")))}


def same_size_cpu_leg():
    """The CPU leg at the HEADLINE size, from the newest committed builder run (profiles/r<N>_bench_cpu_2_20.json: `bench.py --cpu-log-n 20`, one
    full oracle prove() of the 2^20-step trace on one host core of a GPU box of this pool) -- quoted, not re-measured: it takes four minutes."""
    import glob
    import re
    pat = re.compile(r"r(\d+)_bench_cpu_2_20\.json")
    files = sorted((int(pat.fullmatch(os.path.basename(f)).group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_cpu_2_20.json")) if pat.fullmatch(os.path.basename(f)))
    if not files:
        return None
    try:
        line = [l for l in open(files[-1][1]).read().splitlines() if l.startswith("{")][-1]
        cb = json.loads(line)["cpu_baseline"]
        return {"source": "profiles/" + os.path.basename(files[-1][1]), "log_n": 20, "value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                "prove_s": round(cb["prove_ms"] / 1e3, 1), "phase_ms": cb.get("phase_ms"), "sample": cb.get("sample")}
    except Exception as e:                                           # noqa: BLE001  (never lose the bench line over a committed file)
        return {"source": "profiles/" + os.path.basename(files[-1][1]), "error": str(e)}


PMC_WORKLOAD = {"tag": ""}        # which committed counter summary belongs to the workload of this run: "" = default (config 3), "config4_", "config5_", "config2_"


def pmc_workload_tag(workload, log_n, blowup, queries, world):
    """The rocprofv3 counter passes are taken per configuration (tools/profile_round.sh: config 3; tools/profile_config.sh: configs 2, 4, 5):
    '' / 'config2_' / 'config4_' / 'config5_' for the workloads they exist for, None otherwise."""
    if world != 1:
        return None
    if workload == "commit":
        return "config2_" if (log_n, blowup) == (16, 32) else None
    return {(20, 32, 50): "", (22, 32, 50): "config4_", (24, 16, 100): "config5_"}.get((log_n, blowup, queries))


def _pmc_summary():
    """(rows, source) of the newest committed PMC summary of this run's workload (profiles/r<N>_[config<k>_]pmc_per_kernel.csv, newest = largest
    round NUMBER) -- or (None, reason) when there is none, when it carries no stamp, or when its stamp (digest of distaff_amd/csrc at the time
    of the rocprofv3 passes) differs from the sources of this run."""
    import csv
    import glob
    import re
    if PMC_WORKLOAD["tag"] is None:
        return None, "no counter passes exist for this workload"
    pat = re.compile(r"r(\d+)_%spmc_per_kernel\.csv" % re.escape(PMC_WORKLOAD["tag"]))
    files = sorted((int(pat.fullmatch(os.path.basename(f)).group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_per_kernel.csv"))
                   if pat.fullmatch(os.path.basename(f)))
    if not files:
        return None, "no PMC summary of this workload under profiles/"
    newest = files[-1][1]
    meta_path = newest.replace("_pmc_per_kernel.csv", "_meta.json")
    if not os.path.exists(meta_path):
        return None, os.path.basename(newest) + " carries no stamp"
    meta = json.load(open(meta_path))
    if meta.get("csrc_sha16") != csrc_digest():
        return None, "%s was taken on kernel sources %s, this run is %s: refused" % (os.path.basename(newest), meta.get("csrc_sha16"), csrc_digest())
    with open(newest, newline="") as fh:
        return list(csv.DictReader(fh)), "%s (git %s)" % (os.path.basename(newest), str(meta.get("git_head"))[:10])


def _pmc_norm(name):
    import re
    name = name.replace("void ", "").replace(" ", "").replace("false", "0").replace("true", "1")
    return re.sub(r"(\d+)u\b", r"\1", re.sub(r"\(unsignedint\)(\d+)", r"\1", name))       # unsigned template arguments: 240u / (unsigned int)240


def pmc_rows(kernel):
    """EVERY row of the summary whose kernel name starts with `kernel` (names compared without blanks, template booleans as 0 / 1: the
    library's event names print them so).  The library names a launch by its function, not by its template instance, so one event name can
    cover several instances (lincomb_kernel<2> and <4>, the pass kernels of different tile shapes): a per-launch figure of the NAME is the
    call-weighted mean over its instances -- taking the first matching row priced every launch as the most frequent instance and produced
    issue fractions above 1."""
    rows, source = _pmc_summary()
    if rows is None:
        return [], source
    hit = [r for r in rows if _pmc_norm(r["kernel"]).startswith(_pmc_norm(kernel))]
    # "ntt_pass_a" must not swallow a longer function name that merely starts with it: the next character is '<', '(' or the end
    hit = [r for r in hit if _pmc_norm(r["kernel"])[len(_pmc_norm(kernel)):][:1] in ("", "<", "(")] or hit
    return hit, (source if hit else "kernel not in " + source.split(" ")[0])


def pmc_row(kernel):
    """One row for `kernel`: the call-weighted mean of the numeric columns over its template instances (see pmc_rows)."""
    rows, source = pmc_rows(kernel)
    if not rows:
        return None, source
    if len(rows) == 1:
        return rows[0], source
    calls = [float(r.get("calls") or 1) for r in rows]
    out = {"kernel": kernel, "calls": str(int(sum(calls))), "instances": len(rows)}
    for key in rows[0]:
        if key in ("kernel", "calls"):
            continue
        try:
            out[key] = repr(sum(c * float(r[key]) for c, r in zip(calls, rows)) / sum(calls))
        except (TypeError, ValueError):
            pass
    for key in ("fetch_bytes_per_launch_raw", "fetch_bytes_per_launch_x2", "write_bytes_per_launch_raw"):
        if key in out:
            out[key] = str(int(float(out[key])))
    return out, source + " (%d template instances, call-weighted)" % len(rows)


# wave-instructions per second: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz.  The 4 cycles are MEASURED for the
# integer instructions this path is made of (profiles/r6_issue_slots.txt: v_add_co / v_addc_co with SGPR carries, v_cndmask on an SGPR mask,
# v_xor / v_alignbit each reach 0.95 - 0.99 of one wave-instruction per 4 cycles and SIMD; v_mad_u64_u32 takes 1.41 such slots); the guide's
# 2-cycle figure is for v_fma_f32 and does not apply to them.
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4
NOMINAL_SCLK_MHZ = 2400.0
# bytes a memory-side read request carries per 64 B that FETCH_SIZE tallies, per kernel (profiles/r6_fetch_factor.txt): wide streaming reads travel
# as 128-byte requests tallied at 64 (x 2, the guide's figure); the first transform pass reads 64-byte row segments only, tallied in full (x 1)
FETCH_FACTOR = {"ntt_pass_a": 1.0, "ntt_pass_mid": 1.0}


def counter_traffic(name, row):
    """HBM-side bytes per launch from the committed counter row of kernel `name`: FETCH_SIZE x its calibrated factor + WRITE_SIZE"""
    if row is None or not row.get("fetch_bytes_per_launch_raw") or not row.get("write_bytes_per_launch_raw"):
        return None, None
    factor = FETCH_FACTOR.get(name.split("<")[0], 2.0)
    return int(float(row["fetch_bytes_per_launch_raw"]) * factor) + int(float(row["write_bytes_per_launch_raw"])), factor


def valu_issue(stats, proof_ms, steps):
    """Share of the VALU issue slots the kernels of one proof fill: SQ_INSTS_VALU per launch (committed, stamped PMC summary) x the
    launches of this run, against 1024 SIMDs x 2.4 GHz / 4 cycles.  `stats`: {kernel: {"launches", "ms"}} over `steps` proofs."""
    per_kernel, total, missing, refused = {}, 0.0, [], []
    clock_ms, clock_weighted, per_kernel_clock = 0.0, 0.0, {}
    for name, st in stats.items():
        row, _ = pmc_row(name)
        if row is None or not row.get("SQ_INSTS_VALU"):
            if st["ms"] / steps >= 0.05:
                missing.append(name)
            continue
        insts = float(row["SQ_INSTS_VALU"]) * st["launches"]
        total += insts
        if row.get("sclk_MHz"):                      # the clock the kernel ran at during the counter pass (GRBM_GUI_ACTIVE / duration: tools/summarize_profile.py)
            clock_ms += st["ms"]; clock_weighted += st["ms"] * float(row["sclk_MHz"])
            if st["ms"] / steps >= 0.5:
                per_kernel_clock[name] = round(float(row["sclk_MHz"]))
        if st["ms"] / steps >= 0.5:
            frac = insts / (st["ms"] * 1e-3) / VALU_ISSUE_PEAK
            # a fraction above 1 is an accounting error (instruction counts of another launch mix than this run's), never a measurement
            per_kernel[name] = round(frac, 4) if frac <= 1.02 else None
            if frac > 1.02:
                refused.append("%s: %.3f" % (name, frac))
    if total == 0:
        return None
    return {"unit": "wave64 VALU instructions/s", "peak": VALU_ISSUE_PEAK, "proof_achieved": total / steps / (proof_ms * 1e-3),
            "proof_frac": round(total / steps / (proof_ms * 1e-3) / VALU_ISSUE_PEAK, 4), "kernel_frac": per_kernel,
            "kernel_sclk_MHz": per_kernel_clock or None, "sclk_MHz_time_weighted": round(clock_weighted / clock_ms, 1) if clock_ms > 0 else None,
            "not_counted": missing, "refused_above_1": refused, "note": "instruction counts per launch from the stamped rocprofv3 summary, launch times of this run; "
            "the SQ_INSTS_VALU of a kernel name is the call-weighted average over its template instances and launches of a proof (pmc_rows), so kernels whose launches differ in size are exact for the whole proof only"}



def box_fingerprint(cal, mad_peak, mulmod_peak):
    """What tells one box of the pool from another: the two arithmetic calibrations, the straight-line-code probe (dst_bench_code: time per
    instruction of 176 KiB of code against 16 KiB -- 1.0 on a healthy device), device name / CU count and rocm-smi's clocks and partition modes."""
    import subprocess
    import torch
    box = {"mad_peak": mad_peak, "mulmod_peak": mulmod_peak}
    try:
        # the shader clock this box sustains under the path's arithmetic (four fe_mul chains per lane on every SIMD; s_memtime against s_memrealtime)
        box["sclk_under_load_MHz"] = round(cal.bench_clock(1 << 20, 512), 1) if cal is not None else None
        box["sclk_nominal_MHz"] = NOMINAL_SCLK_MHZ
    except Exception as e:                                           # noqa: BLE001
        box["sclk_under_load_MHz"] = None
        box["sclk_error"] = str(e)
    try:
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        box["device"] = props.name
        box["compute_units"] = props.multi_processor_count
        box["hbm_GiB"] = round(props.total_memory / 2**30, 1)
    except Exception as e:                                           # noqa: BLE001
        box["device_error"] = str(e)
    try:
        if cal is None:
            raise RuntimeError("no calibration build")
        t16, t176, t176c = cal.bench_code(16), cal.bench_code(176), cal.bench_code(177)
        box["code_probe"] = {"ms_16KiB": round(t16, 4), "ms_176KiB": round(t176, 4), "per_instruction_ratio": round((t176 / 176.0) / (t16 / 16.0), 3),
                             "ms_176KiB_convoy": round(t176c, 4), "convoy_per_instruction_ratio": round((t176c / 176.0) / (t16 / 16.0), 3),
                             "note": "2^23 lanes, 128 per workgroup, two waves per SIMD, every wavefront runs the code once (kernels_probe.hip); "
                                     "convoy = 256 lanes per workgroup and a workgroup barrier every 16 KiB of code"}
    except Exception as e:                                           # noqa: BLE001
        box["code_probe"] = {"error": str(e)}
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showcomputepartition", "--showmemorypartition", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
        smi = json.loads(r.stdout.decode() or "{}")
        card = smi.get("card%d" % torch.cuda.current_device()) or (list(smi.values())[0] if smi else {})
        box["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "socclk", "partition"))}
    except Exception as e:                                           # noqa: BLE001
        box["rocm_smi"] = {"error": str(e)}
    return box


def kernel_rooflines(stats, steps, default_workload, top=5):
    """Per kernel of the timed region (the heavy kernels are bracketed by HIP events on the launch stream), largest total time first: the
    HBM fraction of its algorithmic bytes, the counter traffic and the VALU-issue fraction (stamped PMC summary of the same command)."""
    rows = []
    for name, st in sorted(stats.items(), key=lambda kv: -kv[1]["ms"])[:top]:
        if st["launches"] == 0 or st["ms"] <= 0:
            continue
        per_launch_ms = st["ms"] / st["launches"]
        per_launch_bytes = st["bytes"] / st["launches"]
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
        row, source = pmc_row(name) if default_workload else (None, "no counter passes exist for this workload")
        traffic, fetch_factor = counter_traffic(name, row)
        valu = None
        if row is not None and row.get("SQ_INSTS_VALU"):
            valu = round(float(row["SQ_INSTS_VALU"]) / (per_launch_ms * 1e-3) / VALU_ISSUE_PEAK, 4)
            valu = valu if valu <= 1.02 else None
        sclk = float(row["sclk_MHz"]) if (row is not None and row.get("sclk_MHz")) else None
        rows.append({"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_over_algorithmic": None if traffic is None else round(traffic / per_launch_bytes, 2), "traffic_source": source,
                     "traffic_fetch_factor": fetch_factor,
                     "valu_issue_frac": valu, "sclk_MHz": None if sclk is None else round(sclk),
                     "valu_issue_frac_at_measured_clock": None if (valu is None or not sclk) else round(valu * NOMINAL_SCLK_MHZ / sclk, 4), "ms_per_step": round(st["ms"] / steps, 3), "launches_per_step": st["launches"] / steps,
                     "avg_launch_ms": round(per_launch_ms, 4), "algorithmic_bytes_per_launch": per_launch_bytes})
    return rows


REFERENCE_PUBLISHED = {
    "source": "/root/reference/README.md:147-161 ('very informal benchmarks'): Fibonacci program, default ProofOptions, execute + prove, Intel Core i5-7300U @ 2.60 GHz, single thread",
    "execute_plus_prove_seconds_by_log2_operations": {"8": 0.19, "10": 0.35, "12": 1.0, "14": 4.5, "16": 18.0, "18": 78.0, "20": 1080.0},
    "note_2^20": "18 min on a machine that ran out of RAM at 5.6 GB; the author's estimate with ~20 GB is ~5 min (README.md:161)",
    "trace_cells_per_sec_2^16": (1 << 16) * W_FIB / 18.0, "trace_cells_per_sec_2^20_as_published": (1 << 20) * W_FIB / 1080.0,
    "trace_cells_per_sec_2^20_author_estimate": (1 << 20) * W_FIB / 300.0,
    "hardware_differs": "another CPU than this box's: quoted beside cpu_baseline as the only figures the reference publishes, not as vs_baseline",
}


def fibonacci_trace_cached(D, log_n):
    """the benchmark input (host-side VM: 14 s at 2^20, minutes at 2^24 -- one thread, the sponge is a chain).  BENCH_TRACE_CACHE=<dir>
    keeps generated traces as .npy files so that several invocations on one box (bench lines + rocprofv3 passes) generate them once."""
    d = os.environ.get("BENCH_TRACE_CACHE")
    if not d:
        return D.fibonacci_trace(log_n)
    f = os.path.join(d, "fibonacci_%d.npz" % log_n)
    if os.path.exists(f):
        z = np.load(f)
        return z["cols"], z["program_hash"].tobytes(), int.from_bytes(z["result"].tobytes(), "little")
    cols, ph, res = D.fibonacci_trace(log_n)
    if int(os.environ.get("RANK", "0")) == 0:
        tmp = f + ".tmp.%d.npz" % os.getpid()
        try:
            os.makedirs(d, exist_ok=True)
            np.savez(tmp, cols=cols, program_hash=np.frombuffer(ph, dtype=np.uint8), result=np.frombuffer(res.to_bytes(16, "little"), dtype=np.uint8))
            os.replace(tmp, f)
        except OSError:                                              # no room: the next invocation generates the trace again
            if os.path.exists(tmp):
                os.remove(tmp)
    return cols, ph, res


def splitmix_columns(log_n, W):
    """BASELINE config 2's input: W columns of 2^log_n elements uniform in [0, p): two 64-bit draws per element from
    splitmix64(seed = 0x44697374616666 + column), low word first, values >= p rejected.  uint64 [W, n, 2]."""
    n = 1 << log_n
    P = 2**128 - 45 * 2**40 + 1
    p_lo, p_hi = np.uint64(P & (2**64 - 1)), np.uint64(P >> 64)
    cols = np.zeros((W, n, 2), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for c in range(W):
            m = 2 * n + 64                                             # rejections are ~2^-82 per element: never in practice, handled anyway
            x = np.uint64(0x44697374616666 + c) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, m + 1, dtype=np.uint64)
            z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            lo, hi = z[0::2], z[1::2]
            ok = (hi < p_hi) | ((hi == p_hi) & (lo < p_lo))
            cols[c, :, 0], cols[c, :, 1] = lo[ok][:n], hi[ok][:n]
    return cols


def cpu_baseline_commit(cols, blowup):
    """config 2's CPU leg: the oracle's TraceTable::extend + build_merkle_tree (prover.rs:22-35) on the same columns"""
    import oracle as O
    p = O.Prover(cols, 1, 0, [], [], ext=blowup)
    t0 = time.time(); p.step(1); t1 = time.time(); p.step(2); t2 = time.time()
    W, n = cols.shape[0], cols.shape[1]
    return {"value": n * W / (t2 - t0), "unit": "trace-cells/s", "cores": 1, "kind": "port",
            "sample": "extend + trace Merkle tree of the SAME %d x 2^%d columns, %.1f s, oracle/liboracle.so -O3, 1 of %d host cores" % (W, n.bit_length() - 1, t2 - t0, os.cpu_count() or 1),
            "prove_ms": (t2 - t0) * 1e3, "phase_ms": [round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1)], "root_hex": p.get_bytes("roots")[:32].hex(),
            "reference_published": REFERENCE_PUBLISHED}


_STATE = {"rank": 0, "stage": "start", "args": None, "stdout_fd": None}


def claim_stdout():
    """stdout carries the ONE JSON line and nothing else: libraries print there too (RCCL writes its version banner to stdout through C
    stdio, flushed when the process exits -- i.e. AFTER the line), so file descriptor 1 is pointed at stderr for the rest of the run and the
    line is written to the saved descriptor."""
    if _STATE["stdout_fd"] is None:
        sys.stdout.flush()
        _STATE["stdout_fd"] = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush(); sys.stderr.flush()
    if _STATE["stdout_fd"] is None:
        print(line, flush=True)
    else:
        os.write(_STATE["stdout_fd"], (line + "\n").encode())


def error_line(message):
    """the contract's ONE JSON line also when the run fails: rank 0 says what failed and where instead of leaving the driver with a hang
    or a bare traceback"""
    a = _STATE["args"]
    if _STATE["rank"] == 0:
        emit(json.dumps({"metric": "trace_cells_per_sec", "value": None, "unit": "trace-cells/s", "n_gpus": getattr(a, "gpus", None), "steps": getattr(a, "steps", None),
                         "warmup": getattr(a, "warmup", None), "error": str(message)[-2000:], "stage": _STATE["stage"]}))


def start_watchdog(seconds):
    """a collective that never returns (a rank died, RCCL cannot reach a peer) must end the run with an error line, not hang the driver.
    It measures STALL time: every stage change and every finished step (progress()) re-arms it, so a long but progressing run (config 5's
    trace generation, the CPU leg, the profiled steps) is not cut off."""
    import threading
    _STATE["progress_at"] = time.monotonic()
    stop = threading.Event()

    def watch():
        while not stop.wait(1.0):
            idle = time.monotonic() - _STATE["progress_at"]
            # one blocking host call each: the VM's trace generation (minutes at 2^24) and the CPU leg get a longer leash
            if idle > seconds * (6 if _STATE["stage"] in ("trace", "cpu baseline") else 1):
                error_line("no progress for %d s in stage '%s' (BENCH_TIMEOUT_S): giving up instead of hanging" % (idle, _STATE["stage"]))
                sys.stdout.flush(); sys.stderr.flush()
                os._exit(3)
    t = threading.Thread(target=watch, daemon=True)
    t.start()
    t.cancel = stop.set
    return t


def progress(stage=None):
    if stage is not None:
        _STATE["stage"] = stage
    _STATE["progress_at"] = time.monotonic()


def library_identity(D, allow_override):
    """Which shared library this process measures.  Anything but the in-tree product build (distaff_amd/libdistaff_hip.so) is refused unless
    --allow-lib-override is given -- DISTAFF_HIP_LIB can point the binding at any other build, the CPU emulation of the tests included --
    and every DISTAFF_* / BENCH_* variable that is set goes into the line."""
    product = os.path.join(os.path.dirname(os.path.abspath(__file__)), "distaff_amd", "libdistaff_hip.so")
    loaded = os.path.realpath(D.library_path())
    ident = {"path": os.path.relpath(loaded, os.path.dirname(os.path.abspath(__file__))), "is_product_build": loaded == os.path.realpath(product),
             "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("DISTAFF_", "BENCH_"))}}
    if not ident["is_product_build"] and not allow_override:
        raise SystemExit("the loaded library is %s, not distaff_amd/libdistaff_hip.so (DISTAFF_HIP_LIB=%r): refusing to measure it; pass --allow-lib-override to do so anyway"
                         % (loaded, os.environ.get("DISTAFF_HIP_LIB")))
    return ident


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)         # the driver's own values: a proof is 37 ms, and the first steps after a cold
    ap.add_argument("--warmup", type=int, default=5)         # start run up to 8 % slower (clock ramp; step_ms shows the spread)
    ap.add_argument("--workload", choices=["prove", "commit"], default="prove",
                    help="prove: full stark::prove of the Fibonacci trace (BASELINE configs 3-5, the metric); commit: LDE + trace Merkle tree only on random columns (BASELINE config 2)")
    ap.add_argument("--log-n", type=int, default=None, help="log2 of the trace length (default 20; 16 for --workload commit)")
    ap.add_argument("--cpu-log-n", type=int, default=int(os.environ.get("BENCH_CPU_LOG_N", "16")))
    ap.add_argument("--log-blowup", type=int, default=5, help="log2 of the extension factor (default ProofOptions: 5; BASELINE config 5: 4)")
    ap.add_argument("--queries", type=int, default=50, help="number of queries (default ProofOptions: 50; BASELINE config 5: 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the one-time check of the timed proof by the oracle's verifier (outside the timed regions)")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip the second timed region (trace starting in pinned host memory)")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded (multi-GPU) code path even with one rank")
    ap.add_argument("--allow-lib-override", action="store_true", help="measure whatever library DISTAFF_HIP_LIB names instead of refusing anything but distaff_amd/libdistaff_hip.so")
    args = ap.parse_args()
    if args.log_n is None:
        args.log_n = int(os.environ.get("BENCH_LOG_N", "16" if args.workload == "commit" else "20"))
    _STATE["args"] = args
    _STATE["rank"] = int(os.environ.get("RANK", "0"))
    claim_stdout()
    # stall limit: with several ranks a stage that makes no progress for minutes is a collective that will never return
    watchdog = start_watchdog(int(os.environ.get("BENCH_TIMEOUT_S", "1500" if args.gpus == 1 else "420")))
    try:
        run(args)
    except SystemExit as e:
        if e.code not in (None, 0):
            error_line(e.code)
        raise
    except BaseException as e:                                       # noqa: BLE001 -- the line first, then the traceback
        import traceback
        traceback.print_exc()
        error_line("%s: %s" % (type(e).__name__, e))
        sys.stdout.flush()
        os._exit(1)                                                  # peers may sit in a collective: do not wait for their clean-up
    finally:
        watchdog.cancel()


def run(args):
    import torch
    os.environ.pop("DISTAFF_TEST_HOOKS", None)                       # bench.py measures the product library, whatever its caller's tests bound
    import distaff_amd as D

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    lib_ident = library_identity(D, args.allow_lib_override)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP prover has no CPU fallback")
    if world > 8 or (world & (world - 1)):
        raise SystemExit("--gpus must be 1, 2, 4 or 8 (cosets of the LDE domain are split evenly)")
    if args.workload == "commit" and (world > 1 or args.force_sharded):
        raise SystemExit("--workload commit is the single-GPU config 2 (LDE + Merkle only)")
    # More ranks than visible devices: the ranks SHARE devices (a one-GPU box exercising the N-process path).  RCCL refuses two ranks on
    # one device, so the library's collectives then travel through the callback transport over a gloo group (host-staged).  Functional
    # run: the JSON line says so and is no scaling measurement.
    ndev = max(1, torch.cuda.device_count())
    shared_devices = world > ndev
    device = local_rank % ndev
    torch.cuda.set_device(device)
    want = os.environ.get("DISTAFF_SHARD_TRANSPORT", "rccl")         # rccl | callbacks
    if want not in ("rccl", "callbacks"):
        raise SystemExit("DISTAFF_SHARD_TRANSPORT must be rccl or callbacks")
    host_group = shared_devices or os.environ.get("DISTAFF_SHARD_BACKEND") == "gloo"
    if shared_devices:
        want = "callbacks"
    dist = None
    progress("process group")
    if world > 1 or args.force_sharded:
        import datetime
        import torch.distributed as dist
        kw = {"timeout": datetime.timedelta(seconds=int(os.environ.get("BENCH_PG_TIMEOUT_S", "600")))}
        if not host_group:
            kw["device_id"] = torch.device("cuda", device)
        if "MASTER_ADDR" not in os.environ:
            kw.update(init_method="tcp://127.0.0.1:29517", rank=0, world_size=1)
        dist.init_process_group("gloo" if host_group else "nccl", **kw)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    log_n = args.log_n
    n = 1 << log_n
    progress("trace")
    if args.workload == "commit":
        cols, program_hash, result = splitmix_columns(log_n, W_FIB), None, None
    else:
        cols, program_hash, result = fibonacci_trace_cached(D, log_n)   # host: VM trace of `begin repeat.K swap dup.2 drop add end end`
    blowup = 1 << args.log_blowup
    progress("context")
    ctx = D.Context(log_n, W_FIB, 1, 0, device=device, rank=rank, world=world, log_blowup=args.log_blowup, num_queries=args.queries)   # defaults = default ProofOptions: blowup 32, 50 queries, grinding 20
    c_orchestration = (world > 1 or args.force_sharded) and os.environ.get("DISTAFF_SHARD_ORCH", "c") != "python"
    if c_orchestration and world > 1:
        ctx.upload_owned(cols)                                      # dst_prove_sharded splits the interpolation by columns: 1 / world of the trace per GPU
    else:
        ctx.upload(cols)                                            # inputs resident in HBM before the timed region

    progress("communicator")
    stage_of = None
    transport_note = None
    comm_info = None
    if args.workload == "commit":
        def prove():
            return ctx.commit_trace()
        transport = "none"
    elif world == 1 and not args.force_sharded:
        def prove():
            return ctx.prove([1, 0], [result])
        transport = "none"
    else:
        # ONE proof sharded over the GPUs by cosets of the LDE domain.  Default: the whole exchange sequence behind the C-ABI
        # (dst_prove_sharded: RCCL all-to-all / all-gather issued by the library; torch.distributed only carries the 128-byte unique id
        # and the timing barrier).  DISTAFF_SHARD_TRANSPORT=callbacks: the same library code with its collectives handed to the host's
        # channel (dst_comm_init_callbacks over the torch.distributed group) -- also the fallback when the library's RCCL binding cannot
        # be brought up on some rank.  DISTAFF_SHARD_ORCH=python keeps the host-orchestrated sequence of distaff_amd/sharded.py
        # (torch.distributed collectives around dst_shard_*) as a cross-check.
        if os.environ.get("DISTAFF_SHARD_ORCH", "c") != "python":
            comm = None
            if want == "rccl":
                failure = None
                try:
                    if os.environ.get("BENCH_SIMULATE_RCCL_FAILURE") == str(rank):      # tests: the fallback below, as if this rank's communicator could not be created
                        raise RuntimeError("simulated failure of dst_comm_init (BENCH_SIMULATE_RCCL_FAILURE)")
                    ids = [D.Comm.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    comm = D.Comm.rccl(ids[0], rank, world, device)
                except Exception as e:                               # noqa: BLE001 -- agreed on below
                    failure = "%s: %s" % (type(e).__name__, e)
                failures = [None] * world
                dist.all_gather_object(failures, failure)
                if any(failures):
                    if comm is not None:
                        comm.close()
                    comm = None
                    transport_note = "the library's RCCL communicator could not be created (%s): collectives carried by the torch.distributed group instead" % next(f for f in failures if f)
                    if rank == 0:
                        print("[bench] " + transport_note, file=sys.stderr, flush=True)
                else:
                    transport = "dst_prove_sharded over RCCL"
            if comm is None:
                comm = D.Comm.over_torch(dist)
                transport = "dst_prove_sharded over the callback transport (torch.distributed %s%s)" % (
                    dist.get_backend(), ", host-staged, ranks sharing %d device(s)" % ndev if shared_devices else "")

            # every host wait behind a collective is bounded by the library (DST_ERR_COMM on expiry); the first proof of a fresh RCCL communicator
            # also pays its lazy connection set-up, so the bench grants more than the library's 60 s -- and less than its own watchdog
            comm.set_timeout(float(os.environ.get("BENCH_COMM_TIMEOUT_S", "180")))
            # what the transport says about itself, from every rank: for RCCL the number of ranks the live communicator connected
            # (ncclCommCount), each rank's index in it and the device it runs on -- the line's proof that RCCL saw `world` ranks
            infos = [None] * world
            dist.all_gather_object(infos, dict(comm.describe(), local_rank=local_rank, visible_device=device))
            comm_info = {"transport": infos[0]["transport"], "ranks_per_rank": [i["rccl_ranks"] for i in infos] if infos[0]["transport"] == "rccl" else None,
                         "rccl_rank_per_rank": [i["rccl_rank"] for i in infos] if infos[0]["transport"] == "rccl" else None,
                         "rccl_version": infos[0]["rccl_version"] or None,
                         "device_per_rank": [i["device"] if i["device"] >= 0 else i["visible_device"] for i in infos],
                         "all_ranks_same_transport": len({i["transport"] for i in infos}) == 1}
            if comm_info["transport"] == "rccl" and any(k != world for k in comm_info["ranks_per_rank"]):
                raise SystemExit("the RCCL communicators report %s ranks, the job has %d" % (comm_info["ranks_per_rank"], world))

            def stage_of():
                return ctx.shard_stage_ms()

            def prove():
                return ctx.prove_sharded(comm, [1, 0], [result])
        else:
            from distaff_amd import sharded
            # hand-off between the library's buffers and the collective's tensors: "device" = through the tensors' data pointers (device tensors
            # of an nccl group), "host" = staged through host arrays (the only form a gloo group of ranks sharing a GPU can take)
            device_path = (os.environ.get("DISTAFF_SHARD_HANDOFF") or ("host" if host_group else "device")) == "device"
            tcomm = sharded.TorchComm(dist, torch.device("cuda", device), device_path=device_path)
            if device_path:
                # self-check of the direct hand-off between the library's buffers and torch tensors; fall back to host staging
                ok = True
                try:
                    ctx.shard_commit_trace()
                    host = np.empty(ctx.shard_export_size(0), dtype=np.uint8)
                    ctx.shard_export(0, 0, host.ctypes.data, False)
                    big = torch.zeros(ctx.shard_export_size(0), dtype=torch.uint8, device=tcomm.device)
                    torch.cuda.synchronize()
                    ctx.shard_export(0, 0, big.data_ptr(), True)
                    ok = bool((big.cpu().numpy() == host).all())
                except Exception:                                       # noqa: BLE001
                    ok = False
                flags = tcomm.all_gather_object(ok)
                if not all(flags):
                    tcomm.device_path = False
            transport = "torch.distributed, " + ("device" if tcomm.device_path else "host-staged")
            prover = sharded.ShardedProver(ctx, tcomm)

            def stage_of():
                return prover.stage_ms

            def prove():
                return prover.prove([1, 0], [result])

    progress("warm-up")
    proof = None
    for _ in range(args.warmup):
        proof = prove()
        progress()
    # timed region: HIP events around the heavy kernels only (NTT passes, constraint kernel, leaf hashing: the dominant kernel is one
    # of them); bracketing all ~300 launches of a proof costs ~5 % and is done on one extra, untimed step for the kernel table
    progress("timed region")
    ctx.set_profiling(2)
    ctx.kernel_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    phase_sum = [0.0] * 9
    stage_sum = {}
    exchange_sum = {}
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        proof = prove()
        step_ms.append((time.perf_counter() - ts) * 1e3)               # prove() returns the finished proof: no extra synchronisation
        progress()
        for i, v in enumerate(ctx.phase_ms()):
            phase_sum[i] += v
        if stage_of is not None:
            for k, v in stage_of().items():
                stage_sum[k] = stage_sum.get(k, 0.0) + v
        if c_orchestration:
            for k, v in ctx.shard_exchange_ms().items():
                exchange_sum[k] = exchange_sum.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_group else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    progress("kernel table")
    stats = ctx.kernel_stats(reset=True)
    ctx.set_profiling(1)
    prove()                                                        # untimed: every launch bracketed, for the "kernels" table
    all_stats = ctx.kernel_stats(reset=True)
    ctx.set_profiling(0)

    # second timed region (single context only): the trace starts in page-locked HOST memory, as stark::prove receives it (prover.rs:17);
    # the upload is asynchronous DMA and the registers are extended group by group as they arrive
    incl_upload_ms = None
    if transport == "none" and not args.no_upload_leg:
        progress("upload leg")
        table, handle = ctx.pinned_trace(cols)
        for _ in range(max(1, args.warmup)):
            ctx.upload_async(table); proof_u = prove()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ctx.upload_async(table)
            proof_u = prove()
        barrier()
        incl_upload_ms = (time.perf_counter() - t1) / args.steps * 1e3
        ctx.release_pinned(handle)
        if proof_u != proof:
            raise SystemExit("the proof made from the asynchronously uploaded trace differs from the one made from the resident trace")

    if rank != 0:
        if dist is not None:
            dist.barrier()                                             # rank 0 may still verify / time the CPU leg: leave together
            dist.destroy_process_group()
        return
    progress("report")

    ms_per_step = elapsed / args.steps * 1e3
    cells = n * W_FIB
    value = cells / (elapsed / args.steps)                      # one proof per step regardless of the number of GPUs
    # kernels by device time, measured with HIP events on the launch stream inside the timed region.  `roofline` is the kernel with the
    # largest total time IN THIS RUN; `roofline_transform` is always the first pass of the transforms (the kernel profiles/README.md
    # describes), so that the two can be told apart when another kernel is dominant on some box; `rooflines` lists the top five.
    PMC_WORKLOAD["tag"] = pmc_workload_tag(args.workload, log_n, blowup, args.queries, world)      # which committed counter passes (if any) were taken on this workload
    default_workload = PMC_WORKLOAD["tag"] is not None
    rooflines = kernel_rooflines(stats, args.steps, default_workload)
    note = ("the path is 128-bit modular integer arithmetic on the VALU (valu_issue_frac, alu_roofline, DESIGN.md section 3), not HBM-bound; `frac` is the per-KERNEL figure: "
            "its algorithmic bytes include the staging array the two-pass transform writes in pass A and re-reads in pass B, which SURVEY section 8(d) does not count -- "
            "the phase-level fraction with section 8(d)'s bytes counted once is phase_hbm.lde (and phase_hbm.proof for the whole proof)")
    roofline = dict(rooflines[0], note=note) if rooflines else None
    transform = [r for r in kernel_rooflines(stats, args.steps, default_workload, top=len(stats)) if r["kernel"].startswith("ntt_pass_a")]
    roofline_transform = dict(transform[0], note=note) if transform else None
    dom = (roofline["kernel"], stats[roofline["kernel"]]) if roofline else (None, None)
    # phase-level HBM fractions: SURVEY section 8(d)'s bytes counted once per phase, against the phase's wall time
    phase_names = ["lde", "trace_merkle", "constraint_eval", "combine", "constraint_lde_merkle", "deep_composition", "fri", "pow_queries", "openings"]
    phase_hbm = None
    if transport == "none":
        pb = phase_algorithmic_bytes(n, W_FIB, blowup)
        if args.workload == "commit":
            pb = {k: pb[k] for k in ("lde", "trace_merkle")}
        phase_hbm = {}
        for k, v in zip(phase_names, phase_sum):
            if k in pb and v > 0:
                gbs = pb[k] / (v / args.steps * 1e-3) / 1e9
                phase_hbm[k] = {"algorithmic_GiB": round(pb[k] / 2**30, 3), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        total_b = sum(pb.values())
        gbs = total_b / (ms_per_step * 1e-3) / 1e9
        phase_hbm["proof" if args.workload == "prove" else "commit"] = {"algorithmic_GiB": round(total_b / 2**30, 3), "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    # integer-multiplier roofline: the peak of v_mad_u64_u32 (the 32x32+64 multiply-add every field multiplication is made of) measured
    # on this device, against the multiply-adds the kernels execute: NTT launches count theirs (18 per table-pair multiplication), the
    # constraint kernels are priced with the static instruction counts of the current build (distaff_amd/_build_info.json)
    mad_iters = 2048
    # calibration kernels: test / bench build (libdistaff_hip_hooks.so), opened beside the measured product library.  The line must not
    # be lost when that build is missing or its kernels fail: the peaks then fall back to the figures measured on this pool and say so.
    calibration_note = None
    try:
        cal = D.Calibration(device)
        mad_ms = cal.bench_mad(1 << 21, mad_iters)
        mad_peak = (1 << 21) * mad_iters * 32 / (mad_ms * 1e-3)
    except Exception as e:                                           # noqa: BLE001
        cal, mad_ms, mad_peak = None, float("nan"), 2.7e13
        calibration_note = "calibration kernels unavailable (%s: %s): mad_peak / mulmod_peak are the pool's usual figures, not measured in this run" % (type(e).__name__, e)
    air_isa = {}
    try:
        air_isa = json.load(open(os.path.join(ROOT, "distaff_amd", "_build_info.json"))).get("air_isa", {})
    except Exception:                                                # noqa: BLE001
        pass
    points = 8 * n // world

    def kernel_mads(name, st):
        if st.get("mads", 0) > 0:
            return st["mads"]
        if name in air_isa:
            return float(air_isa[name]["mad64"]) * points * st["launches"]
        return 0.0
    alu = {"unit": "mad/s (v_mad_u64_u32: 32x32+64 multiply-add per lane)", "peak": mad_peak,
           "peak_source": "mad_peak_kernel, 2^21 lanes x %d iterations x 32 mads, %.3f ms" % (mad_iters, mad_ms)}
    if dom[0]:
        name, st = dom
        m = kernel_mads(name, st)
        alu.update({"kernel": name, "achieved": m / (st["ms"] * 1e-3), "frac": round(m / (st["ms"] * 1e-3) / mad_peak, 4),
                    "mads_per_launch": m / st["launches"]})
    counted = {k: kernel_mads(k, v) for k, v in all_stats.items()}
    proof_mads = sum(counted.values())
    alu["proof"] = {"mads": proof_mads, "achieved": proof_mads / (ms_per_step * 1e-3), "frac": round(proof_mads / (ms_per_step * 1e-3) / mad_peak, 4),
                    "counted_kernels": sorted(k for k, v in counted.items() if v > 0),
                    "note": "multiply-adds of the NTT passes and the constraint kernels (the other kernels' are not counted) over the whole proof time"}
    try:
        alu["valu_issue"] = valu_issue(all_stats, ms_per_step, 1) if default_workload else None      # all_stats: the launches of ONE proof
    except Exception as e:                                           # noqa: BLE001  (never lose the bench line over a summary file)
        alu["valu_issue"] = {"error": str(e)}
    # ALU ceiling: dependent-chain modular multiplications per second measured on this device with the same fe_mul
    try:
        mm_ms = cal.bench_mulmod(1 << 21, 512)
        mulmod_peak = (1 << 21) * 512 * 4 / (mm_ms * 1e-3)
    except Exception as e:                                           # noqa: BLE001
        mulmod_peak = 5.0e11
        calibration_note = calibration_note or "mulmod calibration failed (%s): the pool's usual figure" % e
    if args.workload == "commit":
        workload = ("BASELINE config 2: LDE (iNTT + coset NTTs) + BLAKE3 row hashing + Merkle tree of %d uniform random columns (splitmix64, SURVEY.md 8(d)) of 2^%d steps, "
                    "blowup %d: TraceTable::extend + build_merkle_tree (prover.rs:22-35) only" % (W_FIB, log_n, blowup))
    else:
        workload = ("Fibonacci program (src/examples/fibonacci.rs), 2^%d-step trace, W=20 registers, full stark::prove with %sProofOptions (blowup %d, %d queries, grinding 20, blake3)"
                    % (log_n, "default " if (blowup, args.queries) == (32, 50) else "", blowup, args.queries))
    box = box_fingerprint(cal, mad_peak, mulmod_peak)
    vi = alu.get("valu_issue")
    if isinstance(vi, dict) and vi.get("proof_frac") is not None and (vi.get("sclk_MHz_time_weighted") or box.get("sclk_under_load_MHz")):
        # the slots the device OFFERS shrink with the clock the power management grants under this arithmetic (profiles/r6_power_clock.md):
        # the same instruction counts against 1024 SIMDs x measured clock / 4 cycles.  Above 1 is possible: ~7 % of the instructions are
        # full-rate kinds that take half a slot (profiles/r6_issue_slots.txt).  Clock: per kernel from the committed counter summary
        # (GRBM_GUI_ACTIVE / launch duration) when it has one, else this run's calibration kernel.
        mean_clock = vi.get("sclk_MHz_time_weighted") or box["sclk_under_load_MHz"]
        vi["proof_frac_at_measured_clock"] = round(vi["proof_frac"] * NOMINAL_SCLK_MHZ / mean_clock, 4)
        kc = vi.get("kernel_sclk_MHz") or {}
        vi["kernel_frac_at_measured_clock"] = {k: (None if v is None else round(v * NOMINAL_SCLK_MHZ / (kc.get(k) or mean_clock), 4)) for k, v in vi["kernel_frac"].items()}
        vi["measured_clock_note"] = ("clock per kernel from the stamped counter summary where present (kernel_sclk_MHz), else %s MHz from this run's calibration kernel "
                                     "(field multiplications on every SIMD; box.sclk_under_load_MHz); nominal %.0f MHz" % (box.get("sclk_under_load_MHz"), NOMINAL_SCLK_MHZ))
    out = {
        "metric": "trace_cells_per_sec", "value": value, "unit": "trace-cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
        "dtype": "u128 (prime field 2^128-45*2^40+1, 4x u32 limbs)", "data": "synthetic",
        "config": {"workload": workload,
                   "trace_steps": n, "registers": W_FIB, "blowup": blowup, "queries": args.queries, "grinding": 20,
                   "parallelism": "1 GPU" if (world == 1 and not args.force_sharded) else "one proof sharded over %d GPUs: interpolation by trace columns (coefficients all-gathered), everything on the LDE domain by cosets; "
                                  "Merkle trees finished per k-range (all-to-all of boundary nodes, all-gather of subtree roots), all-gather of constraint evaluations and of the "
                                  "first small FRI layer; %s" % (world, transport)},
        "prover_ms": ms_per_step,
        "phase_ms": None if transport.startswith("torch") else {k: round(v / args.steps, 3) for k, v in zip(phase_names, phase_sum) if args.workload == "prove" or k in ("lde", "trace_merkle")},
        "proof_bytes": len(proof),
        "library": dict(lib_ident, calibration_note=calibration_note),
        "shard_stage_ms_rank0": {k: round(v / args.steps, 3) for k, v in stage_sum.items()} or None,
        # enqueue -> completion of rank 0's collectives per kind and proof, from events on the streams they were queued on (dst_shard_exchange_ms)
        "exchange_ms_rank0": {k: round(v / args.steps, 3) for k, v in exchange_sum.items()} or None,
        "roofline": roofline,
        "roofline_transform": roofline_transform,
        "rooflines": rooflines,
        "step_ms": {"min": round(min(step_ms), 3), "median": round(float(np.median(step_ms)), 3), "max": round(max(step_ms), 3), "all": [round(x, 3) for x in step_ms]},
        "box": box,
        "alu_roofline": dict(alu, mulmod_peak_measured=mulmod_peak, mulmod_kernel="mulmod_bench_kernel: general fe_mul, 4 dependent chains per lane (21 mads + 50 other VALU instructions each)"),
        "phase_hbm": phase_hbm,
        "prover_ms_incl_upload": incl_upload_ms,
        "upload_note": None if incl_upload_ms is None else "trace (%d MiB) in pinned host memory at the start of every step; asynchronous upload in groups of 4 registers, each group interpolated and extended as it lands" % (n * W_FIB * 16 >> 20),
        "kernels": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"], 3), "GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
                    for k, v in sorted(all_stats.items(), key=lambda kv: -kv[1]["ms"])},
        "kernels_note": "one extra untimed proof with every launch bracketed by events; the roofline kernel is timed inside the timed region",
    }
    if world > 1 or args.force_sharded:
        out["devices"] = {"visible": ndev, "ranks": world, "shared": shared_devices,
                          "note": ("%d ranks share %d device(s): a FUNCTIONAL run of the N-process path on this box, not a scaling measurement" % (world, ndev)) if shared_devices else None}
        out["comm"] = comm_info
        if transport_note:
            out["transport_note"] = transport_note
    progress("verification")
    if args.workload == "commit":
        out["proof_bytes"] = None
        out["trace_root_hex"] = proof.hex()
    elif not args.no_verify:
        # the checker, outside every timed region: the oracle's restatement of the reference verifier must accept the timed proof
        import oracle as O
        ok, err = O.verify(proof, program_hash, [1, 0], [result])
        if not ok:
            raise SystemExit("the oracle's verifier rejects the timed proof: " + err)
        out["proof_verified"] = "accepted by oracle/verifier.hpp (restatement of stark::verify) after the timed regions"
    progress("cpu baseline")
    if world == 1 and not args.no_cpu_baseline:
        if args.workload == "commit":
            out["cpu_baseline"] = cpu_baseline_commit(cols, blowup)
            if out["cpu_baseline"]["root_hex"] != proof.hex():
                raise SystemExit("the oracle's trace root differs from the GPU's on the same columns")
            out["proof_verified"] = "trace root equals the oracle's (same columns, CPU leg of this run)"
        else:
            out["cpu_baseline"] = cpu_baseline(args.cpu_log_n, blowup, args.queries)
    emit(json.dumps(out))
    if cal is not None:
        cal.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
