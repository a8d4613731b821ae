"""Host side of libdistaff_hip.so that needs no GPU (runs under -m "not gpu"): the Fibonacci trace generator (host_vm.h), the
Fiat-Shamir helpers (host_util.h: BLAKE3, StdRng/Uniform draws, query positions) against the oracle's restatements, which are
themselves pinned by the reference's vectors (tests/test_oracle_*.py)."""
import os
import numpy as np
import pytest


@pytest.mark.parametrize("log_n", [7, 8, 10, 13])
def test_library_trace_generator_equals_oracle_vm(oracle, log_n):
    """dst_fibonacci_trace (host_vm.h: decoder, Rescue sponge accumulator, context stack, user stack, VOID padding) against
    the oracle VM on the assembled program (processor/mod.rs:23-143, programs/blocks/mod.rs:149-178)."""
    import distaff_amd as D
    cols, program_hash, result = D.fibonacci_trace(log_n)
    t = oracle.fibonacci_trace(1 << log_n)
    assert cols.shape == (20, 1 << log_n, 2)
    assert (cols == t.columns).all()
    assert program_hash == t.program_hash
    assert result == oracle.to_ints(t.columns[16, -1:])[0]            # top of the user stack in the last row


def test_host_field_and_inverse_sbox_chain(tmp_path):
    """host_vm.h's field on 64-bit limbs (product folded twice with 2^128 = 45 * 2^40 - 1, dedicated squaring) and the four-lane
    addition chain for x -> x^((2p - 1) / 3) (utils/sponge.rs:70) against Python integers: every pair of 16 edge values, 4000 seeded
    pairs, 1600 inverse-S-box values (the chain, square-and-multiply and the cube of the result)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_field_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "distaff_amd", "csrc"), "-I", os.path.join(root, "tests", "emu"),
                    "-o", exe, os.path.join(root, "tests", "hostfield", "host_field_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    P = 2**128 - 45 * 2**40 + 1
    E = (2 * P - 1) // 3
    seen = {"m": 0, "p": 0}
    for line in out.splitlines():
        f = line.split()
        v = [int(x, 16) for x in f[1:]]
        if f[0] == "c":
            # 127 squarings: the exponent the chain implements is below 2^128, and the only exponent below 2^128 that agrees with x^E on
            # every x is E itself (E + (p - 1) > 2^128) -- round 5's chain had 143: a 136-bit exponent congruent to E modulo p - 1
            assert v == [127, 12], v
            continue
        if f[0] == "m":
            a, b, m, sq = v
            assert m == a * b % P and sq == a * a % P, (hex(a), hex(b))
        else:
            x, chain, ladder = v
            assert chain == ladder == pow(x, E, P) and pow(chain, 3, P) == x, hex(x)
        seen[f[0]] += 1
    assert seen == {"m": 256 + 4000, "p": 1600} and any(l.startswith("c ") for l in out.splitlines())


def test_library_fiat_shamir_helpers_equal_oracle(oracle):
    import distaff_amd as D
    O = oracle
    for i in range(20):
        seed = O.blake3(bytes([i]) * (i + 1))
        assert D.blake3(bytes([i]) * (i + 1)) == seed
        for count in (1, 2, 344, 516):
            assert (D.prng_vector(seed, count) == O.prng_vector(seed, count)).all()
        for (N, ext, nq) in ((1 << 12, 32, 50), (1 << 25, 32, 50), (1 << 28, 16, 100), (1 << 10, 64, 30)):
            got = D.query_positions(seed, N, ext, nq)
            assert got == O.query_positions(seed, N, ext, nq)
            assert all(p % ext != 0 for p in got) and len(set(got)) == len(got)


def test_blake3_lengths(oracle):
    import distaff_amd as D
    data = bytes(range(256)) * 5
    data = data * 60
    for n in (0, 1, 31, 32, 33, 63, 64, 65, 127, 128, 1023, 1024, 1025, 2048, 2049, 3072, 3073, 4096, 5000, 7 * 1024 + 1, 54239, 65536):
        assert D.blake3(data[:n]) == oracle.blake3(data[:n])


def test_query_positions_argument_checks():
    """dst_query_positions returns an error code for a zero domain / extension factor / query count instead of dividing by zero"""
    import distaff_amd as D
    seed = bytes(range(32))
    assert len(D.query_positions(seed, 1 << 12, 32, 50)) == 50
    for args in ((0, 32, 50), (1 << 12, 0, 50), (1 << 12, 32, 0), (1 << 12, 32, 129)):
        with pytest.raises(D.DistaffError):
            D.query_positions(seed, *args)


def test_context_creation_validates_and_fails_loudly_without_gpu():
    """The product path has no CPU fallback: bad parameters are argument errors; on a box without a GPU a valid request is a HIP error."""
    import torch
    import distaff_amd as D
    for kw in (dict(log_n=3), dict(log_n=25), dict(log_blowup=3), dict(log_blowup=9), dict(width=15), dict(width=128), dict(num_queries=0),
               dict(world=3), dict(world=16), dict(rank=2, world=2)):
        args = dict(log_n=8, width=20, ctx_depth=1, loop_depth=0, log_blowup=5, num_queries=50, grinding=20, rank=0, world=1)
        args.update(kw)
        with pytest.raises(D.DistaffError) as e:
            D.Context(args.pop("log_n"), args.pop("width"), args.pop("ctx_depth"), args.pop("loop_depth"), **args)
        assert e.value.code == D.DST_ERR_ARG, kw
    if not torch.cuda.is_available():
        with pytest.raises(D.DistaffError) as e:
            D.Context(8, 20, 1, 0)
        assert e.value.code == D.DST_ERR_HIP


def _compile_c_host(tmp_path, name="prove_fibonacci"):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", name + ".c"),
                           "-L" + os.path.join(root, "distaff_amd"), "-ldistaff_hip", "-Wl,-rpath," + os.path.join(root, "distaff_amd"), "-o", exe])
    return exe


def test_c_abi_header_is_plain_c_and_links(tmp_path):
    """include/distaff_hip.h compiles as C99 and examples/prove_fibonacci.c links against libdistaff_hip.so (no torch, no Python)."""
    import subprocess
    import torch
    exe = _compile_c_host(tmp_path)
    _compile_c_host(tmp_path, "prove_sharded")             # the multi-GPU host links against the same library, nothing else
    if not torch.cuda.is_available():                      # without a GPU the host fails loudly at context creation (no CPU fallback)
        r = subprocess.run([exe, "8"], capture_output=True, text=True)
        assert r.returncode == 1 and "dst_ctx_create" in r.stderr


def test_limb_level_field_arithmetic_on_the_host(tmp_path):
    """fe.h's gfx950 formulations (windowed and table-pair multiplication, the nine-limb reduction with its overflow limb, sum/difference
    pairs, sums of products) are written over carry primitives that have a plain-integer host form: tools/felab/host_test.cpp runs
    that dataflow on 6 million random and edge-case operands against the portable multiplication, which the oracle tests pin.  (The
    device form of the same primitives -- one instruction each -- is pinned on the GPU by test_device_field_arithmetic.)"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "fe_host_test"
    subprocess.check_call(["g++", "-O2", "-o", str(exe), os.path.join(root, "tools", "felab", "host_test.cpp")])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b"bad=0" in r.stdout, r.stdout.decode()[-2000:]


def test_bench_reads_the_stamped_counter_summary():
    """bench.py takes counter-derived figures (roofline.traffic, alu_roofline.valu_issue) from profiles/*_pmc_per_kernel.csv only while the
    summary's stamp equals the digest of the kernel sources; kernel names match with or without blanks and with template booleans as
    0 / 1.  With a stale summary every lookup is refused with a reason -- never a number from other code."""
    import json, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    row, why = bench.pmc_row("ntt_pass_a")
    meta = json.load(open(sorted(p for p in (os.path.join(root, "profiles", f) for f in os.listdir(os.path.join(root, "profiles"))) if p.endswith("_meta.json"))[-1]))
    if meta["csrc_sha16"] != bench.csrc_digest():
        assert row is None and "refused" in why
        assert bench.valu_issue({"ntt_pass_a": {"launches": 5, "ms": 12.0}}, 38.0, 1) is None
        return
    assert row is not None and float(row["SQ_INSTS_VALU"]) > 0 and int(row["fetch_bytes_per_launch_x2"]) > 0
    assert bench.pmc_row("air_kernel<2,1,4,8,0,240,6>")[0] is not None         # the library prints unsigned template arguments as 240, rocprofv3 as 240u
    assert bench.pmc_row("ntt_pass_b<1024,8,1,10,2,0>")[0] is not None           # ... and template booleans as 0 / 1, rocprofv3 as false / true
    assert bench.pmc_row("no_such_kernel")[0] is None
    newest = sorted(f for f in os.listdir(os.path.join(root, "profiles")) if f.endswith("_bench_default.json"))[-1]
    line = json.loads(open(os.path.join(root, "profiles", newest)).read().strip().splitlines()[-1])
    stats = {k: {"launches": v["launches"], "ms": v["ms_per_step"]} for k, v in line["kernels"].items()}
    v = bench.valu_issue(stats, line["ms_per_step"], 1)
    assert 0.5 < v["proof_frac"] < 1.0 and all(0.3 < f < 1.0 for f in v["kernel_frac"].values()) and not v["not_counted"]


def test_kernel_gate_on_the_built_library():
    """What build() enforces (__graft_entry__._kernel_gate): EVERY kernel of the product library is at most 64 KiB of code (the instruction
    cache of a gfx950 CU pair), spills no vector register and uses no scratch -- no exceptions; the laboratory kernels that do exceed the
    limits (listed with their reasons) exist only in the test / bench build.  Read from the code objects of the built libraries
    (tools/codeobj_info.py), no GPU needed.  The constraint launches of the bench path are named."""
    import os, re, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
    import __graft_entry__ as G
    import codeobj_info
    so = os.path.join(root, "distaff_amd", "libdistaff_hip.so")
    hooks_so = os.path.join(root, "distaff_amd", "libdistaff_hip_hooks.so")
    if not os.path.exists(so) or not os.path.exists(hooks_so):
        G.build()
    kernels = codeobj_info.kernels_of(so)
    hooks = codeobj_info.kernels_of(hooks_so)
    assert len(kernels) > 60
    assert G._kernel_gate(kernels) == []                              # the product: no exception list at all
    assert G._kernel_gate(hooks, G.GATE_EXCEPTIONS) == [] and set(kernels) < set(hooks)
    for pat in G.GATE_EXCEPTIONS:                                     # no stale exception, and none of them is in the product
        assert any(re.match(re.escape(pat), k) for k in hooks), pat
        assert not any(re.match(re.escape(pat), k) for k in kernels), pat
    assert not any(k.startswith(("code_probe_kernel", "mulmod_bench_kernel", "mad_peak_kernel", "air_kernel<16,8,0,32,")) for k in kernels)
    bench_path = [k for k in kernels if k.startswith("air_kernel<2,1,4,8,") and not k.startswith("air_kernel<2,1,4,8,1,")]
    assert len(bench_path) == 5
    for k in bench_path:
        assert kernels[k]["code_bytes"] <= G.CODE_LIMIT and kernels[k]["vgpr_spill"] == 0 and kernels[k]["scratch_bytes"] == 0, (k, kernels[k])
    # the gate itself: a kernel over the limit, a spill and scratch are each reported
    fake = {"some_kernel<1>": {"code_bytes": G.CODE_LIMIT + 4, "vgpr_spill": 0, "scratch_bytes": 0},
            "other": {"code_bytes": 100, "vgpr_spill": 3, "scratch_bytes": 16}}
    bad = G._kernel_gate(fake)
    assert len(bad) == 2 and "bytes of code" in bad[1] and "spilled" in bad[0] and "scratch" in bad[0]


def test_merged_stack_terms_equal_the_per_operation_sum(tmp_path):
    """air_kernel.h's st_desc / st_merge tables (operations of a group sharing one product per slot, zero items above the compile-time
    stack depth) against the plain sum of flag x st_term over all 32 low-degree operations, on random rows and random flag factors, for
    every slot, both auxiliary constraints, stack depths 4..8 and the run-time-depth / deep instances -- including the operations no
    tested program executes (their flags are zero in every proof, so proof-level parity cannot see their table entries).  Host build of
    the kernel header against the stand-in HIP header of tests/emu."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu = os.path.join(root, "tests", "emu")
    exe = tmp_path / "air_merge_test"
    subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-pthread", "-I", emu, "-w", "-DFE_EMULATE_GFX950=1", os.path.join(emu, "air_merge_test.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0 and b" 0 mismatches" in out.stdout, out.stdout.decode()[-2000:]


def test_bench_refuses_another_library_than_the_product_build(tmp_path):
    """DISTAFF_HIP_LIB points the binding at any other build (the tests' CPU emulation included): bench.py must not measure that silently.
    It ends with the contract's error line; --allow-lib-override is the explicit way (tests/emu/bench_harness.py passes it)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    other = tmp_path / "libother.so"
    other.write_bytes(open(os.path.join(root, "distaff_amd", "libdistaff_hip.so"), "rb").read())
    env = dict(os.environ, DISTAFF_HIP_LIB=str(other))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert d["value"] is None and "refusing to measure" in d["error"] and "libother.so" in d["error"]
