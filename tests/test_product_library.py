"""The product library (distaff_amd/libdistaff_hip.so) against the test build (libdistaff_hip_hooks.so): what it exports, what it does
not contain, which environment switches exist and that INTEGRATION.md lists them.  No GPU needed except for the one test marked so."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "distaff_amd", "csrc")


def _switch_table():
    text = open(os.path.join(CSRC, "ctx.h")).read()
    return {m[0]: m[1] == "true" for m in re.findall(r'\{"(DISTAFF_[A-Z0-9_]+)",\s*(true|false),', text)}


def _declared():
    text = open(os.path.join(ROOT, "include", "distaff_hip.h")).read()
    return set(re.findall(r"^DST_API [^;(]*?\b(dst_[a-z0-9_]+)\(", text, flags=re.M))


def _exports(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
    return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("dst_")}


def test_libraries_export_exactly_the_declared_entry_points():
    """hidden visibility: no dst_internal_* hook, no helper leaves either library; every declaration of include/distaff_hip.h is defined"""
    declared = _declared()
    assert len(declared) > 50
    for name in ("libdistaff_hip.so", "libdistaff_hip_hooks.so"):
        assert _exports(os.path.join(ROOT, "distaff_amd", name)) == declared, name


def test_product_and_test_build_identify_themselves():
    import ctypes
    import distaff_amd as D
    assert ctypes.CDLL(D.PRODUCT_LIB).dst_test_hooks() == 0
    assert ctypes.CDLL(D.HOOKS_LIB).dst_test_hooks() == 1
    if not os.environ.get("DISTAFF_HIP_LIB"):
        assert os.path.realpath(D.library_path()) == os.path.realpath(D.HOOKS_LIB)      # tests/conftest.py bound the test build


def test_environment_is_read_once_per_context_and_documented():
    """every switch the sources look up is in ctx.h's DST_SWITCHES and in INTEGRATION.md section 6; nothing calls getenv but
    dst_ctx::read_switches (context creation) and the communicator constructors (DISTAFF_SHARD_DEBUG, the wait limit, the test build's
    fault injection: a communicator has no context)"""
    table = _switch_table()
    assert len(table) >= 20 and sum(table.values()) == 6
    used, getenv_sites = set(), []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(CSRC, f)).read()
        used |= set(re.findall(r'sw(?:_is|_flag)?\("(DISTAFF_[A-Z0-9_]+)"', text))
        getenv_sites += [(f, m) for m in re.findall(r'getenv\(([^)]*)\)', text)]
    comm_only = {"DISTAFF_COMM_TIMEOUT_S", "DISTAFF_LOCAL_TRANSPORT", "DISTAFF_TEST_STALL_COLLECTIVE"}
    assert used | comm_only == set(table) and not (used & comm_only), (used ^ set(table))
    assert sorted(getenv_sites) == [("comm.hip", '"DISTAFF_COMM_TIMEOUT_S"'), ("comm.hip", '"DISTAFF_LOCAL_TRANSPORT"'), ("comm.hip", '"DISTAFF_SHARD_DEBUG"'), ("comm.hip", '"DISTAFF_TEST_STALL_COLLECTIVE"'), ("ctx.h", "d.name")], getenv_sites
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name, product in table.items():
        row = re.search(r"^\| `%s` \| (product \+ test|test only) \|" % name, doc, flags=re.M)
        assert row, name
        assert (row.group(1) == "product + test") == product, name


PRODUCT_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.pop("DISTAFF_TEST_HOOKS", None)
os.environ.update(DISTAFF_AIR="generic", DISTAFF_COMBINE="steps", DISTAFF_NTT="3pass", DISTAFF_FRI_CHAIN="0", DISTAFF_MERKLE_LEVELS="1")
import json
import distaff_amd as D
assert D.load().dst_test_hooks() == 0 and D.library_path() == D.PRODUCT_LIB
cols, program_hash, result = D.fibonacci_trace(12)
ctx = D.Context(12, 20, 1, 0)
ctx.upload(cols)
ctx.set_profiling(1)
ctx.kernel_stats(reset=True)
proof = ctx.prove([1, 0], [result])
names = sorted(ctx.kernel_stats())
try:
    ctx.bench_mad(1 << 10, 4)
    code = 0
except D.DistaffError as e:
    code = e.code
open(sys.argv[1], "wb").write(proof)
json.dump({"kernels": names, "bench_mad": code}, open(sys.argv[2], "w"))
"""


@pytest.mark.gpu
def test_product_library_ignores_the_test_switches(tmp_path):
    """The product build with test-only switches set in its environment: same proof as the default formulations, none of the alternative
    kernels launched (it does not contain them), the calibration entry points answer DST_ERR_STATE."""
    import json
    import distaff_amd as D
    script = tmp_path / "w.py"
    script.write_text(PRODUCT_WORKER % ROOT)
    subprocess.check_call([sys.executable, str(script), str(tmp_path / "proof.bin"), str(tmp_path / "info.json")])
    info = json.load(open(tmp_path / "info.json"))
    assert info["bench_mad"] == D.DST_ERR_STATE
    assert not any(k.startswith(("air_kernel<16,8,0,32", "ntt_pass_mid")) for k in info["kernels"]), info["kernels"]
    assert any(k.startswith("air_kernel<2,1,4,8,") for k in info["kernels"]) and "combine_fused_kernel" in info["kernels"]
    cols, program_hash, result = D.fibonacci_trace(12)
    ctx = D.Context(12, 20, 1, 0)                 # the test build, default switches
    ctx.upload(cols)
    assert ctx.prove([1, 0], [result]) == (tmp_path / "proof.bin").read_bytes()
    ctx.close()


@pytest.mark.gpu
def test_parity_selection_on_the_product_library_itself():
    """The phase-by-phase parity tests bind the test build (same sources, more switches).  A selection of them -- those that need no test-only
    switch -- is run here against distaff_amd/libdistaff_hip.so itself: every intermediate of every phase and the proof bytes, Fibonacci
    traces, other program shapes, traces of the whole instruction set with loops, the sharded prover with thread-ranks -- and the FULL-SIZE
    checks (config 3: sampled oracle point computations in every phase at 2^20; config 5: 2^24 steps, the oracle's verifier accepts, a tampered
    proof is rejected), which before round 6 had only ever run on the test build, a different code object."""
    selection = ["test_which_library_is_bound", "test_fibonacci_all_phases", "test_other_program_shapes", "test_program_shapes_with_stack_depth_5_to_8", "test_blowup_16_and_64",
                 "test_tiny_traces_of_32_and_16_rows", "test_thread_rank_transport_issue_order_and_peer_access", "test_loops_and_macros_at_2_13_and_2_15",
                 "test_whole_instruction_set_and_flow_blocks and not generic", "test_prove_sharded_peer_that_never_arrives",
                 "test_config3_sampled_oracle_parity_at_full_size", "test_config5_full_size_on_one_gpu"]
    env = dict(os.environ, DISTAFF_PRODUCT_ONLY="1")
    env.pop("DISTAFF_TEST_HOOKS", None)
    env.pop("DISTAFF_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        "-k", " or ".join("(%s)" % k for k in selection)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=2400)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert " passed" in out and "failed" not in out, out[-1000:]
