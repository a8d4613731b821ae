"""The kernels' LOGIC on the CPU: a selection of the GPU parity tests run against tests/emu/_build/libdistaff_emu.so.

libdistaff_emu.so is the library's own sources (distaff_amd/csrc/*.hip) compiled by g++ against a stand-in HIP header in which a
kernel launch is executed on the host, one fiber per work-item (tests/emu/hip/hip_runtime.h).  It is test infrastructure: the
package never picks it up by itself (it is selected here, in a subprocess, through DISTAFF_HIP_LIB), `bench.py`, `smoke()` and the
`-m gpu` run use libdistaff_hip.so on the device, and the gfx950 formulation of the field arithmetic is NOT exercised by it (the
host branch of fe.h is) -- that part is pinned on the GPU by test_device_field_arithmetic.  What it does pin, where there is no
GPU: tile / twiddle / tree-level indexing of every kernel, the constraint instances, the sharded layouts and exchanges, the opening
plan and the wire format -- the same assertions as on the GPU, bit for bit against the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "_build", "libdistaff_emu.so")

SELECTION = [
    "test_fibonacci_all_phases[7]",
    "test_fibonacci_all_phases[10]",
    "test_boundary_constraints_by_evaluation[]",
    "test_combination_and_composition_as_whole_array_steps",
    "test_all_phases_with_register_pre_stages",
    "test_merkle_levels_two_per_launch_on_small_trees",
    "test_lde_every_tile_length[pre-13-5]",
    "test_other_program_shapes",
    "test_program_shapes_with_stack_depth_5_to_8",
    "test_blowup_16_and_64",
    "test_blowup_128_and_256[128-7]",
    "test_config2_random_columns_lde_and_merkle[12]",
    "test_invalid_trace_reports_air_error",
    "test_repeated_query_positions_are_refused",
    "test_tiny_traces_of_32_and_16_rows",
    "test_wide_rows_two_chunk_leaves",
    "test_sharded_prover_equals_single_gpu[2]",
    "test_sharded_prover_reports_invalid_trace",
    "test_sharded_fri_protocol_state_errors",
    "test_prove_sharded_behind_the_c_abi[2-8-None-False]",
    "test_prove_sharded_behind_the_c_abi[8-8-9-False]",
    "test_prove_sharded_behind_the_c_abi[4-10-9-True]",
    "test_prove_sharded_fri_layers_with_host_draws",
    "test_prove_sharded_reports_an_invalid_trace_on_every_rank",
    "test_prove_sharded_rank_without_a_trace_on_fresh_contexts",
    "test_prove_sharded_peer_that_never_arrives",
    "test_host_side_abort_from_another_thread",
    "test_sharded_phase_and_exchange_times",
    "test_prove_sharded_in_separate_processes_sharing_the_gpu[2-12-one]",
    "test_prove_sharded_in_separate_processes_sharing_the_gpu[8-12-overlap]",
    "test_bench_with_n_processes_on_one_device[2]",
    "test_bench_with_n_processes_on_one_device[8]",
    "test_bench_prints_an_error_line_instead_of_hanging",
    "test_bench_config2_line",
    "test_sampled_oracle_parity_small[7-5]",
    "test_sampled_oracle_parity_small[10-4]",
    "test_deep_stacks_and_nested_blocks[]",
    "test_deep_stacks_and_nested_blocks[generic]",
    "test_general_constraint_instances_on_the_fibonacci_trace[small]",
    "test_general_constraint_instances_on_the_fibonacci_trace[deep]",
    "test_general_constraint_instances_on_the_fibonacci_trace[generic]",
    "test_lde_every_tile_length[reg-13-5]",
    "test_lde_every_tile_length[lds-13-5]",
    "test_lde_every_tile_length[dit2-13-5]",
    "test_lde_every_tile_length[waves8-13-5]",
    "test_lde_every_tile_length[lds-16-5]",
    "test_trace_from_pinned_host_memory[w17]",
    "test_synthetic_division_by_power_tables",
    "test_trace_in_its_own_buffer",
    "test_small_fri_layers_in_one_launch_equal_the_per_layer_path",
    "test_prove_sharded_with_collectives_on_their_own_stream[2]",
    "test_isa_traces_cover_every_operation",
    "test_sharded_prover_on_loop_and_macro_traces[2]",
    "test_gpu_proofs_of_loop_and_macro_traces_match_golden_digests",
] + ["test_whole_instruction_set_and_flow_blocks[%s-%s]" % (name, instance) for name, instance in (
    ("stack_manipulation", ""), ("choose2", "generic"), ("cswap2", ""), ("math_inv_neg_not", ""), ("bool_and_or", "generic"), ("read_read2", ""), ("eq", ""),
    ("rescr_double_hash", ""), ("cmp_128", ""), ("binacc_128", "generic"), ("if_true", "generic"), ("if_false", ""), ("while_skipped", ""),
    ("while_5_iterations", ""), ("while_5_iterations", "generic"), ("nested_loops", ""), ("nested_loops", "generic"), ("example_comparison", ""),
    ("example_merkle", ""), ("example_range", "generic"))]


@pytest.fixture(scope="module")
def emulated_library():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j8"], stdout=subprocess.DEVNULL)
    assert os.path.exists(EMU_LIB)
    return EMU_LIB


def _run(emulated_library, args, timeout=900):
    env = dict(os.environ, DISTAFF_HIP_LIB=emulated_library, DISTAFF_HIP_RUNTIME="none", DISTAFF_EMU_THREADS="2",
               DISTAFF_BENCH_ENTRY=os.path.join(EMU_DIR, "bench_harness.py"))           # the tests that launch bench.py go through the CPU stand-ins
    # the selected tests are independent: four pytest-xdist workers (each launch of the emulated runtime uses two host threads) when the
    # plugin is there, otherwise in sequence
    import importlib.util
    workers = ["-n", "4"] if importlib.util.find_spec("xdist") is not None and (os.cpu_count() or 1) >= 8 and len(args) > 8 else []
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu"] + workers + args,
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)


def test_parity_selection_on_the_emulated_build(emulated_library):
    ids = ["tests/test_gpu_parity.py::" + t for t in SELECTION]
    r = _run(emulated_library, ids)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-4000:]
    assert "%d passed" % len(SELECTION) in out, out[-2000:]


def test_emulated_build_is_not_a_product_path():
    """nothing under distaff_amd/, bench.py or __graft_entry__.py refers to the emulated build"""
    import glob
    files = glob.glob(os.path.join(ROOT, "distaff_amd", "**", "*"), recursive=True) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for f in files:
        if os.path.isfile(f) and not f.endswith((".so", ".pyc", ".o")):
            text = open(f, errors="ignore").read()
            assert "libdistaff_emu" not in text and "tests/emu" not in text, f


DEVICE_PATH_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import oracle as O
import distaff_amd as D
from distaff_amd.sharded import ShardedProver, TorchComm
dist.init_process_group("gloo")
# the hand-off mode bench.py uses on GPUs: the library reads / writes the collective's tensors directly through their data pointers
comm = TorchComm(dist, torch.device("cpu"), device_path=True)
assert comm.device_path
t = O.fibonacci_trace(256)
p = O.Prover.from_trace(t, 1, grinding=8)
expected = p.prove()
ctx = D.Context(8, t.width, t.ctx_depth, t.loop_depth, rank=comm.rank, world=comm.world, grinding=8)
ctx.upload(t.columns)
prover = ShardedProver(ctx, comm)
for _ in range(2):                                   # twice: the buffers of the first proof are reused
    assert prover.prove(t.public_inputs, p.outputs) == expected, "rank %%d: proof differs from the oracle's" %% comm.rank
assert ShardedProver(ctx, comm, python_openings=True).prove(t.public_inputs, p.outputs) == expected
ctx.close()
comm.barrier()
dist.destroy_process_group()
print("ok", comm.rank)
'''


@pytest.mark.parametrize("world,replicate_log", [(2, 9), (4, 17)])
def test_tensor_hand_off_path_over_gloo(emulated_library, tmp_path, world, replicate_log):
    """`world` real processes over torch.distributed (gloo), each with its own context of the emulated build, exchanging shards
    through tensor data pointers (`is_device` = 1 in the C-ABI) -- the branch of ShardedProver that bench.py takes with RCCL on
    GPUs, which the GPU box of the test run can only exercise with one rank."""
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "device_path_worker.py"
    script.write_text(DEVICE_PATH_WORKER % {"root": ROOT})
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DISTAFF_HIP_LIB=emulated_library, DISTAFF_HIP_RUNTIME="none", DISTAFF_FRI_REPLICATE_LOG=str(replicate_log), DISTAFF_EMU_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


@pytest.mark.parametrize("ranks,orch", [(1, "c"), (2, "python")])      # (N, "c") for N = 2, 8: test_bench_with_n_processes_on_one_device in the selection above
def test_bench_control_flow_on_cpu_stand_ins(emulated_library, ranks, orch):
    """bench.py's own main() launched exactly as the driver launches it for N > 1 (`python -m torch.distributed.run ...`), with gloo,
    CPU tensors and the emulated build standing in for RCCL, device tensors and the GPU (tests/emu/bench_harness.py): one JSON line
    from rank 0 with the contract's fields.  For N = 2 both orchestrations: dst_prove_sharded with its collectives carried by gloo
    through the callback transport (two real processes), and the host-orchestrated sequence of distaff_amd/sharded.py with the
    tensor hand-off.  The numbers mean nothing here."""
    import json
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, DISTAFF_HIP_LIB=emulated_library, DISTAFF_HIP_RUNTIME="none", DISTAFF_EMU_THREADS="2", DISTAFF_SHARD_ORCH=orch,
               DISTAFF_SHARD_HANDOFF="device")       # CPU tensors and the emulated build's host pointers: the tensor hand-off bench.py uses with device tensors on GPUs
    harness = os.path.join(EMU_DIR, "bench_harness.py")
    args = ["--gpus", str(ranks), "--steps", "1", "--warmup", "1", "--log-n", "8", "--cpu-log-n", "7"]
    if ranks == 1:
        cmd = [sys.executable, harness] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port), harness] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == ranks and out["steps"] == 1 and out["vs_baseline"] is None and "workload" in out["config"]
    assert out["scaling"] == ("weak" if ranks == 1 else "strong")
    assert ("cpu_baseline" in out) == (ranks == 1)
    assert out["proof_verified"]
    if ranks > 1 and orch == "python":
        assert "torch.distributed, device" in out["config"]["parallelism"] and out["shard_stage_ms_rank0"]
    if ranks > 1 and orch == "c":
        assert "dst_prove_sharded" in out["config"]["parallelism"] and out["phase_ms"]
    if ranks == 1:
        assert out["prover_ms_incl_upload"] and out["phase_hbm"] and "frac" in out["alu_roofline"]


SILENT_PEER_WORKER = r'''
import datetime, json, os, sys, time
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import distaff_amd as D
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=8))       # the HOST's channel carries its own limit
rank, world = dist.get_rank(), dist.get_world_size()
cols, program_hash, result = D.fibonacci_trace(8)
ctx = D.Context(8, 20, 1, 0, rank=rank, world=world, grinding=8)
ctx.upload(cols)
comm = D.Comm.over_torch(dist)                                               # dst_comm_init_callbacks: gloo carries the library's collectives
out = {"rank": rank}
if rank == 1:
    time.sleep(16)                                                           # this rank never enters the proof and stops answering
    out["code"] = None
else:
    t0 = time.time()
    try:
        ctx.prove_sharded(comm, [1, 0], [result])
        out["code"] = 0
    except D.DistaffError as e:
        out["code"], out["message"] = e.code, str(e)
    out["seconds"], out["comm_error"] = time.time() - t0, comm.last_error()
json.dump(out, open(os.path.join(sys.argv[1], "result_%%d.json" %% rank), "w"))
os._exit(0)                                                                  # no orderly shutdown of a broken group
'''


def test_peer_process_that_stops_answering_over_gloo(emulated_library, tmp_path):
    """Containment of the N-process path without a GPU: two real processes over torch.distributed (gloo), the library's collectives carried by
    the callback transport; rank 1 never enters dst_prove_sharded.  Rank 0 does not hang: when the host's channel gives up (gloo, 8 s) its
    callback reports failure, the communicator is dead and dst_prove_sharded returns DST_ERR_COMM naming the collective (the first one)."""
    import json
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "silent_peer_worker.py"
    script.write_text(SILENT_PEER_WORKER % {"root": ROOT})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DISTAFF_HIP_LIB=emulated_library, DISTAFF_HIP_RUNTIME="none", DISTAFF_EMU_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    res = json.load(open(tmp_path / "result_0.json"))
    assert res["code"] == -5, (res, outs)                                    # DST_ERR_COMM
    assert "callback returned" in res["comm_error"] and "collective #0 (all-gather" in res["comm_error"], res
    assert res["seconds"] < 60.0, res


THREE_PASS_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
os.environ["DISTAFF_NTT"] = "3pass"                      # the plan the library takes by itself from 2^21 points on
import numpy as np
import oracle as O
import distaff_amd as D
for log_n, log_b in ((12, 5), (13, 4), (15, 4)):
    n, B, W = 1 << log_n, 1 << log_b, 16
    rng = np.random.default_rng(log_n)
    cols = rng.integers(0, 2**63, size=(W, n, 2), dtype=np.uint64)
    ctx = D.Context(log_n, W, 0, 0, log_blowup=log_b)
    ctx.upload(cols)
    ctx.commit_trace()
    for c in (0, 9, 15):
        poly = ctx.read_elements("polys").reshape(W, n, 2)[c]
        lde = ctx.read_elements("lde", c)
        g_n, g_N = O.root_of_unity(n), O.root_of_unity(n * B)
        for k in (0, 1, n - 1, 77):
            assert O.poly_eval(poly, O.exp(g_n, k)) == O.to_ints(cols[c, k:k + 1])[0], ("interpolation", log_n, c, k)
        for i in (1, B - 1, B, n * B - 1, 12345):
            assert O.poly_eval(poly, O.exp(g_N, i)) == O.to_ints(lde[i:i + 1])[0], ("extension", log_n, c, i)
        assert (lde[::B] == cols[c]).all()
    ctx.close()
print("ok")
'''


def test_three_pass_transform_plan_at_small_sizes(emulated_library, tmp_path):
    """The three-pass NTT plan (n = n1 * nm * n3, the default from 2^21 points, GPU-tested at 2^20 .. 2^24) forced onto 2^12 .. 2^15
    points so that its index arithmetic is also checked where there is no GPU: interpolation and extension against Horner
    evaluations by the oracle."""
    script = tmp_path / "three_pass_worker.py"
    script.write_text(THREE_PASS_WORKER % {"root": ROOT})
    env = dict(os.environ, DISTAFF_HIP_LIB=emulated_library, DISTAFF_HIP_RUNTIME="none")
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stdout.decode()[-3000:]


FIXED_SHAPE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import oracle as O
import distaff_amd as D
for log_n, log_b, W in ((20, 4, 16), (16, 4, 18)):       # 18 registers: the last chunk of the coset-fast block order holds two
    n, B = 1 << log_n, 1 << log_b
    rng = np.random.default_rng(log_n)
    cols = rng.integers(0, 2**63, size=(W, n, 2), dtype=np.uint64)
    ctx = D.Context(log_n, W, 0, 0, log_blowup=log_b)
    ctx.upload(cols)
    ctx.commit_trace()
    g_n, g_N = O.root_of_unity(n), O.root_of_unity(n * B)
    polys = ctx.read_elements("polys").reshape(W, n, 2)
    for c in (0, 9, W - 1):
        poly = polys[c]
        lde = ctx.read_elements("lde", c)
        for k in (0, 1, n - 1, 7777):
            assert O.poly_eval(poly, O.exp(g_n, k)) == O.to_ints(cols[c, k:k + 1])[0], ("interpolation", log_n, c, k)
        for i in (1, B - 1, B, n * B - 1, 123456, 7 * n + 5):
            assert O.poly_eval(poly, O.exp(g_N, i)) == O.to_ints(lde[i:i + 1])[0], ("extension", log_n, c, i)
        assert (lde[::B] == cols[c]).all()
    ctx.close()
print("ok")
'''


def test_transform_instances_of_the_bench_size(emulated_library, tmp_path):
    """n = 2^20 runs on instances compiled for its 1024 x 4 tiles (1024-lane workgroups, rounds with literal strides, coset-fast block
    order): interpolation and a 16-fold extension of 16 random registers on the emulated build against Horner evaluations by the oracle."""
    script = tmp_path / "fixed_shape_worker.py"
    script.write_text(FIXED_SHAPE_WORKER % {"root": ROOT})
    env = dict(os.environ, DISTAFF_HIP_LIB=emulated_library, DISTAFF_HIP_RUNTIME="none")
    for k in ("DISTAFF_NTT", "DISTAFF_NTT_FIXED", "DISTAFF_NTT_WAVES", "DISTAFF_NTT_ORDER", "DISTAFF_NTT_DIF"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert r.returncode == 0 and b"ok" in r.stdout, r.stdout.decode()[-3000:]
