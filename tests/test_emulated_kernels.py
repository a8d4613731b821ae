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
    "test_other_program_shapes",
    "test_program_shapes_with_stack_depth_5_to_8",
    "test_blowup_16_and_64",
    "test_blowup_128_and_256[128-7]",
    "test_config2_random_columns_lde_and_merkle",
    "test_invalid_trace_reports_air_error",
    "test_wide_rows_two_chunk_leaves",
    "test_sharded_prover_equals_single_gpu[2]",
    "test_sharded_prover_equals_single_gpu[8]",
    "test_sharded_prover_reports_invalid_trace",
    "test_sharded_fri_protocol_state_errors",
    "test_deep_stacks_and_nested_blocks[]",
    "test_deep_stacks_and_nested_blocks[generic]",
    "test_general_constraint_instances_on_the_fibonacci_trace[small]",
    "test_general_constraint_instances_on_the_fibonacci_trace[deep]",
    "test_general_constraint_instances_on_the_fibonacci_trace[generic]",
    "test_lde_every_tile_length[reg-13-5]",
    "test_lde_every_tile_length[lds-13-5]",
]


@pytest.fixture(scope="module")
def emulated_library():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j8"], stdout=subprocess.DEVNULL)
    assert os.path.exists(EMU_LIB)
    return EMU_LIB


def _run(emulated_library, args, timeout=900):
    env = dict(os.environ, DISTAFF_HIP_LIB=emulated_library, DISTAFF_HIP_RUNTIME="none")
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu"] + args,
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
