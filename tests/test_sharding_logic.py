"""CPU tests of the multi-GPU (coset-sharded) host logic: batch-opening plans and wire format against the oracle, ownership of
tree nodes, and the collective layer over torch.distributed (gloo, world_size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_batch_matches_reference_prove_batch(oracle):
    from distaff_amd import sharded
    O = oracle
    rng = np.random.default_rng(5)
    for num_leaves in (8, 64, 1024):
        leaves = [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(num_leaves)]
        nodes = O.merkle_nodes(b"".join(leaves))
        for count in (1, 2, 5, min(50, num_leaves // 2)):
            idx = [int(x) for x in rng.choice(num_leaves, size=count, replace=False)]
            expected = O.merkle_prove_batch(b"".join(leaves), idx)
            values, plan, depth = sharded.plan_batch(idx, num_leaves)
            assert depth == expected["depth"]
            assert [leaves[v] for v in values] == expected["values"]
            got = [[leaves[i] if is_leaf else nodes[32 * i:32 * i + 32] for is_leaf, i in lst] for lst in plan]
            assert got == expected["nodes"]


def test_tree_geometry_partitions_the_tree():
    """every leaf / node has exactly one home: a (rank, local index) for the low levels, the replicated upper heap above."""
    from distaff_amd.sharded import TreeGeometry
    for L, Bt, G in ((1 << 10, 32, 2), (1 << 10, 32, 8), (1 << 9, 16, 4), (1 << 8, 32, 1)):
        geom = TreeGeometry(L, Bt, G)
        seen = {g: set() for g in range(G)}
        for i in range(L):
            g, li = geom.leaf(i)
            assert 0 <= li < L // G and li not in seen[g]
            seen[g].add(li)
            k, j = divmod(i, Bt)
            assert g == j // (Bt // G) and li == k * (Bt // G) + j % (Bt // G)
        local_nodes = {g: set() for g in range(G)}
        for heap in range(1, L):
            g, hi = geom.node(heap)
            level_count = 1 << (heap.bit_length() - 1)
            span = L // level_count
            if span >= Bt // G:
                assert g is None and hi == heap                       # replicated, same heap index
            else:
                assert hi not in local_nodes[g]
                local_nodes[g].add(hi)
        for g in range(G):
            assert len(local_nodes[g]) == (L // Bt) * (Bt // G - 2)           # levels strictly below the boundary level


def test_tree_geometry_single_rank_is_the_plain_heap():
    """One rank: every node's local index is its heap index -- also for trees with fewer leaves than columns (FRI layers of fewer
    rows than the extension factor, e.g. 128 rows at extension 256), where the column count does not divide the leaf count."""
    from distaff_amd.sharded import TreeGeometry
    for L, Bt in ((1 << 8, 32), (1 << 7, 256), (1 << 5, 64), (4, 256)):
        geom = TreeGeometry(L, Bt, 1)
        for i in range(L):
            assert geom.leaf(i) == (0, i)
        for heap in range(1, L):
            g, hi = geom.node(heap)
            assert hi == heap and g in (None, 0)


def test_sharded_tree_equals_full_tree(oracle):
    """Model of the tree exchange on the CPU: local sub-heaps per rank + all-gathered boundary nodes give the reference's root and nodes."""
    from distaff_amd.sharded import TreeGeometry
    O = oracle
    rng = np.random.default_rng(9)
    L, Bt, G = 512, 32, 4
    leaves = [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(L)]
    full = O.merkle_nodes(b"".join(leaves))
    geom = TreeGeometry(L, Bt, G)
    Bct, K = Bt // G, L // Bt
    boundary = []
    local_heaps = []
    for g in range(G):
        loc = [None] * (K * Bct)
        for i in range(L):
            gg, li = geom.leaf(i)
            if gg == g:
                loc[li] = leaves[i]
        heap = {}
        level, count = loc, K * Bct
        while count > K:
            count //= 2
            level = [O.blake3(level[2 * t] + level[2 * t + 1]) for t in range(count)]
            for t in range(count):
                heap[count + t] = level[t]
        local_heaps.append(heap)
        boundary.append(level)                                        # K nodes, one per k
    # all-gather (rank-major) then interleave: node G*k + g
    upper = {K * G + G * k + g: boundary[g][k] for g in range(G) for k in range(K)}
    count = K * G
    while count > 1:
        count //= 2
        for t in range(count):
            upper[count + t] = O.blake3(upper[2 * (count + t)] + upper[2 * (count + t) + 1])
    assert upper[1] == full[32:64]
    for heap in range(1, L):
        g, hi = geom.node(heap)
        expect = full[32 * heap:32 * heap + 32]
        assert (upper[hi] if g is None else local_heaps[g][hi]) == expect


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from distaff_amd.sharded import TorchComm
dist.init_process_group("gloo")
comm = TorchComm(dist)
r = comm.rank
got = comm.all_gather(np.arange(6, dtype=np.uint8) + 10 * r)
assert got.tolist() == list(range(6)) + list(range(10, 16)), got
objs = comm.all_gather_object({"rank": r, "x": [r] * 2})
assert objs == [{"rank": 0, "x": [0, 0]}, {"rank": 1, "x": [1, 1]}]
comm.barrier()
dist.destroy_process_group()
print("ok", r)
'''


def test_torch_comm_gloo_world2(tmp_path):
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_library_exports_every_declared_symbol():
    """the C-ABI shared library loads without a GPU and exports everything include/distaff_hip.h declares"""
    import re
    import distaff_amd as D
    lib = D.load()
    header = open(os.path.join(ROOT, "include", "distaff_hip.h")).read()
    declared = set(re.findall(r"\b(dst_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(D.EXPORTS), declared ^ set(D.EXPORTS)


# ---- the whole sharded orchestration on the CPU: ShardedProver over a mock rank-local context (tests/mock_shard.py) -------------
def _mock_world(oracle, world, trace, prover, comms, **options):
    import threading
    from mock_shard import MockShardContext
    from distaff_amd.sharded import ShardedProver
    results, errors = [None] * world, [None] * world

    def run(rank):
        try:
            ctx = MockShardContext(oracle, prover, trace, rank, world, **options)
            results[rank] = ShardedProver(ctx, comms[rank], python_openings=True).prove(trace.public_inputs, prover.outputs)
        except BaseException as e:      # noqa: BLE001
            errors[rank] = e
            comms[rank].shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    if first is not None:
        raise first
    return results


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sharded_orchestration_reproduces_the_oracle_proof(oracle, world):
    """every rank of ShardedProver.prove (exchanges, Fiat-Shamir, opening plan, ownership, wire format) assembles the oracle's
    proof byte for byte when the rank-local buffers hold the oracle's values in the sharded layouts"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from distaff_amd.sharded import LocalComm
    O = oracle
    t = O.fibonacci_trace(128)
    p = O.Prover.from_trace(t, 1, grinding=8)
    proof = p.prove()
    for replicate_log in (17, 10, 8):        # FRI: everything in the replicated tail | one | two sharded layers before it
        results = _mock_world(O, world, t, p, LocalComm.create(world), grinding=8, replicate_log=replicate_log)
        assert all(r == proof for r in results), replicate_log
    assert O.verify(results[-1], t.program_hash, t.public_inputs, p.outputs) == (True, "")


def test_sharded_orchestration_other_options_and_program(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from distaff_amd.sharded import LocalComm
    O = oracle
    t = O.fibonacci_trace(256)
    p = O.Prover.from_trace(t, 1, ext=16, num_queries=100, grinding=10)          # config 5's options at a small size
    proof = p.prove()
    for world in (2, 4, 8):                  # 8 ranks at blowup 16: two cosets per rank, the constraint tree's boundary is its leaf level
        assert all(r == proof for r in _mock_world(O, world, t, p, LocalComm.create(world), log_blowup=4, num_queries=100, grinding=10, replicate_log=9))
    t = O.Trace("begin add block push.5 mul push.7 end end", [1, 2])             # context register, two outputs
    p = O.Prover.from_trace(t, 2, grinding=8)
    proof = p.prove()
    assert all(r == proof for r in _mock_world(O, 2, t, p, LocalComm.create(2), grinding=8))


def test_mock_detects_a_gather_in_the_wrong_rank_order(oracle):
    """the mock is a real check of the collective layer: pieces delivered in another order are refused"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from distaff_amd.sharded import LocalComm
    O = oracle
    t = O.fibonacci_trace(128)
    p = O.Prover.from_trace(t, 1, grinding=8)
    p.prove()

    class Reversed(LocalComm):
        def all_gather_object(self, obj):
            return super().all_gather_object(obj)[::-1]

    shared = LocalComm._Shared(2)
    with pytest.raises(AssertionError, match="rank order"):
        _mock_world(O, 2, t, p, [Reversed(shared, r) for r in range(2)], grinding=8)


MOCK_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch.distributed as dist
import oracle as O
from mock_shard import MockShardContext
from distaff_amd.sharded import ShardedProver, TorchComm
dist.init_process_group("gloo")
comm = TorchComm(dist)
t = O.fibonacci_trace(128)
p = O.Prover.from_trace(t, 1, grinding=8)
proof = p.prove()
ctx = MockShardContext(O, p, t, comm.rank, comm.world, grinding=8, replicate_log=8)      # two sharded FRI layers, then the replicated tail
prover = ShardedProver(ctx, comm, python_openings=True)
got = prover.prove(t.public_inputs, p.outputs)
assert got == proof, "rank %%d: proof differs from the oracle's" %% comm.rank
assert set(prover.stage_ms) >= {"trace_tree_exchange", "ceval_exchange", "fri", "openings"}
comm.barrier()
dist.destroy_process_group()
print("ok", comm.rank, len(got))
'''


def test_sharded_orchestration_gloo_world2(oracle, tmp_path):
    """two processes over torch.distributed (gloo): the N > 1 path of bench.py / INTEGRATION.md minus the device"""
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "mock_worker.py"
    script.write_text(MOCK_WORKER % {"root": ROOT})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)
