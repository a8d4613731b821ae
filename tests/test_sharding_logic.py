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
