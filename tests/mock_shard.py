"""A stand-in for the rank-local library context, for CPU tests of the multi-GPU orchestration (distaff_amd/sharded.py).

TEST INFRASTRUCTURE ONLY.  `MockShardContext` answers the shard calls of `distaff_amd.lib.Context` (include/distaff_hip.h, the
dst_shard_* group) from the intermediates of the CPU oracle's prover, sliced into the rank-local layouts DESIGN.md section 6
describes: rank g of G owns the cosets [g*B/G, (g+1)*B/G) of every extension, keeps evaluations coset-major, tree leaves at
k*Bct + j_local, the lowest tree levels in a local heap and the replicated levels in an "upper" heap.  Nothing here is imported
by the package; it lets `ShardedProver.prove` run over torch.distributed (gloo) with world_size 2 where there is no GPU, and it
checks on the way that
  * every rank derives the oracle's Fiat-Shamir draws (constraint coefficients, composition coefficients, PoW seed),
  * every all-gather delivers the ranks' pieces in rank order and of the advertised size,
so that the assembled proof can be compared byte for byte with the oracle's.
"""
import ctypes
import types

import numpy as np

from distaff_amd import sharded as S


def _heap_bytes(raw):
    return [raw[32 * i:32 * i + 32] for i in range(len(raw) // 32)]


class _Tree:
    """One Merkle tree of the proof in the sharded geometry (leaves Bt*k + j', columns j' split over the ranks)."""

    def __init__(self, heap, leaves, Bt, G):
        self.heap, self.leaves = heap, leaves              # heap[1] = root, len(heap) = number of leaves
        self.L, self.Bt, self.G = len(heap), Bt, G
        self.Bct, self.K = Bt // G, len(heap) // Bt

    def boundary(self, g):
        """the nodes whose leaves are exactly rank g's columns of one row k: level with K*G nodes, offset G*k + g"""
        level = self.K * self.G
        if level == self.L:              # one leaf column per rank (constraint tree, two cosets per rank): the leaves are the boundary
            return b"".join(self.leaves[self.G * k + g] for k in range(self.K))
        return b"".join(self.heap[level + self.G * k + g] for k in range(self.K))

    def global_leaf(self, g, local):
        k, jl = divmod(local, self.Bct)
        return self.Bt * k + g * self.Bct + jl

    def local_node(self, g, hi):
        """node `hi` of rank g's local heap (over its K*Bct leaves in local order) -> the global heap's node"""
        level = 1 << (hi.bit_length() - 1)
        span = (self.L // self.G) // level                  # leaves below the node
        assert span < self.Bct, "that level is replicated, not local"
        first = self.global_leaf(g, (hi - level) * span)
        return self.heap[self.L // span + first // span]


class MockShardContext:
    def __init__(self, oracle, prover, trace, rank, world, log_blowup=5, num_queries=50, grinding=20, replicate_log=17):
        O, p = oracle, prover
        self.O, self.p = O, p
        self.B, self.n, self.W = 1 << log_blowup, trace.length, trace.columns.shape[0]
        self.N = self.B * self.n
        self.G, self.g, self.Bc = world, rank, self.B // world
        self.params = types.SimpleNamespace(world=world, rank=rank, grinding_factor=grinding, log_blowup=log_blowup,
                                            num_queries=num_queries, ctx_depth=trace.ctx_depth, loop_depth=trace.loop_depth)
        self.op_count = int(trace.columns[0, self.n - 1, 0])
        self.stack_depth = self.W - 15 - trace.ctx_depth - trace.loop_depth
        B, G = self.B, world
        self.registers = p.get("registers")                                      # [W][N][2]
        self.cevals = p.get("constraint_evaluations")                            # [N][2]
        self.t_evals = p.get("t_evaluations")                                    # [8n][2]
        self.trace_tree = _Tree(_heap_bytes(p.get_bytes("trace_nodes")), _heap_bytes(p.get_bytes("trace_leaves")), B, G)
        pairs = self.cevals.reshape(self.N // 2, 4).tobytes()
        cleaves = [pairs[32 * i:32 * i + 32] for i in range(self.N // 2)]       # the leaves are the raw pairs (prover.rs:96)
        self.c_tree = _Tree(_heap_bytes(p.get_bytes("constraint_nodes")), cleaves, B // 2, G)
        self.layers = p.get_u64("fri_layers")[0]
        self.fri_rows, self.fri_tree = [], []
        for d in range(self.layers):
            rows = p.get("fri_values", d)                                        # [R][4][2]
            raw = rows.tobytes()
            leaves = [O.blake3(raw[64 * i:64 * i + 64]) for i in range(rows.shape[0])]
            self.fri_rows.append(rows)
            self.fri_tree.append(_Tree(_heap_bytes(p.get_bytes("fri_nodes", d)), leaves, B, G))
        roots = p.get_bytes("roots")
        self.trace_root, self.constraint_root = roots[:32], roots[32:]
        fr = p.get_bytes("fri_roots")
        self.fri_roots = [fr[32 * d:32 * d + 32] for d in range(self.layers)]
        # layers >= rep_from are replicated: natural order, full heaps on every rank (shard.hip fri_replicated_from)
        sizes = [4 * rows.shape[0] for rows in self.fri_rows]
        self.rep_from = next(d for d in range(self.layers) if sizes[d] <= (1 << replicate_log) or sizes[d] // B < 4 or d == self.layers - 1)
        for d in range(self.rep_from, self.layers):
            self.fri_tree[d] = _Tree(self.fri_tree[d].heap, self.fri_tree[d].leaves, B, 1)
        self.fri_d = 0
        self.tail_pending = False
        self.calls = []

    # -- layouts
    def _fri_element(self, d, position):
        rows = self.fri_rows[d]
        R = rows.shape[0]
        return rows[position % R, position // R].tobytes()

    def _export(self, what, arg, g):
        B, n, Bc = self.B, self.n, self.Bc
        if what == S.SH_TRACE_TREE:
            return self.trace_tree.boundary(g)
        if what == S.SH_CONSTRAINT_TREE:
            return self.c_tree.boundary(g)
        if what == S.SH_FRI_TREE:
            return self.fri_tree[arg].boundary(g)
        if what == S.SH_CEVAL:                                                   # this rank's cosets of the 8n domain, [Q][n]
            Q = 8 // self.G
            q = np.arange(g * Q, (g + 1) * Q)[:, None]
            k = np.arange(n)[None, :]
            return np.ascontiguousarray(self.t_evals[8 * k + q]).tobytes()
        if what == S.SH_FRI_LAST:                                                # the whole remainder in natural order (replicated)
            d = self.layers - 1
            return b"".join(self._fri_element(d, i) for i in range(4 * self.fri_rows[d].shape[0]))
        if what == "fri_tail":                                                   # rank g's cosets of the first replicated layer, [Bc][nd]
            d = self.rep_from
            nd = self.fri_rows[d].shape[0] * 4 // B
            return b"".join(self._fri_element(d, B * k + g * Bc + jl) for jl in range(Bc) for k in range(nd))
        raise AssertionError("unknown export %r" % what)

    # -- the calls ShardedProver makes
    def shard_commit_trace(self):
        self.calls.append("commit_trace")

    def shard_export_size(self, what, arg=0):
        if what == S.SH_FRI_SEND_CAP:
            return max([len(self._export("fri_tail", 0, self.g))] + [len(self._export(S.SH_FRI_TREE, d, self.g)) for d in range(self.rep_from)])
        return len(self._export(what, arg, self.g))

    def shard_export(self, what, arg, dst_ptr, is_device):
        assert not is_device
        data = self._export(what, arg, self.g)
        ctypes.memmove(dst_ptr, data, len(data))

    def shard_import(self, what, arg, src_ptr, is_device):
        assert not is_device
        expect = b"".join(self._export(what, arg, g) for g in range(self.G))
        assert ctypes.string_at(src_ptr, len(expect)) == expect, "all-gather of %d/%d is not the ranks' pieces in rank order" % (what, arg)
        self.calls.append(("import", what, arg))
        if what == S.SH_FRI_TREE:
            return self.fri_roots[arg]
        return {S.SH_TRACE_TREE: self.trace_root, S.SH_CONSTRAINT_TREE: self.constraint_root, S.SH_CEVAL: bytes(32)}[what]

    def shard_eval_constraints(self, inputs, outputs, coeffs):
        assert np.array_equal(np.asarray(coeffs, dtype=np.uint64).reshape(-1, 2), self.p.get("constraint_draws")), "constraint draws"
        return -1

    def shard_combine(self):
        self.calls.append("combine")

    def compose(self, draws):
        assert np.array_equal(np.asarray(draws, dtype=np.uint64).reshape(-1, 2), self.p.get("deep_draws")), "composition draws"
        return self.p.get("trace_at_z1").copy(), self.p.get("trace_at_z2").copy()

    def shard_fri_begin(self, send_ptr, is_device, cap):
        assert not is_device and not self.tail_pending
        if self.fri_d == self.rep_from:
            data, more = self._export("fri_tail", 0, self.g), False
            self.tail_pending = True
        else:
            assert self.fri_d < self.rep_from
            data, more = self._export(S.SH_FRI_TREE, self.fri_d, self.g), True
        assert len(data) <= cap, "send buffer smaller than the item (SH_FRI_SEND_CAP)"
        ctypes.memmove(send_ptr, data, len(data))
        return len(data), more

    def shard_fri_end(self, gathered_ptr, is_device):
        assert not is_device
        if self.tail_pending:
            expect = b"".join(self._export("fri_tail", 0, g) for g in range(self.G))
            assert ctypes.string_at(gathered_ptr, len(expect)) == expect, "all-gather of the first replicated FRI layer is not the ranks' pieces in rank order"
            self.tail_pending = False
            root = self.fri_roots[self.rep_from]
            self.fri_d = self.layers
            return root
        root = self.shard_import(S.SH_FRI_TREE, self.fri_d, gathered_ptr, is_device)
        self.fri_d += 1
        return root

    def shard_fri_roots(self):
        assert self.fri_d == self.layers, "FRI commit phase not finished"
        return list(self.fri_roots), self.rep_from

    def pow_grind(self, seed, grinding):
        seeds = self.p.get_bytes("query_seeds")
        assert bytes(seed) == seeds[:32], "proof-of-work seed"
        assert grinding == self.params.grinding_factor
        return seeds[32:], self.p.get_u64("pow_nonce")[0]

    def shard_info(self):
        return self.op_count, self.layers, self.stack_depth

    def shard_read(self, buffer, arg, indices):
        B, Bc, n, g = self.B, self.Bc, self.n, self.g
        out = []
        for i in (int(v) for v in indices):
            if buffer == S.RD_TRACE_LEAF:
                out.append(self.trace_tree.leaves[self.trace_tree.global_leaf(g, i)])
            elif buffer == S.RD_TRACE_NODE:
                out.append(self.trace_tree.local_node(g, i))
            elif buffer == S.RD_TRACE_UPPER:
                out.append(self.trace_tree.heap[i])
            elif buffer == S.RD_CEVAL:                                           # coset-major [Bc][n]
                jl, k = divmod(i, n)
                out.append(self.cevals[B * k + g * Bc + jl].tobytes())
            elif buffer == S.RD_C_NODE:
                out.append(self.c_tree.local_node(g, i))
            elif buffer == S.RD_C_UPPER:
                out.append(self.c_tree.heap[i])
            elif buffer == S.RD_FRI_E and arg >= self.rep_from:                   # replicated layer: natural order
                out.append(self._fri_element(arg, i))
            elif buffer == S.RD_FRI_E:                                           # coset-major [Bc][nd]
                nd = self.fri_rows[arg].shape[0] * 4 // B
                jl, k = divmod(i, nd)
                out.append(self._fri_element(arg, B * k + g * Bc + jl))
            elif buffer == S.RD_FRI_LEAF:
                t = self.fri_tree[arg]
                out.append(t.leaves[t.global_leaf(0 if arg >= self.rep_from else g, i)])
            elif buffer == S.RD_FRI_NODE:
                out.append(self.fri_tree[arg].local_node(0 if arg >= self.rep_from else g, i))
            elif buffer == S.RD_FRI_UPPER:
                out.append(self.fri_tree[arg].heap[i])
            elif buffer == S.RD_LDE_ROW:                                         # natural position owned by this rank
                assert (i % B) // Bc == g, "row %d is not this rank's" % i
                out.append(np.ascontiguousarray(self.registers[:, i]).tobytes())
            else:
                raise AssertionError("unknown buffer %r" % buffer)
        return b"".join(out)
