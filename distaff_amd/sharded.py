"""Coset-sharded STARK proving on several GPUs: one process (or, for tests, one thread) per GPU, SPMD.

Rank g of G owns the cosets [g*B/G, (g+1)*B/G) of every low-degree extension.  Between the phases of
``stark::prove`` (/root/reference/src/stark/prover.rs:17-168) the ranks exchange
  * the boundary nodes of each Merkle tree (the level at which a node's leaves stop being rank-local) -- all-gather,
  * the combined constraint evaluations before the cross-coset inverse transform -- all-gather,
  * the evaluations of the first small FRI layer (<= 2^17 elements), after which every rank finishes the FRI commit phase on its
    own (replicated tail) -- all-gather,
  * the handful of leaves / nodes / rows that the query openings need -- small object gathers,
and every rank derives the same Fiat-Shamir challenges from the same roots.  There is no all-reduce.

The pure index logic (batch-opening plan, ownership of tree nodes, wire format) lives here and is tested on the CPU;
collectives go through a ``Comm`` object: ``TorchComm`` (torch.distributed: RCCL on GPUs, gloo on CPUs) or ``LocalComm``
(threads in one process, used to exercise the sharded kernels on a single GPU).
"""
import struct
import threading
import time

import numpy as np

from . import lib as L

SH_TRACE_TREE, SH_CONSTRAINT_TREE, SH_FRI_TREE, SH_CEVAL, SH_FRI_LAST, SH_FRI_SEND_CAP = 0, 1, 2, 3, 4, 5
RD_TRACE_LEAF, RD_TRACE_NODE, RD_TRACE_UPPER, RD_CEVAL, RD_C_NODE, RD_C_UPPER, RD_FRI_E, RD_FRI_LEAF, RD_FRI_NODE, RD_FRI_UPPER, RD_LDE_ROW = range(11)


# ---- pure host logic ---------------------------------------------------------------------------------------------------------
def plan_batch(indexes, num_leaves):
    """Which leaves / nodes MerkleTree::prove_batch (src/crypto/merkle.rs:64-124) puts into a batch proof.
    Returns (values: leaf index per requested position, nodes: per normalised pair a list of (is_leaf, heap_or_leaf_index), depth)."""
    index_map = {idx: i for i, idx in enumerate(indexes)}
    assert len(index_map) == len(indexes), "repeating indexes detected"
    norm = sorted({i - (i & 1) for i in indexes})
    nodes, nxt = [], []
    for index in norm:
        has1, has2 = index in index_map, (index + 1) in index_map
        if has1 and has2:
            nodes.append([])
        elif has1:
            nodes.append([(True, index + 1)])
        else:
            nodes.append([(True, index)])
        nxt.append((index + num_leaves) >> 1)
    depth = num_leaves.bit_length() - 1
    for _ in range(1, depth):
        cur, nxt = nxt, []
        i = 0
        while i < len(cur):
            sib = cur[i] ^ 1
            if i + 1 < len(cur) and cur[i + 1] == sib:
                i += 1
            else:
                nodes[i].append((False, sib))
            nxt.append(sib >> 1)
            i += 1
    return list(indexes), nodes, depth


def augmented_positions(positions, column_length):          # src/stark/fri/utils.rs:4
    row_length = column_length // 4
    out = []
    for p in positions:
        ap = p % row_length
        if ap not in out:
            out.append(ap)
    return out


def constraint_positions(positions):                         # src/stark/utils/mod.rs:46
    out = []
    for p in positions:
        if p // 2 not in out:
            out.append(p // 2)
    return out


class TreeGeometry:
    """A Merkle tree over L = Bt*K natural-order leaves (leaf = Bt*k + j') whose leaf columns j' are split evenly over G ranks.
    Rank g keeps leaf (k, j') at local index k*Bct + (j' - g*Bct) and the nodes of the lowest log2(Bct) levels in a local heap;
    levels with 2^l >= Bct are replicated ("upper" heap, same indices as the global heap)."""

    def __init__(self, num_leaves, Bt, G):
        self.L, self.Bt, self.G = num_leaves, Bt, G
        self.Bct = Bt // G
        self.K = num_leaves // Bt

    def leaf(self, i):
        k, j = divmod(i, self.Bt)
        g = j // self.Bct
        return g, k * self.Bct + (j - g * self.Bct)

    def node(self, heap_index):
        """(rank or None when replicated, local-or-upper heap index)"""
        level_count = 1 << (heap_index.bit_length() - 1)      # number of nodes on this level
        t = heap_index - level_count
        span = self.L // level_count                          # leaves covered by the node
        if span >= self.Bct:
            return None, heap_index
        g, local_leaf = self.leaf(t * span)
        local_count = self.L // (self.G * span)
        return g, local_count + local_leaf // span


class Writer:
    """bincode's default encoding as used for StarkProof (src/main.rs:44)."""

    def __init__(self):
        self.parts = []

    def u8(self, v): self.parts.append(struct.pack("<B", v))
    def u32(self, v): self.parts.append(struct.pack("<I", v))
    def u64(self, v): self.parts.append(struct.pack("<Q", v))
    def raw(self, b): self.parts.append(bytes(b))
    def bytes(self): return b"".join(self.parts)


# ---- communicators -----------------------------------------------------------------------------------------------------------------
class LocalComm:
    """All 'ranks' are threads of one process (single-GPU exercise of the sharded path, CPU tests of the logic)."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world
        self.device_path = False

    @classmethod
    def create(cls, world):
        shared = cls._Shared(world)
        return [cls(shared, r) for r in range(world)]

    def all_gather_object(self, obj):
        s = self.shared
        s.slots[self.rank] = obj
        s.barrier.wait()
        out = list(s.slots)
        s.barrier.wait()
        return out

    def all_gather(self, arr):
        return np.concatenate([np.asarray(a).reshape(-1) for a in self.all_gather_object(np.asarray(arr).copy())])

    def barrier(self):
        self.shared.barrier.wait()


class TorchComm:
    """torch.distributed communicator.  With the nccl (= RCCL) backend the payload travels GPU-to-GPU over xGMI; shards move
    between libdistaff_hip.so and the torch tensors either directly on the device (``device_path``) or staged through the host."""

    def __init__(self, dist, device=None, device_path=False):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device                      # torch.device for collectives (None: CPU / gloo)
        self.device_path = bool(device_path and device is not None)

    def all_gather_object(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def all_gather(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
        if self.device is not None:
            t = t.to(self.device)
        out = self.torch.empty(t.numel() * self.world, dtype=self.torch.uint8, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy()

    def synchronize(self):
        """the collective has landed in the tensors before the library (its own stream) reads them"""
        if self.device is not None and self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)

    def all_gather_device(self, send):
        """send: uint8 torch tensor on self.device -> gathered tensor (rank-major)"""
        out = self.torch.empty(send.numel() * self.world, dtype=self.torch.uint8, device=send.device)
        self.dist.all_gather_into_tensor(out, send)
        self.synchronize()
        return out

    def barrier(self):
        self.dist.barrier()


# ---- the SPMD prover ---------------------------------------------------------------------------------------------------------------------
class ShardedProver:
    def __init__(self, ctx, comm, python_openings=False):
        self.ctx, self.comm, self.python_openings = ctx, comm, python_openings
        self.G, self.rank = comm.world, comm.rank
        assert ctx.params.world == self.G and ctx.params.rank == self.rank
        self.B, self.n, self.N, self.W = ctx.B, ctx.n, ctx.N, ctx.W
        self.Bc = self.B // self.G

    def _mark(self, name):
        now = time.perf_counter()
        self.stage_ms[name] = self.stage_ms.get(name, 0.0) + (now - self._t) * 1e3
        self._t = now

    # -- failures: a rank-local exception (out of memory, a refused argument, a staging buffer that is too small) must not leave the other
    #    ranks waiting in the next collective: every phase's outcome is agreed on before anyone goes on
    def _agreed(self, phase, fn, *args):
        err, out = None, None
        try:
            out = fn(*args)
        except L.DistaffError as e:
            err = (e.code, str(e))
        states = self.comm.all_gather_object(err)
        bad = [(g, st) for g, st in enumerate(states) if st is not None]
        if bad:
            g, (code, msg) = bad[0]
            raise L.DistaffError(code, "%s failed on rank %d: %s" % (phase, g, msg))
        return out

    # -- exchanges
    def _exchange(self, what, arg=0):
        ctx, comm = self.ctx, self.comm
        size = ctx.shard_export_size(what, arg)
        if getattr(comm, "device_path", False):
            torch = comm.torch
            send = torch.empty(size, dtype=torch.uint8, device=comm.device)
            ctx.shard_export(what, arg, send.data_ptr(), True)
            gathered = comm.all_gather_device(send)
            root = ctx.shard_import(what, arg, gathered.data_ptr(), True)
            del send, gathered
            return root
        buf = np.empty(size, dtype=np.uint8)
        ctx.shard_export(what, arg, buf.ctypes.data, False)
        gathered = np.ascontiguousarray(comm.all_gather(buf))
        return ctx.shard_import(what, arg, gathered.ctypes.data, False)

    # -- openings
    def _fetch(self, requests, extra=None):
        """requests: list of (owner_rank or None, buffer, arg, local_index); returns the items in order (replicated items from
        rank 0, owned items from their owner) and the list of every rank's `extra` object.  ONE object all-gather."""
        mine = [(i, r) for i, r in enumerate(requests) if r[0] == self.rank or (r[0] is None and self.rank == 0)]
        groups = {}
        for i, (_, buf, arg, idx) in mine:
            groups.setdefault((buf, arg), []).append((i, idx))
        got = {}
        for (buf, arg), items in groups.items():
            data = self.ctx.shard_read(buf, arg, [idx for _, idx in items])
            item = len(data) // len(items)
            for n_, (i, _) in enumerate(items):
                got[i] = data[n_ * item:(n_ + 1) * item]
        merged, extras = {}, []
        for part, ex in self.comm.all_gather_object((got, extra)):
            merged.update(part)
            extras.append(ex)
        return [merged[i] for i in range(len(requests))], extras

    def _tree_requests(self, geom, refs, leaf_buf, node_buf, upper_buf, arg=0):
        reqs = []
        for is_leaf, idx in refs:
            if is_leaf:
                g, li = geom.leaf(idx)
                reqs.append((g, leaf_buf, arg, li))
            else:
                g, hi = geom.node(idx)
                reqs.append((g, node_buf if g is not None else upper_buf, arg, hi))
        return reqs

    def _element_request(self, position, nd):
        """element at natural position B*k + j of a coset-major [Bc][nd] array"""
        k, j = divmod(position, self.B)
        g = j // self.Bc
        return g, (j - g * self.Bc) * nd + k

    def prove(self, inputs, outputs):
        """stark::prove across the ranks; every rank returns the same serialised StarkProof."""
        ctx, comm, G = self.ctx, self.comm, self.G
        B, n, N, W = self.B, self.n, self.N, self.W
        p = ctx.params
        self.stage_ms, self._t = {}, time.perf_counter()
        # steps 1-2
        self._agreed("extension", ctx.shard_commit_trace)
        self._mark("lde_trace_leaves")
        trace_root = self._exchange(SH_TRACE_TREE)
        self._mark("trace_tree_exchange")
        # steps 3-5
        coeffs = L.prng_vector(trace_root, 344)
        bad = ctx.shard_eval_constraints(inputs, outputs, coeffs)
        self._mark("constraint_eval")
        bads = [int(b) for b in comm.all_gather(np.array([bad], dtype=np.int64)).view(np.int64) if b >= 0]
        if bads:
            raise L.DistaffError(L.DST_ERR_AIR, "transition constraints were not satisfied at step %d" % min(bads))
        self._exchange(SH_CEVAL)
        self._mark("ceval_exchange")
        self._agreed("combination", ctx.shard_combine)
        self._mark("combine_constraint_lde")
        constraint_root = self._exchange(SH_CONSTRAINT_TREE)
        self._mark("constraint_tree_exchange")
        # step 6
        draws = L.prng_vector(constraint_root, 516)
        z1, z2 = self._agreed("composition", ctx.compose, draws)
        self._mark("deep_composition")
        # step 7: per exchange two library calls around one all-gather.  Sharded layers: leaves + local levels + export of the
        # boundary nodes | upper tree + draw + fold; then once the evaluations of the first small layer | the rest of the commit phase
        cap = ctx.shard_export_size(SH_FRI_SEND_CAP, 0)
        device_path = getattr(comm, "device_path", False)
        if device_path:
            torch = comm.torch
            send_t = torch.empty(cap, dtype=torch.uint8, device=comm.device)
            recv_t = torch.empty(cap * G, dtype=torch.uint8, device=comm.device)
        else:
            send_h = np.empty(cap, dtype=np.uint8)
        while True:
            if device_path:
                size, more = ctx.shard_fri_begin(send_t.data_ptr(), True, cap)
                out = recv_t[:size * G]
                comm.dist.all_gather_into_tensor(out, send_t[:size])
                comm.synchronize()
                ctx.shard_fri_end(out.data_ptr(), True)
            else:
                size, more = ctx.shard_fri_begin(send_h.ctypes.data, False, cap)
                gathered = np.ascontiguousarray(comm.all_gather(send_h[:size]))
                ctx.shard_fri_end(gathered.ctypes.data, False)
            if not more:
                break
        fri_roots, rep_from = ctx.shard_fri_roots()
        layers = len(fri_roots)
        self._mark("fri")
        # step 8
        seed0 = L.blake3(b"".join(fri_roots))
        seed1, nonce = self._agreed("proof of work", ctx.pow_grind, seed0, p.grinding_factor)
        positions = L.query_positions(seed1, N, B, p.num_queries)
        self._mark("pow_queries")
        if not self.python_openings:
            # step 9 behind the C-ABI: every rank plans the same openings, gathers the items it owns (one device gather), the blobs
            # are all-gathered (padded to the longest; the lengths follow from the plan) and every rank fills the same proof
            blob, lens = self._agreed("openings", ctx.shard_open, positions)
            width = max(max(lens), 1)
            padded = np.zeros(width, dtype=np.uint8)
            padded[:len(blob)] = blob
            gathered = np.asarray(comm.all_gather(padded)).view(np.uint8).reshape(G, width)
            blobs = np.concatenate([gathered[g, :lens[g]] for g in range(G)])
            proof = ctx.shard_assemble(positions, nonce, blobs, lens)
            self._mark("openings")
            return proof
        # step 9 planned in Python (kept as an independent statement of the same plan; tests compare the two)
        op_count, _, stack_depth = ctx.shard_info()
        reqs = []

        def ask(lst):
            o = len(reqs)
            reqs.extend(lst)
            return o, len(lst)

        def pair_requests(u):
            return [(g, RD_CEVAL, 0, li) for g, li in (self._element_request(2 * u, n), self._element_request(2 * u + 1, n))]

        tgeom = TreeGeometry(N, B, G)
        _, tnodes, _ = plan_batch(positions, N)
        h_tnodes = ask(self._tree_requests(tgeom, [r for lst in tnodes for r in lst], RD_TRACE_LEAF, RD_TRACE_NODE, RD_TRACE_UPPER))
        h_rows = ask([((pos % B) // self.Bc, RD_LDE_ROW, 0, pos) for pos in positions])
        cpos = constraint_positions(positions)
        cvals, cnodes, cdepth = plan_batch(cpos, N // 2)
        cgeom = TreeGeometry(N // 2, B // 2, G)
        h_cvals = ask([r for u in cvals for r in pair_requests(u)])
        creqs, spans = [], []
        for is_leaf, idx in (r for lst in cnodes for r in lst):
            if is_leaf:
                creqs += pair_requests(idx); spans.append(2)
            else:
                g, hi = cgeom.node(idx)
                creqs.append((g, RD_C_NODE if g is not None else RD_C_UPPER, 0, hi)); spans.append(1)
        h_cnodes = ask(creqs)
        fri_plan, pos, size = [], list(positions), N
        for dd in range(layers - 1):
            R, nd = size // 4, size // B
            pos = augmented_positions(pos, size)
            fvals, fnodes, fdepth = plan_batch(pos, R)
            if dd >= rep_from:          # replicated tail: natural order and full heaps on every rank, rank 0 serves
                h_vals = ask([(None, RD_FRI_E, dd, r + s * R) for r in pos for s in range(4)])
                fgeom = TreeGeometry(R, B, 1)
            else:
                h_vals = ask([(g, RD_FRI_E, dd, li) for r in pos for s in range(4) for g, li in (self._element_request(r + s * R, nd),)])
                fgeom = TreeGeometry(R, B, G)
            h_nodes = ask(self._tree_requests(fgeom, [r for lst in fnodes for r in lst], RD_FRI_LEAF, RD_FRI_NODE, RD_FRI_UPPER, dd))
            fri_plan.append((len(pos), h_vals, fnodes, h_nodes, fdepth))
            size //= 4
        last = np.empty(ctx.shard_export_size(SH_FRI_LAST, 0), dtype=np.uint8)     # the remainder (replicated, natural order)
        ctx.shard_export(SH_FRI_LAST, 0, last.ctypes.data, False)
        got, _ = self._fetch(reqs)

        def take(h):
            return got[h[0]:h[0] + h[1]]

        w = Writer()
        w.raw(trace_root)
        w.u8(N.bit_length() - 1); w.u8(p.ctx_depth); w.u8(p.loop_depth); w.u8(stack_depth); w.u32(op_count)
        self._write_nodes(w, tnodes, take(h_tnodes))
        w.u64(len(positions))
        for r in take(h_rows):
            w.u64(W); w.raw(r)
        w.raw(constraint_root)
        w.u64(len(cvals)); w.raw(b"".join(take(h_cvals)))
        part, items, o = take(h_cnodes), [], 0
        for sp in spans:
            items.append(b"".join(part[o:o + sp])); o += sp
        self._write_nodes(w, cnodes, items)
        w.u8(cdepth)
        w.u64(W); w.raw(z1.tobytes())
        w.u64(W); w.raw(z2.tobytes())
        # FRI proof
        w.u64(layers - 1)
        for dd, (count, h_vals, fnodes, h_nodes, fdepth) in enumerate(fri_plan):
            w.raw(fri_roots[dd])
            w.u64(count); w.raw(b"".join(take(h_vals)))
            self._write_nodes(w, fnodes, take(h_nodes))
            w.u8(fdepth)
        w.raw(fri_roots[-1])
        w.u64(size); w.raw(last.tobytes())
        w.u64(nonce)
        w.u8(p.log_blowup); w.u8(p.num_queries); w.u8(p.grinding_factor); w.u8(0)
        self._mark("openings")
        return w.bytes()

    @staticmethod
    def _write_nodes(w, node_lists, items):
        w.u64(len(node_lists))
        o = 0
        for lst in node_lists:
            w.u64(len(lst))
            for _ in lst:
                w.raw(items[o]); o += 1


def prove_local(columns, log_n, width, ctx_depth, loop_depth, inputs, outputs, world, device=0, python_openings=False, **options):
    """Runs the sharded prover with `world` ranks as threads on ONE device (tests / single-GPU validation of the sharded path)."""
    comms = LocalComm.create(world)
    results, errors = [None] * world, [None] * world

    def run(rank):
        try:
            ctx = L.Context(log_n, width, ctx_depth, loop_depth, device=device, rank=rank, world=world, **options)
            ctx.upload(columns)
            results[rank] = ShardedProver(ctx, comms[rank], python_openings).prove(inputs, outputs)
            ctx.close()
        except BaseException as e:      # noqa: BLE001 -- surfaced below; break the barrier so the other ranks do not hang
            errors[rank] = e
            comms[rank].shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    return results
