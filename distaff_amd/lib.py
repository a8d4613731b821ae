"""ctypes binding of libdistaff_hip.so (the product).  Nothing here falls back to a CPU implementation: if the shared
library is missing, or a call needs a GPU that is not there, an exception is raised.

Field elements cross the C-ABI as 16 little-endian bytes (the memory image of the reference's ``u128``); on the Python side
bulk data are numpy ``uint64`` arrays whose last axis has length 2 (low word, high word).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "libdistaff_hip.so")             # the product: what a host binds, what bench.py measures, what smoke() runs
HOOKS_LIB = os.path.join(_HERE, "libdistaff_hip_hooks.so")          # the same sources with -DDISTAFF_TEST_HOOKS (tests, bench.py's calibration kernels)
# DISTAFF_HIP_LIB names any other build (the tests' CPU emulation); DISTAFF_TEST_HOOKS=1 selects the test / bench build (tests/conftest.py)
LIB_PATH = os.environ.get("DISTAFF_HIP_LIB") or (HOOKS_LIB if os.environ.get("DISTAFF_TEST_HOOKS") == "1" else PRODUCT_LIB)


def use_test_hooks():
    """Bind the test / bench build (libdistaff_hip_hooks.so) instead of the product library -- before the first load(); child processes that
    import this package inherit the choice through DISTAFF_TEST_HOOKS=1 (bench.py removes the variable: it measures the product)."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_test_hooks() must be called before the library is loaded")
    os.environ["DISTAFF_TEST_HOOKS"] = "1"
    if not os.environ.get("DISTAFF_HIP_LIB"):
        LIB_PATH = HOOKS_LIB


def use_product():
    """Bind the product library whatever the environment said at import (an inherited DISTAFF_TEST_HOOKS=1) -- before the first load().
    An explicit DISTAFF_HIP_LIB still wins: it names a library, not a build flavour."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_product() must be called before the library is loaded")
    os.environ.pop("DISTAFF_TEST_HOOKS", None)
    if not os.environ.get("DISTAFF_HIP_LIB"):
        LIB_PATH = PRODUCT_LIB


def library_path():
    return LIB_PATH

DST_OK, DST_ERR_ARG, DST_ERR_HIP, DST_ERR_AIR, DST_ERR_STATE, DST_ERR_COMM = 0, -1, -2, -3, -4, -5

# ids of dst_read_buffer (distaff_amd/csrc/ctx.h)
BUF = {"polys": 0, "lde": 1, "trace_leaves": 2, "trace_nodes": 3, "ceval_i": 4, "ceval_f": 5, "ceval_t": 6, "cpoly": 7, "cevals": 8,
       "cnodes": 9, "comp_poly": 10, "comp_evals": 11, "fri_evals": 12, "fri_nodes": 13, "fri_leaves": 14}

EXPORTS = ["dst_ctx_create", "dst_ctx_destroy", "dst_last_error", "dst_phase_ms", "dst_trace_upload", "dst_trace_upload_contiguous", "dst_trace_upload_owned",
           "dst_commit_trace", "dst_eval_constraints", "dst_compose", "dst_fri_commit_layer", "dst_fri_fold", "dst_pow_grind",
           "dst_build_proof", "dst_prove", "dst_prng_vector", "dst_query_positions", "dst_blake3", "dst_fibonacci_trace",
           "dst_read_buffer", "dst_bench_mulmod", "dst_bench_mad", "dst_bench_code", "dst_trace_upload_async", "dst_pinned_alloc", "dst_pinned_free", "dst_set_profiling", "dst_kernel_stats", "dst_field_op",
           "dst_shard_commit_trace", "dst_shard_eval_constraints", "dst_shard_combine", "dst_shard_fri_layer", "dst_shard_fri_fold",
           "dst_shard_export_size", "dst_shard_export", "dst_shard_import", "dst_shard_read", "dst_shard_fri_begin", "dst_shard_fri_end", "dst_shard_fri_roots", "dst_shard_open", "dst_shard_assemble", "dst_shard_info",
           "dst_comm_unique_id", "dst_comm_init", "dst_comm_init_local", "dst_comm_init_callbacks", "dst_comm_destroy", "dst_comm_last_error", "dst_comm_copy", "dst_prove_sharded", "dst_prove_sharded_local", "dst_shard_stage_ms",
           "dst_comm_describe", "dst_comm_trace", "dst_test_hooks", "dst_comm_set_timeout", "dst_comm_abort", "dst_shard_exchange_ms", "dst_bench_clock"]


class DistaffError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libdistaff_hip error %d: %s" % (code, message))
        self.code = code


class Params(ctypes.Structure):
    _fields_ = [("log_trace_length", ctypes.c_uint32), ("log_blowup", ctypes.c_uint32), ("width", ctypes.c_uint32),
                ("ctx_depth", ctypes.c_uint32), ("loop_depth", ctypes.c_uint32), ("num_queries", ctypes.c_uint32),
                ("grinding_factor", ctypes.c_uint32), ("device", ctypes.c_int32), ("rank", ctypes.c_uint32), ("world", ctypes.c_uint32)]


class Public(ctypes.Structure):
    _fields_ = [("num_inputs", ctypes.c_uint32), ("num_outputs", ctypes.c_uint32),
                ("inputs", (ctypes.c_uint8 * 16) * 8), ("outputs", (ctypes.c_uint8 * 16) * 8)]


class CommInfo(ctypes.Structure):
    _fields_ = [("transport", ctypes.c_uint32), ("rank", ctypes.c_uint32), ("world", ctypes.c_uint32), ("device", ctypes.c_int32),
                ("rccl_ranks", ctypes.c_uint32), ("rccl_rank", ctypes.c_uint32), ("rccl_version", ctypes.c_uint32),
                ("peers_other_device", ctypes.c_uint32), ("peers_enabled", ctypes.c_uint32)]


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (soname libamdhip64.so.7) next to libtorch_hip.so;
    libdistaff_hip.so needs `libamdhip64.so.7`.  If the ROCm copy under /opt/rocm is loaded first, torch later brings its own copy
    in as a SECOND runtime, which then finds no devices; loaded the other way round, the dynamic linker resolves our NEEDED entry to
    torch's copy by soname and both share one runtime (and device pointers can be exchanged directly, which the sharded prover
    relies on).  So when torch is installed, its copy is loaded first, without importing torch.  Without torch nothing happens and
    the library binds to /opt/rocm as usual (that is the case for a Rust host binding the C-ABI)."""
    if os.environ.get("DISTAFF_HIP_RUNTIME", "torch") != "torch":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        pass


def _open(path):
    if not os.path.exists(path):
        raise ImportError("%s is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; g.build()'), there is no CPU fallback" % path)
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    lib.dst_last_error.restype = ctypes.c_char_p
    lib.dst_last_error.argtypes = [ctypes.c_void_p]
    return lib


def load():
    """Loads the shared library (raises if it has not been built: run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


_hooks_lib = None


def load_hooks():
    """The test / bench build as a SECOND handle beside the product library (bench.py: the calibration kernels live only there)."""
    global _hooks_lib
    if _hooks_lib is None:
        bound = load()
        has = getattr(bound, "dst_test_hooks", None)
        _hooks_lib = bound if (has is not None and has() == 1) else _open(HOOKS_LIB)       # the bound library is itself a test build (tests, the CPU emulation)
    return _hooks_lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def int_to_bytes(v):
    return int(v).to_bytes(16, "little")


def ints_to_arr(values):
    out = np.empty((len(values), 2), dtype=np.uint64)
    for i, v in enumerate(values):
        out[i, 0] = int(v) & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = int(v) >> 64
    return out


def arr_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 2)
    return [int(lo) | (int(hi) << 64) for lo, hi in a]


def make_public(inputs, outputs):
    p = Public()
    p.num_inputs, p.num_outputs = len(inputs), len(outputs)
    for i, v in enumerate(inputs):
        p.inputs[i][:] = list(int_to_bytes(v))
    for i, v in enumerate(outputs):
        p.outputs[i][:] = list(int_to_bytes(v))
    return p


def prng_vector(seed, count):
    """field::prng_vector (src/math/field.rs:271) -- host side, StdRng(ChaCha20) + Uniform restatement inside the library."""
    out = np.zeros((count, 2), dtype=np.uint64)
    load().dst_prng_vector(bytes(seed), ctypes.c_uint32(count), _ptr(out))
    return out


def query_positions(seed, domain_size, blowup, num_queries):
    out = np.zeros(num_queries, dtype=np.uint64)
    k = load().dst_query_positions(bytes(seed), ctypes.c_uint64(domain_size), ctypes.c_uint32(blowup), ctypes.c_uint32(num_queries), _ptr(out))
    if k < 0:
        raise DistaffError(k, "could not generate enough query positions")
    return [int(x) for x in out[:k]]


def blake3(data):
    out = ctypes.create_string_buffer(32)
    load().dst_blake3(bytes(data), ctypes.c_size_t(len(data)), out)
    return out.raw


def fibonacci_trace(log_n):
    """Fibonacci example trace filling 2^log_n rows: returns (columns [20, n, 2] uint64, program_hash bytes, result int)."""
    n = 1 << log_n
    cols = np.zeros((20, n, 2), dtype=np.uint64)
    ph = ctypes.create_string_buffer(32)
    res = ctypes.create_string_buffer(16)
    r = load().dst_fibonacci_trace(ctypes.c_uint32(log_n), _ptr(cols), ph, res)
    if r != DST_OK:
        raise DistaffError(r, "dst_fibonacci_trace failed")
    return cols, ph.raw, int.from_bytes(res.raw, "little")


class Comm:
    """One rank's communicator handle (``dst_comm``) for ``Context.prove_sharded``: RCCL over xGMI, the unique id created by
    ``Comm.unique_id()`` on rank 0 and handed to the other ranks by the host's own channel."""

    def __init__(self, handle):
        self.lib, self._h = load(), handle

    def describe(self):
        """dst_comm_describe: transport, rank / world, and what the transport says about itself -- for RCCL the number of ranks the live
        communicator connected (ncclCommCount), this rank's index and device; for the in-process transport the peer-access picture."""
        info = CommInfo()
        if self.lib.dst_comm_describe(self._h, ctypes.byref(info)) != DST_OK:
            raise DistaffError(DST_ERR_ARG, "dst_comm_describe failed")
        d = {f: int(getattr(info, f)) for f, _ in CommInfo._fields_}
        d["transport"] = ("rccl", "local", "callbacks")[d["transport"]]
        return d

    def set_timeout(self, seconds):
        """dst_comm_set_timeout: limit of every host wait behind a collective of this communicator (default 60 s; <= 0: none).  On expiry
        the rank aborts the communicator and prove_sharded raises DistaffError(DST_ERR_COMM)."""
        if self.lib.dst_comm_set_timeout(self._h, ctypes.c_double(seconds)) != DST_OK:
            raise DistaffError(DST_ERR_ARG, "dst_comm_set_timeout failed")

    def abort(self):
        """dst_comm_abort: gives the communicator up from the host's side; every later collective of this handle fails at once"""
        self.lib.dst_comm_abort(self._h)

    def last_error(self):
        self.lib.dst_comm_last_error.restype = ctypes.c_char_p
        self.lib.dst_comm_last_error.argtypes = [ctypes.c_void_p]
        return self.lib.dst_comm_last_error(self._h).decode()

    def trace(self, enable=-1):
        """dst_comm_trace: the collectives this rank has issued so far as [(kind, bytes per rank, stream index or None)]; enable = 1 starts a
        fresh record, 0 stops, -1 leaves recording as it is."""
        n = ctypes.c_size_t(0)
        self.lib.dst_comm_trace(self._h, ctypes.c_int(-1), None, ctypes.c_size_t(0), ctypes.byref(n))
        buf = ctypes.create_string_buffer(n.value + 1)
        if self.lib.dst_comm_trace(self._h, ctypes.c_int(enable), buf, ctypes.c_size_t(n.value + 1), ctypes.byref(n)) != DST_OK:
            raise DistaffError(DST_ERR_ARG, "dst_comm_trace failed")
        out = []
        for line in buf.value.decode().splitlines():
            k, b, st = line.split()
            out.append((k, int(b), None if st == "-" else int(st)))
        return out

    @staticmethod
    def unique_id():
        lib = load()
        buf = ctypes.create_string_buffer(128)
        if lib.dst_comm_unique_id(buf) != DST_OK:
            lib.dst_comm_last_error.restype = ctypes.c_char_p
            raise DistaffError(DST_ERR_HIP, lib.dst_comm_last_error(None).decode())
        return buf.raw

    @classmethod
    def rccl(cls, unique_id, rank, world, device):
        lib = load()
        h = ctypes.c_void_p()
        if lib.dst_comm_init(bytes(unique_id), ctypes.c_uint32(rank), ctypes.c_uint32(world), ctypes.c_int(device), ctypes.byref(h)) != DST_OK:
            lib.dst_comm_last_error.restype = ctypes.c_char_p
            raise DistaffError(DST_ERR_HIP, lib.dst_comm_last_error(None).decode())
        return cls(h)

    @classmethod
    def local(cls, world):
        """`world` handles whose ranks are threads of this process"""
        lib = load()
        arr = (ctypes.c_void_p * world)()
        if lib.dst_comm_init_local(ctypes.c_uint32(world), arr) != DST_OK:
            raise DistaffError(DST_ERR_ARG, "dst_comm_init_local failed")
        return [cls(ctypes.c_void_p(arr[r])) for r in range(world)]

    @classmethod
    def callbacks(cls, rank, world, fn):
        """The host's own transport: fn(kind, send_address, recv_address, bytes) -> 0 performs kind 0 = all-gather, 1 = all-to-all (device
        buffers), 2 = all-gather of host values."""
        lib = load()
        proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

        def tramp(user, kind, send, recv, nbytes):
            try:
                return int(fn(kind, send, recv, nbytes))
            except Exception:                                         # noqa: BLE001 -- reported as a failed collective
                import traceback
                traceback.print_exc()
                return 1
        cb = proto(tramp)
        h = ctypes.c_void_p()
        if lib.dst_comm_init_callbacks(ctypes.c_uint32(rank), ctypes.c_uint32(world), cb, None, ctypes.byref(h)) != DST_OK:
            raise DistaffError(DST_ERR_ARG, "dst_comm_init_callbacks failed")
        c = cls(h)
        c._keep = cb                                                  # the trampoline must outlive the handle
        return c

    @classmethod
    def over_torch(cls, dist, group=None):
        """dst_prove_sharded's collectives carried by an initialised torch.distributed process group through the callback transport
        (dst_comm_init_callbacks): the host's own channel instead of the library's RCCL binding.  With a gloo group the device buffers
        are staged through host memory (dst_comm_copy), so the ranks may be processes that SHARE one GPU -- which RCCL refuses --
        or sit on hosts without RCCL; with an nccl group the exchange runs on device tensors.  The callbacks block until the
        exchange is complete, as the transport's contract asks."""
        import numpy as np
        import torch
        lib = load()
        lib.dst_comm_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        on_device = str(dist.get_backend(group)).lower() == "nccl"

        def copy(dst, src, nbytes):
            if lib.dst_comm_copy(ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(nbytes)) != DST_OK:
                lib.dst_comm_last_error.restype = ctypes.c_char_p
                raise DistaffError(DST_ERR_HIP, lib.dst_comm_last_error(None).decode())

        def staged(src, nbytes):
            """`nbytes` at address `src` as a tensor the process group can send"""
            t = torch.empty(nbytes, dtype=torch.uint8, device="cuda" if on_device else "cpu")
            copy(t.data_ptr(), src, nbytes)
            return t

        def fn(kind, send, recv, nbytes):
            if nbytes == 0:
                return 0
            if kind == 1:                                             # all-to-all: chunk g of `send` goes to rank g
                mine = staged(send, nbytes * world)
                out = torch.empty_like(mine)
                if on_device:
                    dist.all_to_all_single(out, mine, group=group)
                    torch.cuda.synchronize()
                else:                                                 # gloo has no all-to-all on CPU tensors: gather and slice
                    every = [torch.empty_like(mine) for _ in range(world)]
                    dist.all_gather(every, mine, group=group)
                    out = torch.cat([e[rank * nbytes:(rank + 1) * nbytes] for e in every])
                copy(recv, out.data_ptr(), nbytes * world)
                return 0
            if kind == 2 and not on_device:                           # host values
                mine = torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(send)).copy())
            else:
                mine = staged(send, nbytes)                           # (an in-place all-gather hands in send == recv + rank * nbytes: staged first)
            if on_device:
                out = torch.empty(nbytes * world, dtype=torch.uint8, device="cuda")
                dist.all_gather_into_tensor(out, mine, group=group)
                torch.cuda.synchronize()
            else:
                every = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(every, mine, group=group)
                out = torch.cat(every)
            if kind == 2 and not on_device:
                ctypes.memmove(recv, out.data_ptr(), nbytes * world)
            else:
                copy(recv, out.data_ptr(), nbytes * world)
            return 0
        return cls.callbacks(rank, world, fn)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dst_comm_destroy(self._h)
            self._h = None


class Calibration:
    """The calibration kernels of the test / bench build (dependent multiplication chains, multiply-add peak, straight-line-code probe) on
    `device`, through a small context of libdistaff_hip_hooks.so -- the product library does not contain them."""

    def __init__(self, device=0):
        self.ctx = Context(10, 20, 1, 0, device=device, lib=load_hooks())
        if not self.ctx.lib.dst_test_hooks():
            raise RuntimeError("the calibration kernels need the test / bench build (libdistaff_hip_hooks.so)")

    def bench_mad(self, lanes=1 << 21, iters=2048):
        return self.ctx.bench_mad(lanes, iters)

    def bench_mulmod(self, lanes=1 << 20, iters=256, portable=False):
        return self.ctx.bench_mulmod(lanes, iters, portable)

    def bench_code(self, code_kib):
        return self.ctx.bench_code(code_kib)

    def bench_clock(self, lanes=1 << 20, iters=1024):
        """dst_bench_clock: MHz of the shader clock under four fe_mul chains per lane on every SIMD (median over the wavefronts)"""
        mhz = ctypes.c_double(0)
        self.ctx._check(self.ctx.lib.dst_bench_clock(self.ctx._h, ctypes.c_uint64(lanes), ctypes.c_uint32(iters), ctypes.byref(mhz)))
        return mhz.value

    def close(self):
        self.ctx.close()


def prove_sharded_local(contexts, inputs, outputs, cap=1 << 22):
    """dst_prove_sharded_local: one proof over `len(contexts)` contexts (rank r of world, trace uploaded on each), one thread per
    context inside the library, in-process transport."""
    lib = load()
    world = len(contexts)
    arr = (ctypes.c_void_p * world)(*[c._h.value for c in contexts])
    pub = make_public(inputs, outputs)
    buf = ctypes.create_string_buffer(cap)
    ln = ctypes.c_size_t(0)
    r = lib.dst_prove_sharded_local(arr, ctypes.c_uint32(world), ctypes.byref(pub), buf, ctypes.c_size_t(cap), ctypes.byref(ln))
    if r != DST_OK:
        msgs = [lib.dst_last_error(c._h).decode() for c in contexts]
        raise DistaffError(r, "; ".join("rank %d: %s" % (i, m) for i, m in enumerate(msgs) if m))
    return buf.raw[:ln.value]


class Context:
    """One proving job on one GPU (``dst_ctx``): owns the device buffers; phases mirror stark::prove (prover.rs:17-168)."""

    def __init__(self, log_n, width, ctx_depth, loop_depth, log_blowup=5, num_queries=50, grinding=20, device=0, rank=0, world=1, lib=None):
        self.lib = lib or load()
        self.params = Params(log_n, log_blowup, width, ctx_depth, loop_depth, num_queries, grinding, device, rank, world)
        self.n, self.B, self.W = 1 << log_n, 1 << log_blowup, width
        self.N = self.n * self.B
        h = ctypes.c_void_p()
        r = self.lib.dst_ctx_create(ctypes.byref(self.params), ctypes.byref(h))
        if r != DST_OK:
            raise DistaffError(r, self.lib.dst_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dst_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, r):
        if r != DST_OK:
            raise DistaffError(r, self.lib.dst_last_error(self._h).decode())

    def upload(self, columns):
        cols = np.ascontiguousarray(columns, dtype=np.uint64)
        assert cols.shape == (self.W, self.n, 2), cols.shape
        self._check(self.lib.dst_trace_upload_contiguous(self._h, _ptr(cols)))

    def pinned_trace(self, columns):
        """Copies the trace into page-locked host memory owned by the library and returns (pointer table, keep-alive handle) for
        upload_async: the host-resident TraceTable of the reference, placed where asynchronous DMA can read it."""
        cols = np.ascontiguousarray(columns, dtype=np.uint64)
        assert cols.shape == (self.W, self.n, 2), cols.shape
        p = ctypes.c_void_p()
        if self.lib.dst_pinned_alloc(ctypes.c_size_t(cols.nbytes), ctypes.byref(p)) != DST_OK:
            raise DistaffError(DST_ERR_HIP, "dst_pinned_alloc failed")
        ctypes.memmove(p, cols.ctypes.data, cols.nbytes)
        table = (ctypes.c_void_p * self.W)(*[p.value + i * self.n * 16 for i in range(self.W)])
        return table, p

    def release_pinned(self, handle):
        self.lib.dst_pinned_free(handle)

    def upload_owned(self, cols):
        """dst_trace_upload_owned: only the registers this rank interpolates (r = rank mod world) travel to the device; for prove_sharded"""
        cols = np.ascontiguousarray(cols, dtype=np.uint64)
        rank, world = self.params.rank, self.params.world
        ptrs = (ctypes.c_void_p * self.W)(*[cols[i].ctypes.data if i % world == rank else None for i in range(self.W)])
        self._check(self.lib.dst_trace_upload_owned(self._h, ptrs))

    def upload_async(self, table):
        """dst_trace_upload_async: returns at once, the next commit_trace() / prove() consumes the registers as they arrive"""
        self._check(self.lib.dst_trace_upload_async(self._h, table))

    def prove_sharded(self, comm, inputs, outputs, cap=1 << 22):
        """dst_prove_sharded: this rank's part of one proof over the communicator's ranks (every rank calls it, every rank gets the proof)"""
        pub = make_public(inputs, outputs)
        buf = ctypes.create_string_buffer(cap)
        ln = ctypes.c_size_t(0)
        self._check(self.lib.dst_prove_sharded(self._h, comm._h, ctypes.byref(pub), buf, ctypes.c_size_t(cap), ctypes.byref(ln)))
        return buf.raw[:ln.value]

    def shard_stage_ms(self):
        """host-side view of the last prove_sharded: ms inside the transport's calls, ms waiting for tree roots, tree exchanges"""
        v = (ctypes.c_double * 3)()
        self._check(self.lib.dst_shard_stage_ms(self._h, v))
        return {"transport_calls": v[0], "root_waits": v[1], "tree_exchanges": int(v[2])}

    def shard_exchange_ms(self):
        """dst_shard_exchange_ms: enqueue -> completion of the collectives of the last prove_sharded, per kind, from events on their streams"""
        v = (ctypes.c_double * 8)()
        self._check(self.lib.dst_shard_exchange_ms(self._h, v))
        return {"coefficients": v[0], "tree_all_to_all": v[1], "tree_all_gather": v[2], "constraint_evaluations": v[3], "fri_tail": v[4],
                "host_values": v[5], "collectives": int(v[6]), "timed_by_events": int(v[7])}

    def commit_trace(self):
        root = ctypes.create_string_buffer(32)
        self._check(self.lib.dst_commit_trace(self._h, root))
        return root.raw

    def eval_constraints(self, inputs, outputs, coeffs):
        c = np.ascontiguousarray(coeffs, dtype=np.uint64)
        assert c.shape == (344, 2)
        root = ctypes.create_string_buffer(32)
        bad = ctypes.c_int64(-1)
        pub = make_public(inputs, outputs)
        r = self.lib.dst_eval_constraints(self._h, ctypes.byref(pub), _ptr(c), root, ctypes.byref(bad))
        self.bad_step = bad.value
        self._check(r)
        return root.raw

    def compose(self, draws):
        d = np.ascontiguousarray(draws, dtype=np.uint64)
        assert d.shape == (516, 2)
        z1 = np.zeros((self.W, 2), dtype=np.uint64)
        z2 = np.zeros((self.W, 2), dtype=np.uint64)
        self._check(self.lib.dst_compose(self._h, _ptr(d), _ptr(z1), _ptr(z2)))
        return z1, z2

    def fri_commit_layer(self):
        root = ctypes.create_string_buffer(32)
        more = ctypes.c_int(0)
        self._check(self.lib.dst_fri_commit_layer(self._h, root, ctypes.byref(more)))
        return root.raw, bool(more.value)

    def fri_fold(self, special_x):
        self._check(self.lib.dst_fri_fold(self._h, int_to_bytes(special_x)))

    def pow_grind(self, seed, grinding):
        out = ctypes.create_string_buffer(32)
        nonce = ctypes.c_uint64(0)
        self._check(self.lib.dst_pow_grind(self._h, bytes(seed), ctypes.c_uint32(grinding), out, ctypes.byref(nonce)))
        return out.raw, nonce.value

    def build_proof(self, positions, pow_nonce):
        pos = np.asarray(positions, dtype=np.uint64)
        ln = ctypes.c_size_t(0)
        self._check(self.lib.dst_build_proof(self._h, _ptr(pos), ctypes.c_uint32(len(positions)), ctypes.c_uint64(pow_nonce), None, ctypes.c_size_t(0), ctypes.byref(ln)))
        buf = ctypes.create_string_buffer(ln.value)
        self._check(self.lib.dst_build_proof(self._h, _ptr(pos), ctypes.c_uint32(len(positions)), ctypes.c_uint64(pow_nonce), buf, ctypes.c_size_t(ln.value), ctypes.byref(ln)))
        return buf.raw[:ln.value]

    def prove(self, inputs, outputs, cap=1 << 22):
        """stark::prove on the uploaded trace; returns the serialised StarkProof."""
        pub = make_public(inputs, outputs)
        buf = ctypes.create_string_buffer(cap)
        ln = ctypes.c_size_t(0)
        self._check(self.lib.dst_prove(self._h, ctypes.byref(pub), buf, ctypes.c_size_t(cap), ctypes.byref(ln)))
        return buf.raw[:ln.value]

    def phase_ms(self):
        ms = (ctypes.c_double * 9)()
        self._check(self.lib.dst_phase_ms(self._h, ms))
        return list(ms)

    def read(self, what, arg=0):
        ln = ctypes.c_size_t(0)
        self._check(self.lib.dst_read_buffer(self._h, ctypes.c_uint32(BUF[what]), ctypes.c_uint32(arg), None, ctypes.c_size_t(0), ctypes.byref(ln)))
        buf = np.zeros(ln.value, dtype=np.uint8)
        self._check(self.lib.dst_read_buffer(self._h, ctypes.c_uint32(BUF[what]), ctypes.c_uint32(arg), _ptr(buf), ctypes.c_size_t(ln.value), ctypes.byref(ln)))
        return buf

    def read_elements(self, what, arg=0):
        return self.read(what, arg).view(np.uint64).reshape(-1, 2)

    # ---- coset-sharded phases (world > 1): see distaff_amd/sharded.py ------------------------------------------------
    def shard_commit_trace(self):
        self._check(self.lib.dst_shard_commit_trace(self._h))

    def shard_eval_constraints(self, inputs, outputs, coeffs):
        c = np.ascontiguousarray(coeffs, dtype=np.uint64)
        bad = ctypes.c_int64(-1)
        pub = make_public(inputs, outputs)
        r = self.lib.dst_shard_eval_constraints(self._h, ctypes.byref(pub), _ptr(c), ctypes.byref(bad))
        if r == DST_ERR_AIR:
            return bad.value
        self._check(r)
        return -1

    def shard_combine(self):
        self._check(self.lib.dst_shard_combine(self._h))

    def shard_fri_layer(self):
        more = ctypes.c_int(0)
        self._check(self.lib.dst_shard_fri_layer(self._h, ctypes.byref(more)))
        return bool(more.value)

    def fri_fold_shard(self, special_x):
        self._check(self.lib.dst_shard_fri_fold(self._h, int_to_bytes(special_x)))

    def shard_export_size(self, what, arg=0):
        n = ctypes.c_size_t(0)
        self._check(self.lib.dst_shard_export_size(self._h, ctypes.c_uint32(what), ctypes.c_uint32(arg), ctypes.byref(n)))
        return n.value

    def shard_export(self, what, arg, dst_ptr, is_device):
        self._check(self.lib.dst_shard_export(self._h, ctypes.c_uint32(what), ctypes.c_uint32(arg), ctypes.c_void_p(dst_ptr), int(is_device)))

    def shard_import(self, what, arg, src_ptr, is_device):
        root = ctypes.create_string_buffer(32)
        self._check(self.lib.dst_shard_import(self._h, ctypes.c_uint32(what), ctypes.c_uint32(arg), ctypes.c_void_p(src_ptr), int(is_device), root))
        return root.raw

    def shard_read(self, buffer, arg, indices):
        idx = np.asarray(indices, dtype=np.uint64)
        item = self.W * 16 if buffer == 10 else (16 if buffer in (3, 6, 11) else 32)
        out = np.zeros(len(indices) * item, dtype=np.uint8)
        self._check(self.lib.dst_shard_read(self._h, ctypes.c_uint32(buffer), ctypes.c_uint32(arg), _ptr(idx), ctypes.c_uint32(len(indices)), _ptr(out)))
        return out.tobytes()

    def shard_fri_begin(self, send_ptr, is_device, cap):
        """-> (bytes written into the send buffer, another layer follows)"""
        n, more = ctypes.c_size_t(0), ctypes.c_int(0)
        self._check(self.lib.dst_shard_fri_begin(self._h, ctypes.c_void_p(send_ptr), ctypes.c_int(1 if is_device else 0), ctypes.c_size_t(cap),
                                                 ctypes.byref(n), ctypes.byref(more)))
        return n.value, bool(more.value)

    def shard_fri_end(self, gathered_ptr, is_device):
        root = ctypes.create_string_buffer(32)
        self._check(self.lib.dst_shard_fri_end(self._h, ctypes.c_void_p(gathered_ptr), ctypes.c_int(1 if is_device else 0), root))
        return root.raw

    def shard_fri_roots(self):
        """-> (roots of all FRI layers after the commit phase, first layer of the replicated tail)"""
        layers, rep = ctypes.c_uint32(0), ctypes.c_uint32(0)
        self._check(self.lib.dst_shard_fri_roots(self._h, None, ctypes.c_size_t(0), ctypes.byref(layers), ctypes.byref(rep)))
        buf = ctypes.create_string_buffer(32 * layers.value)
        self._check(self.lib.dst_shard_fri_roots(self._h, buf, ctypes.c_size_t(32 * layers.value), ctypes.byref(layers), ctypes.byref(rep)))
        return [buf.raw[32 * d:32 * d + 32] for d in range(layers.value)], rep.value

    def shard_open(self, positions):
        """-> (this rank's blob of openings, [blob length of every rank])"""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        ln = ctypes.c_size_t(0)
        lens = np.zeros(self.params.world, dtype=np.uint64)
        self._check(self.lib.dst_shard_open(self._h, _ptr(pos), ctypes.c_uint32(len(pos)), None, ctypes.c_size_t(0), ctypes.byref(ln), _ptr(lens)))
        blob = np.zeros(max(ln.value, 1), dtype=np.uint8)
        self._check(self.lib.dst_shard_open(self._h, _ptr(pos), ctypes.c_uint32(len(pos)), _ptr(blob), ctypes.c_size_t(ln.value), ctypes.byref(ln), None))
        return blob[:ln.value], [int(v) for v in lens]

    def shard_assemble(self, positions, nonce, blobs, lens):
        """blobs: uint8 array holding the ranks' blobs back to back"""
        pos = np.ascontiguousarray(positions, dtype=np.uint64)
        bl = np.ascontiguousarray(blobs, dtype=np.uint8)
        ls = np.ascontiguousarray(lens, dtype=np.uint64)
        ln = ctypes.c_size_t(0)
        args = (self._h, _ptr(pos), ctypes.c_uint32(len(pos)), ctypes.c_uint64(nonce), _ptr(bl), _ptr(ls))
        self._check(self.lib.dst_shard_assemble(*args, None, ctypes.c_size_t(0), ctypes.byref(ln)))
        out = np.zeros(ln.value, dtype=np.uint8)
        self._check(self.lib.dst_shard_assemble(*args, _ptr(out), ctypes.c_size_t(ln.value), ctypes.byref(ln)))
        return out.tobytes()

    def shard_info(self):
        op = ctypes.c_uint64(0); layers = ctypes.c_uint32(0); sd = ctypes.c_uint32(0)
        self._check(self.lib.dst_shard_info(self._h, ctypes.byref(op), ctypes.byref(layers), ctypes.byref(sd)))
        return op.value, layers.value, sd.value

    def set_profiling(self, level=1):
        """0 / False: off, 1 / True: every kernel launch, 2: heavy kernels only (see include/distaff_hip.h)"""
        self._check(self.lib.dst_set_profiling(self._h, int(level)))

    def kernel_stats(self, reset=True):
        import json
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self.lib.dst_kernel_stats(self._h, buf, ctypes.c_size_t(1 << 16), int(reset)))
        return json.loads(buf.value.decode())

    def field_op(self, op, a, b):
        ops = {"add": 0, "sub": 1, "mul": 2, "mul_portable": 3, "inv": 4, "pow": 5, "dot40": 6}
        a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros_like(a)
        self._check(self.lib.dst_field_op(self._h, ops[op], _ptr(a), _ptr(b), _ptr(out), ctypes.c_size_t(a.shape[0])))
        return out

    def bench_mad(self, lanes=1 << 21, iters=2048):
        """milliseconds for lanes * iters * 32 multiply-adds v_mad_u64_u32 (the integer-multiplier peak of the device)"""
        ms = ctypes.c_double(0)
        self._check(self.lib.dst_bench_mad(self._h, ctypes.c_uint64(lanes), ctypes.c_uint32(iters), ctypes.byref(ms)))
        return ms.value

    def bench_code(self, code_kib):
        """milliseconds for 2^23 lanes to run 16 or 176 KiB of straight-line multiply-adds once (box fingerprint, kernels_probe.hip)"""
        ms = ctypes.c_double(0)
        self._check(self.lib.dst_bench_code(self._h, ctypes.c_uint32(code_kib), ctypes.byref(ms)))
        return ms.value

    def bench_mulmod(self, lanes=1 << 20, iters=256, portable=False):
        ms = ctypes.c_double(0)
        self._check(self.lib.dst_bench_mulmod(self._h, ctypes.c_uint64(lanes), ctypes.c_uint32(iters | (0x80000000 if portable else 0)), ctypes.byref(ms)))
        return ms.value
