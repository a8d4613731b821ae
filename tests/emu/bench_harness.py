"""TEST INFRASTRUCTURE: runs bench.py's own main() where there is no GPU -- the emulated build of the library (DISTAFF_HIP_LIB),
torch.distributed over gloo instead of RCCL, CPU tensors instead of device tensors -- so that the benchmark's control flow for
N > 1 ranks (sharded prover, tensor hand-off, max over ranks, one JSON line on rank 0) is exercised by `pytest -m "not gpu"`.
The numbers it prints mean nothing.   python tests/emu/bench_harness.py <bench.py arguments>"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 0
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None


def _on_cpu(fn):
    def wrapped(*args, **kwargs):
        if "device" in kwargs:
            kwargs["device"] = "cpu"
        return fn(*args, **kwargs)
    return wrapped


torch.zeros, torch.empty, torch.tensor = _on_cpu(torch.zeros), _on_cpu(torch.empty), _on_cpu(torch.tensor)
_init = dist.init_process_group


def _init_gloo(backend=None, **kwargs):
    kwargs.pop("device_id", None)
    return _init("gloo", **kwargs)


dist.init_process_group = _init_gloo

import distaff_amd as D
from distaff_amd import sharded

D.Calibration.bench_mulmod = lambda self, lanes, iters: 1.0       # the ALU calibration kernels would take minutes on the host
D.Calibration.bench_mad = lambda self, lanes, iters: 1.0


# bench.py finds no device here (torch.cuda.device_count() == 0 counts as one), so with N > 1 ranks it takes its "ranks share a device"
# route by itself: a gloo group and the library's collectives through the callback transport (distaff_amd.Comm.over_torch) -- the same
# code a one-GPU box runs with N processes; dst_comm_copy of the emulated build is a memcpy.
_comm_init = sharded.TorchComm.__init__


def _comm_on_cpu(self, dist_, device=None, device_path=False):
    _comm_init(self, dist_, torch.device("cpu") if device is not None else None, device_path)


sharded.TorchComm.__init__ = _comm_on_cpu
sys.argv = [os.path.join(ROOT, "bench.py"), "--allow-lib-override"] + sys.argv[1:]     # the emulated build is not the product: bench.py refuses it otherwise
runpy.run_path(sys.argv[0], run_name="__main__")
