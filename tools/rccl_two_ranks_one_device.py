#!/usr/bin/env python3
"""Can two RCCL ranks share ONE device?  (the single-GPU test box can only run the RCCL transport of
dst_prove_sharded with world = 1.)  Two processes create a 2-rank communicator on device 0 through the library's own binding
(dst_comm_unique_id / dst_comm_init) and, if that works, run a sharded 2^10 proof.  Prints what happened; exit code 0 either way."""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, uid, q):
    try:
        import distaff_amd as D
        comm = D.Comm.rccl(uid, rank, 2, 0)
        cols, program_hash, result = D.fibonacci_trace(10)
        ctx = D.Context(10, 20, 1, 0, rank=rank, world=2)
        ctx.upload_owned(cols)
        proof = ctx.prove_sharded(comm, [1, 0], [result])
        q.put((rank, "proof of %d bytes, blake3 %s" % (len(proof), D.blake3(proof).hex()[:16])))
    except Exception as e:                                            # noqa: BLE001
        q.put((rank, "FAILED: %s: %s" % (type(e).__name__, e)))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    import distaff_amd as D
    uid = D.Comm.unique_id()
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(r, uid, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = []
    for p in ps:
        p.join(120)
        if p.is_alive():
            p.terminate(); out.append("a rank did not finish within 120 s (terminated)")
    while not q.empty():
        out.append("rank %d: %s" % q.get())
    print("two RCCL ranks on one device: " + ("; ".join(sorted(out)) or "no result"))
