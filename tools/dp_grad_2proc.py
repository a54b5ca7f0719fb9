"""Two-process rehearsal of data-parallel training on ONE GPU (gloo backend, both ranks on cuda:0): each rank
differentiates the loss of its column block (cnf_loss_grad), parallel.distributed_loss_and_grad combines them,
and the result is compared with the single-process gradient of the whole batch.
Run it from a process that has not touched the GPU:  python tools/dp_grad_2proc.py"""
import os
import socket
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("CNF_PERSISTENT", "0")      # two processes on one GPU: no one-launch solves (they need every CU)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import continuousnf.jl_amd as cnf
        from continuousnf.jl_amd import configs
        from continuousnf.jl_amd.parallel import distributed_loss_and_grad, shard_range
        wl = configs.BASELINE[3]
        B = 1001                                   # ragged: 501 + 500 columns
        flat = torch.from_numpy(configs.glorot_params(wl.dims, 7, 0.1)).cuda()
        xs, eps = configs.synthetic_inputs(wl, B, 7)
        lo, hi = shard_range(B, world, rank)
        ic = configs.build(wl, sol_kwargs=dict(adaptive=False, dt=1 / 8))
        cx = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        val, grad = distributed_loss_and_grad(ic, cnf.TrainMode(), cx(xs[:, lo:hi]), flat, {}, eps=cx(eps[:, lo:hi]))
        full = None
        if rank == 0:
            ic2 = configs.build(wl, sol_kwargs=dict(adaptive=False, dt=1 / 8))
            fv, fg = cnf.loss_and_grad(ic2, cnf.TrainMode(), cx(xs), flat, {}, eps=cx(eps))
            full = (fv, fg.cpu().numpy())
        q.put((rank, val, grad.cpu().numpy(), full))
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(60)
    fv, fg = got[0][3]
    for r, val, grad, _ in got:
        e = np.abs(grad - fg).max() / np.abs(fg).max()
        print(f"rank {r}: loss {val:.6f} (unsharded {fv:.6f}), max gradient difference {e:.2e} of the largest entry")
        assert abs(val - fv) <= 1e-5 * max(1.0, abs(fv)) and e < 1e-5
    assert np.array_equal(got[0][2], got[1][2])          # identical on both ranks: replicas stay in sync
    print("data-parallel gradient rehearsal OK")


if __name__ == "__main__":
    main()
