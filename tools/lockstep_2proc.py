"""Two-process rehearsal of the lock-step sharded solve on ONE GPU (gloo backend, both ranks on
cuda:0): each rank solves its column block of an adaptive problem with the error norm all-reduced
per step, and the concatenation is compared with the unsharded solve of rank 0.
Run it from a process that has not touched the GPU:  python tools/lockstep_2proc.py"""
import os
import socket
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("CNF_PERSISTENT", "0")      # two processes on one GPU: no one-launch solves (they need every CU)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import continuousnf.jl_amd as cnf
        from continuousnf.jl_amd.parallel import lockstep, shard_range
        from continuousnf.jl_amd import configs
        cfg = configs.BASELINE[3]
        B = 1000
        flat = configs.glorot_params(cfg.dims, 5, 0.2)
        xs, eps = configs.synthetic_inputs(cfg, B, 5)
        xs[:, B // 2:] *= 2.0
        kw = dict(configs.README_TOLERANCES)
        lo, hi = shard_range(B, world, rank)
        res = {}
        for lock in (True, False):
            icnf = configs.build(cfg, kernel="auto", sol_kwargs=kw)
            if lock:
                lockstep(icnf)
            p = cnf.inference_prob(icnf, cnf.TrainMode(), np.ascontiguousarray(xs[:, lo:hi]), flat, {},
                                   eps=np.ascontiguousarray(eps[:, lo:hi]))
            res[lock] = (cnf.base_sol(icnf, p).view().copy(), dict(p.stats))
        full = None
        if rank == 0:
            icnf = configs.build(cfg, kernel="auto", sol_kwargs=kw)
            p = cnf.inference_prob(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
            full = (cnf.base_sol(icnf, p).view().copy(), dict(p.stats))
        q.put((rank, res, full))
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
    full, fst = got[0][2]
    for lock in (True, False):
        cat = np.concatenate([g[1][lock][0] for g in got], axis=1)
        sts = [g[1][lock][1] for g in got]
        err = np.abs(cat - full).max() / np.abs(full).max()
        print(f"lockstep={lock}: shard stats {[(s['naccept'], s['nreject'], s['dt_last']) for s in sts]} "
              f"unsharded {(fst['naccept'], fst['nreject'], fst['dt_last'])} max rel diff {err:.2e}")
        if lock:
            assert sts[0]["dt_last"] == sts[1]["dt_last"] and sts[0]["naccept"] == sts[1]["naccept"]
            assert abs(sts[0]["naccept"] - fst["naccept"]) <= 1
            assert err < 1e-4
    print("lockstep 2-process rehearsal OK")


if __name__ == "__main__":
    main()
