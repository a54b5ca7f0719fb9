"""World-size-2 `gloo` test of the sharded loss path: each rank takes its
column block, computes its local (logpx, E, n, A) -- here with the CPU oracle, since this
suite has no GPU -- and the 5-float all-reduce + mean equals the unsharded loss."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from continuousnf.jl_amd.parallel import allreduce_mean_weighted, allreduce_sums, loss_from_global_sums, make_shard_reduce, shard_range
from oracle import cnf_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(21)
    flat = O.glorot_params(cfg.net, rng, np.float64, 0.1)
    B = 37                                   # ragged on purpose: 19 + 18 columns
    xs = rng.standard_normal((cfg.nvars, B))
    eps = rng.standard_normal((cfg.n_in, B))
    return cfg, flat, xs, eps


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, flat, xs, eps = _problem()
        lo, hi = shard_range(xs.shape[1], world, rank)
        out = {}
        for train in (True, False):
            _, logpx, regs, _ = O.inference(cfg, flat, xs[:, lo:hi], eps[:, lo:hi], train,
                                            dt=1 / 16, adaptive=False)
            E, n, A = (np.zeros(hi - lo) if r is None else r for r in regs)
            sums = torch.tensor([logpx.sum(), E.sum(), n.sum(), A.sum(), hi - lo], dtype=torch.float32)
            g = allreduce_sums(sums)
            out[train] = (loss_from_global_sums(g, train, (cfg.lam1, cfg.lam2, cfg.lam3)), float(g[4]))
        # the lock-step controller's callback (cnf_set_shard_reduce): in-place sum of 3 floats
        v = np.array([1.5 + rank, 0.0, 100.0 * (rank + 1)], dtype=np.float32)
        make_shard_reduce()(v)
        out["lockstep"] = v.tolist()
        # data-parallel gradient (row f3 over shards): local mean loss/gradient (oracle) -> global mean
        from oracle import cnf_grad_oracle as G
        lv, lg, _ = G.loss_and_grad(cfg, flat, xs[:, lo:hi], eps[:, lo:hi], adaptive=False, dt=1 / 4)
        gv, gg = allreduce_mean_weighted(lv, lg.astype(np.float32), hi - lo)
        out["grad"] = (gv, gg.tolist())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_sharded_loss_matches_unsharded_gloo_ws2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    cfg, flat, xs, eps = _problem()
    for train in (True, False):
        _, logpx, regs, _ = O.inference(cfg, flat, xs, eps, train, dt=1 / 16, adaptive=False)
        ref = O.loss(cfg, logpx, regs, train)
        for r in range(world):
            got, cnt = res[r][train]
            assert cnt == xs.shape[1]
            assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref)), (got, ref)
    assert res[0] == res[1]      # every rank holds the same mean
    assert res[0]["lockstep"] == [4.0, 0.0, 300.0]
    from oracle import cnf_grad_oracle as G
    rv, rg, _ = G.loss_and_grad(cfg, flat, xs, eps, adaptive=False, dt=1 / 4)
    gv, gg = res[0]["grad"]
    assert abs(gv - rv) <= 1e-5 * max(1.0, abs(rv))
    assert np.abs(np.asarray(gg) - rg).max() <= 1e-5 * np.abs(rg).max()


def test_allreduce_is_identity_without_process_group():
    s = torch.arange(5, dtype=torch.float32)
    assert allreduce_sums(s) is s
    v = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    make_shard_reduce()(v)
    assert v.tolist() == [1.0, 2.0, 3.0]
