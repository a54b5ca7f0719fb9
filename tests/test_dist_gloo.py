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


def _comm_worker(rank, world, port, fake, q):
    """cnf_comm_* end to end (the entry points a torch-less caller binds), RCCL replaced by the test double
    tests/support/fake_rccl.c: id on rank 0, broadcast over gloo, init, all-reduce of the 5 sums, destroy."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CNFHIP_RCCL_LIB=fake)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        from continuousnf.jl_amd import _lib
        l = _lib.lib()
        box = [None]
        if rank == 0:
            buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
            assert l.cnf_comm_unique_id(buf) == _lib.OK, l.cnf_comm_last_error()
            box[0] = buf.raw
        dist.broadcast_object_list(box, src=0)
        comm = C.c_void_p()
        assert l.cnf_comm_init(C.byref(comm), world, rank, box[0], -1) == _lib.OK, l.cnf_comm_last_error()
        n = C.c_int()
        assert l.cnf_comm_size(comm, C.byref(n)) == _lib.OK and n.value == world
        out = []
        for it in range(3):                    # several rounds: the slots are reused
            sums = np.array([1.0 + rank + it, 10.0 * (rank + 1), 0.25, -2.0 * rank, 19 - rank], dtype=np.float32)
            assert l.cnf_comm_allreduce(comm, sums.ctypes.data, sums.size, None) == _lib.OK, l.cnf_comm_last_error()
            out.append(sums.tolist())
        # bad arguments come back as status codes, never as crashes
        assert l.cnf_comm_allreduce(None, sums.ctypes.data, 5, None) == _lib.ERR_BAD_ARG
        bad = C.c_void_p()
        assert l.cnf_comm_init(C.byref(bad), world, world, box[0], -1) == _lib.ERR_BAD_ARG
        assert l.cnf_comm_destroy(comm) == _lib.OK
        # the key the ranks compare before cnf_comm_init: no device here, so a status code (and an empty key), not a crash
        key = C.create_string_buffer(192)
        assert l.cnf_comm_device_key(0, key, len(key)) in (_lib.ERR_NO_DEVICE, _lib.ERR_HIP) and key.value == b""
        assert l.cnf_comm_device_key(0, None, 0) == _lib.ERR_BAD_ARG
        q.put((rank, out, l.cnf_comm_library().decode()))
    finally:
        dist.destroy_process_group()


def test_comm_entry_points_ws2_with_rccl_test_double(tmp_path):
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    fake = str(tmp_path / "libfakerccl.so")
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", fake, os.path.join(here, "support", "fake_rccl.c"), "-lrt"],
                   check=True)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_comm_worker, args=(r, world, port, fake, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = {r: (o, lib) for r, o, lib in (q.get(timeout=120) for _ in range(world))}
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert res[0][0] == res[1][0]                    # every rank holds the same sums
    for it in range(3):
        assert res[0][0][it] == [3.0 + 2 * it, 30.0, 0.5, -2.0, 37.0]
    assert res[0][1] == fake


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment must itself start two rank processes that find
    each other (dry run: rendezvous and rank count only -- the solve needs a GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(CNF_BENCH_DRYRUN="1", CNF_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2
    # under a launcher (WORLD_SIZE set) it is one rank of many and must not spawn again
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and json.loads(r.stdout.splitlines()[-1])["ranks_seen"] == 1


def test_allreduce_is_identity_without_process_group():
    s = torch.arange(5, dtype=torch.float32)
    assert allreduce_sums(s) is s
    v = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    make_shard_reduce()(v)
    assert v.tolist() == [1.0, 2.0, 3.0]
