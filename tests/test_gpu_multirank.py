"""The multi-rank path of today's bench.py on a GPU ("next round" item 1).

The driver's 1/2/4/8-GPU run needs a whole node and was skipped in every round, so the `world > 1` branch of bench.py --
column shards per rank (src/base_icnf.jl:266-286 on a rank's own columns), submitted solves, the all-reduce of the five
sums behind `loss` (src/icnf.jl:489), max-over-ranks timing -- is run here with TWO rank processes sharing the one card of
a gpurun box, over gloo (bench.py's rehearsal mode: CNF_BENCH_BACKEND=gloo; RCCL itself refuses two ranks on one GPU).
Asserted: both ranks were seen and timed, the all-reduce was timed, the loss every rank forms from the reduced sums is one
number, and it equals the loss of the CONCATENATED batch solved unsharded by this process (fixed dt, so the shards and the
unsharded solve take identical steps: 1e-5).  With CNF_BENCH_TRY_RCCL=1 the ranks first ATTEMPT the C ABI's own RCCL
communicator (cnf_comm_unique_id / cnf_comm_init): on one card RCCL answers "Duplicate GPU detected", and the test checks
that every rank takes the same clean fallback instead of hanging or dying."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra_env, args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(CNF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", CNF_NO_PARITY_REPORT="1")
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 alone prints
    return lines[0], r.stderr


def _unsharded_loss(B_per_rank, world, dt):
    """The loss of the concatenated batch (rank r's columns are synthetic_inputs(seed 1 + r), bench.py), one solve, this process."""
    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import configs
    wl = configs.BASELINE[3]
    flat = configs.glorot_params(wl.dims, 12345)
    parts = [configs.synthetic_inputs(wl, B_per_rank, 1 + r) for r in range(world)]
    xs = np.concatenate([p[0] for p in parts], axis=1)
    eps = np.concatenate([p[1] for p in parts], axis=1)
    icnf = configs.build(wl, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=dt))
    dev = torch.device("cuda", 0)
    xs_d = torch.from_numpy(np.ascontiguousarray(xs.T)).to(dev).t()
    eps_d = torch.from_numpy(np.ascontiguousarray(eps.T)).to(dev).t()
    _, _, sums = cnf.inference(icnf, cnf.TrainMode(), xs_d, torch.from_numpy(flat).to(dev), {}, eps=eps_d, with_sums=True)
    torch.cuda.synchronize()
    val = cnf.loss_from_sums(icnf, cnf.TrainMode(), sums)
    assert float(sums[4]) == B_per_rank * world
    icnf.close()
    return val


def test_bench_two_ranks_on_one_card_over_gloo():
    B, dt = 1024, 0.125
    line, err = _run_bench({"CNF_BENCH_TRY_RCCL": "1"},
                           ["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--no-reference-suite",
                            "--batch", str(B), "--fixed-dt", str(dt), "--prewarm", "0.05"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_2ranks_gloo.json"), "w") as f:
        json.dump(line, f, indent=1)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"
    assert len(line["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in line["per_rank_ms_per_step"])
    assert line["allreduce_us"] is not None and line["allreduce_us"] > 0
    assert line["config"]["global_batch"] == 2 * B
    # whole-job value = the evaluations of BOTH ranks over the slowest rank's time
    assert line["ms_per_step"] >= max(line["per_rank_ms_per_step"]) * 0.999
    nf = line["nf_per_solve"]
    assert nf == 1 + 6 * 8                                       # fixed dt = 1/8: 8 steps, no initial-dt probe
    assert abs(line["value"] - 2 * nf / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    # one loss on every rank, equal to the unsharded loss of the concatenated batch
    assert len(line["per_rank_loss"]) == 2 and np.isfinite(line["loss"])
    assert line["per_rank_loss"][0] == line["per_rank_loss"][1] == line["loss"]
    want = _unsharded_loss(B, 2, dt)
    assert abs(line["loss"] - want) <= 1e-5 * max(1.0, abs(want)), (line["loss"], want)
    # the C ABI's RCCL communicator was attempted by both ranks; on one card it is refused, and refused CLEANLY: the ranks
    # agreed, fell back to the rehearsal backend and finished (were it ever accepted, the collective is the ABI's own)
    att = line["rccl_attempt"]
    assert att is not None
    if att.startswith("refused"):
        assert line["collective"] == "torch.distributed all_reduce (gloo)"
    else:
        assert att == "accepted" and line["collective"].startswith("cnf_loss_allreduce")


def test_bench_two_ranks_adaptive_submitted_solves():
    """The driver's form of the run (adaptive solve, submitted steps, README tolerances) with two ranks: finishes, counts both
    ranks' evaluations, and the per-rank adaptive solves land within the solver tolerance of each other's mean log-density."""
    line, _ = _run_bench({}, ["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--no-reference-suite",
                              "--batch", "2048", "--prewarm", "0.05"])
    assert line["ranks_seen"] == 2 and line["steps_in_flight"] == 2
    assert np.isfinite(line["loss"]) and line["per_rank_loss"][0] == line["per_rank_loss"][1]
    assert line["nf_per_solve"] >= 50 and line["naccept"] >= 8
    assert line["value"] > 0 and line["rccl_attempt"] is None
