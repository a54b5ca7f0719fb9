"""Parity of the HIP path with the oracle, through the C ABI (via the host mirror), on a
real MI355X.  Bar: 1e-4 relative in fp32 (tests/helpers.py:RTOL)."""
import ctypes as C
import os

import numpy as np
import pytest

import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib
from oracle import c_oracle as CO
from oracle import cnf_oracle as O
from tests import helpers
from tests.helpers import GOLDEN_CASES, assert_parity, load_golden, make_icnf

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

KERNELS = ["generic", "mfma"]


def _skip_if_unsupported(icnf, mode, B):
    if icnf.compute_mode.kernel == "mfma":
        l = _lib.lib()
        if l.cnf_kernel_for(icnf.handle(), mode.cnf, B) != _lib.KERNEL_MFMA:
            pytest.skip("no MFMA kernel for this shape")


def _supported(icnf, mode, B):
    if icnf.compute_mode.kernel != "mfma":
        return True
    return _lib.lib().cnf_kernel_for(icnf.handle(), mode.cnf, B) == _lib.KERNEL_MFMA


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


# ---------------------------------------------------------------------------------------
# RHS level (rows a1, a2, a3)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_rhs_matches_golden(name, kernel):
    g, cfg = load_golden(name)
    for jvp in (False, True):
        icnf = make_icnf(cnf, cfg, jvp=jvp, kernel=kernel)
        if not _supported(icnf, cnf.TrainMode(), g["u_train"].shape[1]):
            icnf.close()
            continue
        tag = "jvp" if jvp else "vjp"
        # host arrays -> cnf_rhs_host; device tensors -> cnf_rhs
        du = cnf.augmented_f(g["u_train"], g["flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, g["eps"])
        assert_parity(du, g[f"du_train_{tag}"], f"{name} train {tag} host", trace_row=cfg.n_in)
        du_d = cnf.augmented_f(_dev(g["u_train"]), g["flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, _dev(g["eps"]))
        assert du_d.shape == g["u_train"].shape
        assert_parity(du_d.cpu().numpy(), g[f"du_train_{tag}"], f"{name} train {tag} device", trace_row=cfg.n_in)
        if _supported(icnf, cnf.TestMode(), g["u_train"].shape[1]):
            dt = cnf.augmented_f(g["u_train"][: cfg.n_in + 1], g["flat"], 0.0, icnf, cnf.TestMode(), icnf.nn, {}, None)
            assert_parity(dt, g["du_test"], f"{name} test", trace_row=cfg.n_in)
        else:   # an explicit request for a kernel that does not exist fails loudly
            with pytest.raises(cnf.CNFError) as e:
                cnf.augmented_f(g["u_train"][: cfg.n_in + 1], g["flat"], 0.0, icnf, cnf.TestMode(), icnf.nn, {}, None)
            assert e.value.status == _lib.ERR_UNSUPPORTED
        icnf.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_rhs_inplace_form(kernel):
    g, cfg = load_golden("cfg2_regression")
    icnf = make_icnf(cnf, cfg, kernel=kernel)
    _skip_if_unsupported(icnf, cnf.TrainMode(), g["u_train"].shape[1])
    u, eps = _dev(g["u_train"]), _dev(g["eps"])
    du = torch.full_like(u, float("nan"))
    r = cnf.augmented_f(du, u, g["flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, eps)   # icnf.jl:352-382
    assert r is None
    assert_parity(du.cpu().numpy(), g["du_train_vjp"], "in-place", trace_row=cfg.n_in)
    # du laid out as the reference keeps it (column-major D x B): written in place, no temporary -- device and host
    D, B = u.shape
    ucm, ecm = u.t().contiguous().t(), eps.t().contiguous().t()
    ducm = torch.full((B, D), float("nan"), device=u.device).t()
    assert cnf.augmented_f(ducm, ucm, g["flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, ecm) is None
    assert torch.equal(ducm, du)
    duh = np.full((D, B), np.nan, dtype=np.float32, order="F")
    assert cnf.augmented_f(duh, np.asfortranarray(g["u_train"], dtype=np.float32), g["flat"], 0.0, icnf, cnf.TrainMode(),
                           icnf.nn, {}, np.asfortranarray(g["eps"], dtype=np.float32)) is None
    assert np.array_equal(duh, du.cpu().numpy())


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 257, 1000])
def test_rhs_ragged_batches(B, kernel):
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(B)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    u = rng.standard_normal((cfg.D(True), B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    icnf = make_icnf(cnf, cfg, kernel=kernel)
    _skip_if_unsupported(icnf, cnf.TrainMode(), B)
    du = cnf.augmented_f(u, flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, eps)
    ref = cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), True)(u.astype(np.float64))
    assert_parity(du, ref, f"B={B}", trace_row=cfg.n_in)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("i", [1, 2, 3, 5])
def test_rhs_baseline_configs_vs_c_oracle(i, kernel):
    """Seeded inputs at sizes the C oracle finishes in seconds; all three modes."""
    cfg, B, _ = O.baseline_cfg(i)
    B = min(B, 1024 if i != 5 else 128)
    rng = np.random.default_rng(100 + i)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    u = rng.standard_normal((cfg.D(True), B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    for jvp in (False, True):
        cfg.use_jvp = jvp
        icnf = make_icnf(cnf, cfg, jvp=jvp, kernel=kernel)
        if not _supported(icnf, cnf.TrainMode(), B):
            icnf.close()
            continue
        du = cnf.augmented_f(u, flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, eps)
        ref = cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), True)(u.astype(np.float64))
        assert_parity(du, ref, f"cfg{i} train jvp={jvp}", trace_row=cfg.n_in)
        assert_parity(du, CO.rhs(cfg, flat, u, eps, True), f"cfg{i} vs C oracle", trace_row=cfg.n_in)
        icnf.close()
    cfg.use_jvp = False
    icnf = make_icnf(cnf, cfg, kernel=kernel)
    Bt = min(B, 64)
    if not _supported(icnf, cnf.TestMode(), Bt):
        return
    dt = cnf.augmented_f(u[: cfg.n_in + 1, :Bt], flat, 0.0, icnf, cnf.TestMode(), icnf.nn, {}, None)
    assert_parity(dt, CO.rhs(cfg, flat, u[: cfg.n_in + 1, :Bt], None, False), f"cfg{i} test", trace_row=cfg.n_in)


# ---------------------------------------------------------------------------------------
# solve level (rows a6, a7, a8, a9)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_fixed_dt_inference_matches_golden(name, kernel):
    g, cfg = load_golden(name)
    kw = dict(adaptive=False, dt=float(g["dt"]))
    for jvp in (False, True):
        tag = "jvp" if jvp else "vjp"
        icnf = make_icnf(cnf, cfg, jvp=jvp, kernel=kernel, sol_kwargs=kw)
        if not _supported(icnf, cnf.TrainMode(), g["xs"].shape[1]):
            icnf.close()
            continue
        prob = cnf.inference_prob(icnf, cnf.TrainMode(), _dev(g["xs"]), g["flat"], {}, eps=_dev(g["eps"]))
        fsol = cnf.base_sol(icnf, prob)
        assert prob.stats["nf"] == int(g[f"nf_train_{tag}"])
        assert_parity(fsol.view().cpu().numpy(), g[f"fsol_train_{tag}"], f"{name} fsol {tag}", trace_row=cfg.n_in)
        logpx, (E, n, A) = cnf.inference(icnf, cnf.TrainMode(), _dev(g["xs"]), g["flat"], {}, eps=_dev(g["eps"]))
        assert_parity(logpx.cpu().numpy(), g[f"logpx_train_{tag}"], f"{name} logpx {tag}")
        assert_parity(torch.stack([E, n, A]).cpu().numpy(), g[f"regs_train_{tag}"], f"{name} regs {tag}")
        L = cnf.loss(icnf, cnf.TrainMode(), _dev(g["xs"]), g["flat"], {}, eps=_dev(g["eps"]))
        assert abs(L - float(g[f"loss_train_{tag}"])) <= 1e-4 * max(1.0, abs(float(g[f"loss_train_{tag}"])))
        icnf.close()
    icnf = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw)
    if not _supported(icnf, cnf.TestMode(), g["xs"].shape[1]):
        return
    logpx, (E, n, A) = cnf.inference(icnf, cnf.TestMode(), g["xs"], g["flat"], {})      # host arrays
    assert_parity(logpx, g["logpx_test"], f"{name} logpx test")
    assert np.all(E == 0) and np.all(n == 0)
    assert_parity(A, g["A_test"], f"{name} A test") if np.any(g["A_test"]) else None
    L = cnf.loss(icnf, cnf.TestMode(), g["xs"], g["flat"], {})
    assert abs(L - float(g["loss_test"])) <= 1e-4 * max(1.0, abs(float(g["loss_test"])))
    d = cnf.ICNFDist(icnf, cnf.TestMode(), g["flat"], {})                              # dist_ext/core_icnf.jl:23-31
    assert_parity(cnf.logpdf(d, g["xs"]), g["logpx_test"], f"{name} logpdf")


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("i", [1, 2, 3])
def test_adaptive_solve_vs_oracles(i, kernel):
    """README tolerances (README.md:64-65).  Step-for-step the GPU controller must follow
    the float32 C oracle (same control law); values agree with a tight float64 solve to
    the solver tolerance."""
    cfg, B, _ = O.baseline_cfg(i)
    B = min(B, 512)
    rng = np.random.default_rng(200 + i)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    icnf = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw)
    _skip_if_unsupported(icnf, cnf.TrainMode(), B)
    prob = cnf.inference_prob(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
    fsol = cnf.base_sol(icnf, prob).view()
    st = prob.stats
    u0 = O.inference_u0(cfg, xs, True)
    cref, cst = CO.solve(cfg, flat, u0, eps, True, **kw)
    assert st["nf"] == 2 + 6 * (st["naccept"] + st["nreject"])
    assert abs(st["naccept"] - cst["naccept"]) <= 2, (st, cst)
    assert abs(st["t_final"] - cfg.tspan[1]) < 1e-6
    ref64, _ = O.tsit5_solve(cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), True),
                             u0.astype(np.float64), *cfg.tspan, reltol=1e-10, abstol=1e-10)
    assert_parity(fsol, ref64, f"cfg{i} adaptive vs float64", rtol=5e-3, trace_row=cfg.n_in)
    # two float32 adaptive solves that round differently take slightly different steps: each is within the solver
    # tolerance of the true solution, so their mutual distance gets the same bar as the distance to the float64 solve
    # (measured: 2e-3 relative at config 1, t in (0, 13); 1e-4..6e-4 elsewhere)
    assert_parity(fsol, cref, f"cfg{i} adaptive vs C oracle", rtol=5e-3, trace_row=cfg.n_in)


@pytest.mark.parametrize("kernel", KERNELS)
def test_backward_time_roundtrip(kernel):
    """Integrating t0->t1 and back returns the data rows (the direction `generate` uses)."""
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(7)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    B = 100
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    kw = dict(adaptive=False, dt=1 / 64)
    icnf = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw)
    _skip_if_unsupported(icnf, cnf.TrainMode(), B)
    prob = cnf.inference_prob(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
    u0 = prob.u0.view().copy()
    fwd = cnf.base_sol(icnf, prob)
    prob2 = cnf.ODEProblem(icnf, cnf.TrainMode(), fwd, prob.eps, (1.0, 0.0), flat)
    back = cnf.base_sol(icnf, prob2).view()
    assert np.max(np.abs(back[: cfg.n_in] - u0[: cfg.n_in])) < 2e-5
    assert np.max(np.abs(back[cfg.n_in])) < 2e-4           # dlogp integrates back to 0


# ---------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties (the oracle is too slow there)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", KERNELS)
def test_full_size_cfg3_properties(kernel):
    cfg, B, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(33)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    u = rng.standard_normal((cfg.D(True), B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    icnf = make_icnf(cnf, cfg, kernel=kernel)
    _skip_if_unsupported(icnf, cnf.TrainMode(), B)
    ud, ed = _dev(u), _dev(eps)
    f = lambda uu, ee: cnf.augmented_f(uu, flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, ee)
    du = f(ud, ed)
    assert torch.isfinite(du).all()
    # columns are independent: any permutation of the columns permutes the output bit for bit
    perm = torch.from_numpy(rng.permutation(B)).cuda()
    assert torch.equal(f(ud[:, perm].contiguous(), ed[:, perm].contiguous()), du[:, perm])
    # a sub-batch reproduces its columns bit for bit (no cross-column coupling, ragged tail)
    assert torch.equal(f(ud[:, 5:1234].contiguous(), ed[:, 5:1234].contiguous()), du[:, 5:1234])
    # the RHS reads only the z rows of u (src/icnf.jl:330)
    u2 = ud.clone(); u2[cfg.n_in:] += 7.0
    assert torch.equal(f(u2, ed), du)
    # eps -> -eps leaves eps^T J eps and both norms unchanged; zdot does not depend on eps
    dm = f(ud, -ed)
    assert torch.equal(dm[: cfg.n_in], du[: cfg.n_in])
    assert torch.allclose(dm[cfg.n_in:], du[cfg.n_in:], rtol=1e-5, atol=1e-6)
    # E row is the column norm of zdot; sampled columns against the float64 oracle
    assert torch.allclose(du[cfg.n_in + 1], du[: cfg.n_in].norm(dim=0), rtol=1e-5)
    idx = rng.choice(B, 256, replace=False)
    ref = cfg.rhs(flat.astype(np.float64), eps[:, idx].astype(np.float64), True)(u[:, idx].astype(np.float64))
    assert_parity(du[:, torch.from_numpy(idx).cuda()].cpu().numpy(), ref, "cfg3 sampled columns", trace_row=cfg.n_in)
    # FFJORD switches (lambda = 0): regulariser rows are exactly zero (icnf.jl:337-340, 344-347)
    cfg4, _, _ = O.baseline_cfg(4)
    ic4 = make_icnf(cnf, cfg4, kernel=kernel)
    d4 = cnf.augmented_f(ud, flat, 0.0, ic4, cnf.TrainMode(), ic4.nn, {}, ed)
    assert torch.all(d4[cfg.n_in + 1:] == 0) and torch.equal(d4[: cfg.n_in + 1], du[: cfg.n_in + 1])


@pytest.mark.parametrize("kernel", KERNELS)
def test_full_size_cfg3_solve_is_shard_invariant(kernel):
    """Fixed dt: solving the whole batch equals solving two column shards separately
    (what the multi-GPU path relies on), bit for bit."""
    cfg, B, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(34)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    xs = _dev(rng.standard_normal((cfg.nvars, B)).astype(np.float32))
    eps = _dev(rng.standard_normal((cfg.n_in, B)).astype(np.float32))
    icnf = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=dict(adaptive=False, dt=1 / 8))
    _skip_if_unsupported(icnf, cnf.TrainMode(), B)
    lp, (E, n, A) = cnf.inference(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
    assert icnf.last_stats["nf"] == 1 + 6 * 8
    h = 3000
    lp1, r1 = cnf.inference(icnf, cnf.TrainMode(), xs[:, :h].contiguous(), flat, {}, eps=eps[:, :h].contiguous())
    lp2, r2 = cnf.inference(icnf, cnf.TrainMode(), xs[:, h:].contiguous(), flat, {}, eps=eps[:, h:].contiguous())
    assert torch.equal(torch.cat([lp1, lp2]), lp)
    assert torch.equal(torch.cat([r1[0], r2[0]]), E) and torch.equal(torch.cat([r1[1], r2[1]]), n)
    # the 5-float reduction composes: sums of shards == sums of the whole (to fp32 rounding)
    s = cnf.loss_sums(icnf, lp, (E, n, A)).cpu().numpy()
    s12 = (cnf.loss_sums(icnf, lp1, r1) + cnf.loss_sums(icnf, lp2, r2)).cpu().numpy()
    assert s[4] == B and np.allclose(s, s12, rtol=1e-5)


def test_hutchinson_mean_approaches_exact_trace():
    """TrainMode's -eps^T J eps averages to TestMode's -tr J (the two estimators of row
    n_in+1 agree in expectation)."""
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(9)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    icnf = make_icnf(cnf, cfg)
    z = rng.standard_normal((cfg.n_in, 1)).astype(np.float32)
    N = 32768
    u = np.vstack([np.repeat(z, N, 1), np.zeros((3, N), np.float32)])
    eps = rng.standard_normal((cfg.n_in, N)).astype(np.float32)
    tr_h = cnf.augmented_f(_dev(u), flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, _dev(eps))[cfg.n_in].cpu().numpy()
    tr_e = cnf.augmented_f(u[: cfg.n_in + 1, :1], flat, 0.0, icnf, cnf.TestMode(), icnf.nn, {}, None)[cfg.n_in, 0]
    assert abs(tr_h.mean() - tr_e) < 4 * tr_h.std() / np.sqrt(N)


# ---------------------------------------------------------------------------------------
# edge cases and error behaviour
# ---------------------------------------------------------------------------------------
def test_empty_batch_and_errors():
    g, cfg = load_golden("calltests_aug")
    icnf = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=0.25))
    D = cfg.D(True)
    du = cnf.augmented_f(np.zeros((D, 0), np.float32), g["flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {},
                         np.zeros((cfg.n_in, 0), np.float32))
    assert du.shape == (D, 0)
    with pytest.raises(ValueError):      # wrong row count
        cnf.augmented_f(np.zeros((D + 1, 3), np.float32), g["flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {},
                        np.zeros((cfg.n_in, 3), np.float32))
    with pytest.raises(ValueError):      # TrainMode without eps
        cnf.augmented_f(np.zeros((D, 3), np.float32), g["flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, None)
    with pytest.raises(cnf.CNFError) as e:   # wrong parameter count
        cnf.augmented_f(np.zeros((D, 3), np.float32), g["flat"][:-1], 0.0, icnf, cnf.TrainMode(), icnf.nn, {},
                        np.zeros((cfg.n_in, 3), np.float32))
    assert e.value.status == _lib.ERR_BAD_SHAPE
    # raw ABI: params not set, aliasing, maxiters
    l = _lib.lib()
    ic2 = make_icnf(cnf, cfg)
    h = ic2.handle()
    x = torch.zeros(D * 4, device="cuda")
    assert l.cnf_rhs(h, 1, 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, None) == _lib.ERR_NO_PARAMS
    ic2.set_params(g["flat"])
    assert l.cnf_rhs(h, 1, 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, None) == _lib.ERR_BAD_ARG
    assert b"alias" in l.cnf_last_error(h)
    assert l.cnf_rhs(h, 7, 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, None) == _lib.ERR_BAD_ARG
    ic3 = make_icnf(cnf, cfg, sol_kwargs=dict(reltol=1e-6, abstol=1e-9, maxiters=2))
    with pytest.raises(cnf.CNFError) as e:
        cnf.inference(ic3, cnf.TrainMode(), g["xs"], g["flat"], {}, eps=g["eps"])
    assert e.value.status == _lib.ERR_MAXITERS


def test_explicit_mfma_request_fails_loudly_when_unsupported():
    net = O.Net((5, 7, 3, 5), (O.ACT_SOFTPLUS, O.ACT_SWISH, O.ACT_IDENTITY))
    cfg = O.Cfg(net, 3, 2)
    icnf = make_icnf(cnf, cfg, kernel="mfma")
    flat = O.glorot_params(net, np.random.default_rng(0), np.float32)
    if _lib.lib().cnf_kernel_for(icnf.handle(), 1, 8) == _lib.KERNEL_MFMA:
        pytest.skip("MFMA path covers this shape")
    with pytest.raises(cnf.CNFError) as e:
        cnf.augmented_f(np.zeros((8, 8), np.float32), flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {},
                        np.zeros((5, 8), np.float32))
    assert e.value.status == _lib.ERR_UNSUPPORTED


def test_bench_line_on_this_gpu():
    """bench.py end to end on the GPU in a child process, as the driver runs it (N = 1): one JSON line with BASELINE.json's
    metric, the roofline and CPU-baseline objects, the submitted steps and the one-at-a-time figure beside them."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "RHS-evals/s" and d["n_gpus"] == 1
    assert d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e4 and abs(d["value"] - d["nf_per_solve"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0.0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9 and rf["launch_us"] * 1e-3 <= d["ms_per_step"] * 1.05
    if _one_launch_expected():
        # roofline.traffic: HBM bytes per launch from two rocprofv3 --pmc child passes of this very run (a committed figure,
        # labelled as such, only where the profiler is not available): far below the algorithmic bytes -- the state stays on chip
        assert rf["traffic"] is None or 1e6 < rf["traffic"] < rf["algorithmic_bytes_per_launch"], rf["traffic"]
        assert rf["traffic_source"] and ("measured in this run" in rf["traffic_source"] or "not measured" in rf["traffic_source"])
        assert d["launches_per_solve"] == 1 and d["steps_in_flight"] == 2
        assert d["one_at_a_time"]["value"] > 1e4 and d["one_at_a_time"]["ms_per_step"] >= 0.9 * d["ms_per_step"]


def test_rhs_work_model():
    cfg, B, _ = O.baseline_cfg(3)
    icnf = make_icnf(cnf, cfg)
    fl, by = C.c_double(), C.c_double()
    _lib.check(_lib.lib().cnf_rhs_work(icnf.handle(), 1, B, C.byref(fl), C.byref(by)))
    M = 32 * 128 + 128 * 128 + 128 * 32
    P = M + 128 + 128 + 32
    assert fl.value == B * (4 * M + 6 * 32) and by.value == 4 * B * (32 + 32 + 35) + 4 * P


@pytest.mark.parametrize("kernel", KERNELS)
def test_generate_matches_backward_oracle_and_inverts_inference(kernel):
    """`generate` = the same RHS over reverse(tspan) from a base-distribution
    draw (src/base_icnf.jl:358-380, :202-211).  Checked against the float64 oracle integrated
    backwards, and as the inverse map of `inference` (x -> z -> x)."""
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(17)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    n = 96
    kw = dict(adaptive=False, dt=1 / 32)
    icnf = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw)
    if not _supported(icnf, cnf.TestMode(), n):
        pytest.skip("no MFMA kernel")
    z0 = rng.standard_normal((cfg.n_in, n)).astype(np.float32)
    xs = cnf.generate(icnf, cnf.TestMode(), flat, {}, n, z0=z0)
    assert xs.shape == (cfg.nvars, n)
    u0 = np.vstack([z0, np.zeros((1, n), np.float32)]).astype(np.float64)
    ref, _ = O.tsit5_solve(cfg.rhs(flat.astype(np.float64), None, False), u0, 1.0, 0.0, dt=1 / 32, adaptive=False)
    assert_parity(xs, ref[: cfg.nvars], "generate vs backward oracle")
    # default draw: right shape, finite, reproducible from the rng seed
    a = cnf.rand(cnf.ICNFDist(make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw, rng=5), cnf.TestMode(), flat, {}), 40)
    b = cnf.rand(cnf.ICNFDist(make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw, rng=5), cnf.TestMode(), flat, {}), 40)
    assert a.shape == (cfg.nvars, 40) and np.all(np.isfinite(a)) and np.array_equal(a, b)
    # without augmentation the flow is a bijection: generate(inference's z) returns x
    cfg0 = O.Cfg(O.Net((8, 24, 8), (O.ACT_TANH,) * 2), 8, 0, tspan=(0.0, 1.0))
    flat0 = O.glorot_params(cfg0.net, rng, np.float32, 0.05)
    ic0 = make_icnf(cnf, cfg0, kernel=kernel, sol_kwargs=dict(adaptive=False, dt=1 / 64))
    x = rng.standard_normal((8, 50)).astype(np.float32)
    prob = cnf.inference_prob(ic0, cnf.TestMode(), x, flat0, {})
    z = cnf.base_sol(ic0, prob).view()[:8]
    back = cnf.generate(ic0, cnf.TestMode(), flat0, {}, 50, z0=np.array(z))
    assert np.max(np.abs(back - x)) < 5e-5


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dims,nvars,naugs", [((11, 24, 8), 8, 0), ((37, 64, 48, 32), 24, 8), ((6, 5, 3), 2, 1)])
def test_conditional_models(kernel, dims, nvars, naugs):
    """CondRNODE -- nn(vcat(z, ys)) (src/layers/cond_layer.jl:7-9,
    src/base_icnf.jl:288-309), AD products w.r.t. z only.  RHS in all three modes and a
    fixed-dt inference against the float64 oracle."""
    n_in = nvars + naugs
    n_cond = dims[0] - n_in
    net = O.Net(dims, (O.ACT_TANH,) * (len(dims) - 1))
    rng = np.random.default_rng(sum(dims))
    flat = O.glorot_params(net, rng, np.float32, 0.1)
    B = 77
    xs = rng.standard_normal((nvars, B)).astype(np.float32)
    ys = rng.standard_normal((n_cond, B)).astype(np.float32)
    eps = rng.standard_normal((n_in, B)).astype(np.float32)
    u = rng.standard_normal((n_in + 3, B)).astype(np.float32)
    layers = [cnf.Dense(i, o, "tanh") for i, o in zip(dims[:-1], dims[1:])]
    f64 = lambda a: a.astype(np.float64)
    for jvp in (False, True):
        cm = cnf.HIPJacVecMatrixMode(kernel) if jvp else cnf.HIPVecJacMatrixMode(kernel)
        icnf = cnf.construct(cnf.CondRNODE, cnf.Chain(*layers), nvars, naugs, compute_mode=cm,
                             lambda3=1e-2 if naugs else 0.0, sol_kwargs=dict(adaptive=False, dt=1 / 16))
        assert icnf.cond and icnf.n_cond == n_cond
        if not _supported(icnf, cnf.TrainMode(), B):
            icnf.close()
            continue
        cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, 1e-2 if naugs else 0.0, jvp)
        nnc = cnf.CondLayer(icnf.nn, ys)
        du = cnf.augmented_f(u, flat, 0.0, icnf, cnf.TrainMode(), nnc, {}, eps)
        assert_parity(du, cfg.rhs(f64(flat), f64(eps), True, f64(ys))(f64(u)), f"cond train jvp={jvp}", trace_row=n_in)
        # device arrays
        du_d = cnf.augmented_f(_dev(u), flat, 0.0, icnf, cnf.TrainMode(), cnf.CondLayer(icnf.nn, _dev(ys)), {}, _dev(eps))
        assert_parity(du_d.cpu().numpy(), du, "cond device == host", rtol=1e-6, trace_row=n_in)
        if _supported(icnf, cnf.TestMode(), B):
            dt = cnf.augmented_f(u[: n_in + 1], flat, 0.0, icnf, cnf.TestMode(), nnc, {}, None)
            assert_parity(dt, cfg.rhs(f64(flat), None, False, f64(ys))(f64(u[: n_in + 1])), "cond test", trace_row=n_in)
        logpx, (E, n, A) = cnf.inference(icnf, cnf.TrainMode(), xs, ys, flat, {}, eps=eps)
        _, ref_lp, (rE, rn, rA), st = O.inference(cfg, f64(flat), f64(xs), f64(eps), True, f64(ys), dt=1 / 16, adaptive=False)
        assert_parity(logpx, ref_lp, f"cond logpx jvp={jvp}")
        assert_parity(np.stack([E, n, A]), np.stack([rE, rn, rA]), f"cond regs jvp={jvp}")
        L = cnf.loss(icnf, cnf.TrainMode(), xs, ys, flat, {}, eps=eps)
        assert abs(L - O.loss(cfg, ref_lp, (rE, rn, rA), True)) <= 1e-4 * max(1.0, abs(L))
        d = cnf.CondICNFDist(icnf, cnf.TrainMode(), ys, flat, {})
        assert_parity(cnf.logpdf(d, xs, eps=eps), ref_lp, "CondICNFDist logpdf")
        # a different ys changes the result; the stale-ys guard of the raw ABI
        lp2, _ = cnf.inference(icnf, cnf.TrainMode(), xs, ys + 1.0, flat, {}, eps=eps)
        assert np.max(np.abs(lp2 - logpx)) > 1e-3
        icnf.close()
    ic = cnf.construct(cnf.CondFFJORD, cnf.Chain(*layers), nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode(kernel))
    if _supported(ic, cnf.TrainMode(), B):
        with pytest.raises(ValueError):      # conditional model without ys
            cnf.augmented_f(u, flat, 0.0, ic, cnf.TrainMode(), ic.nn, {}, eps)
        ic.set_params(flat)
        x = torch.zeros(B * (n_in + 3), device="cuda")
        assert _lib.lib().cnf_rhs(ic.handle(), 1, 0, x.data_ptr(), x.data_ptr(), x.data_ptr() + 4, B, None) == _lib.ERR_NO_PARAMS


def test_conditional_model_of_the_headline_shape_is_one_launch():
    """CondRNODE / CondFFJORD with the headline network behind the conditioning columns ((32 + 5)-128-128-32): the
    per-sample first-layer bias rows (cnf_set_cond) are read by the one-launch solve itself -- one tile and several tiles
    per workgroup, VJP and JVP compute modes -- against the float64 oracle on sampled columns and the streamed / generic
    kernels on the rest."""
    dims, nvars, n_cond = (37, 128, 128, 32), 32, 5
    net = O.Net(dims, (O.ACT_TANH,) * 3)
    rng = np.random.default_rng(3705)
    flat = O.glorot_params(net, rng, np.float32, 0.1)
    layers = [cnf.Dense(i, o, "tanh") for i, o in zip(dims[:-1], dims[1:])]
    f64 = lambda a: a.astype(np.float64)
    for B in (1000, 8224):
        xs = rng.standard_normal((nvars, B)).astype(np.float32)
        ys = rng.standard_normal((n_cond, B)).astype(np.float32)
        eps = rng.standard_normal((nvars, B)).astype(np.float32)
        for tag, jvp in ((cnf.CondRNODE, False), (cnf.CondRNODE, True), (cnf.CondFFJORD, False)):
            cm = cnf.HIPJacVecMatrixMode("mfma") if jvp else cnf.HIPVecJacMatrixMode("mfma")
            kw = dict(adaptive=False, dt=1 / 8)
            lam = dict(lambda1=1e-2, lambda2=1e-2) if tag is cnf.CondRNODE else {}
            ic = cnf.construct(tag, cnf.Chain(*layers), nvars, 0, compute_mode=cm, sol_kwargs=kw, **lam)
            logpx, (E, n, A) = cnf.inference(ic, cnf.TrainMode(), _dev(xs), _dev(ys), flat, {}, eps=_dev(eps))
            if _one_launch_expected():
                assert ic.last_stats["launches"] <= 3, (B, jvp, ic.last_stats)
            idx = rng.choice(B, 40, replace=False)
            cfg = O.Cfg(net, nvars, 0, 1e-2 if lam else 0.0, 1e-2 if lam else 0.0, 0.0, jvp)
            _, ref_lp, (rE, rn, rA), _ = O.inference(cfg, f64(flat), f64(xs[:, idx]), f64(eps[:, idx]), True, f64(ys[:, idx]),
                                                      dt=1 / 8, adaptive=False)
            ii = torch.from_numpy(idx).cuda()
            assert_parity(logpx[ii].cpu().numpy(), ref_lp, f"cond headline logpx B={B} jvp={jvp}")
            assert_parity(np.stack([E[ii].cpu().numpy(), n[ii].cpu().numpy()]), np.stack([rE, rn]), f"cond headline regs B={B} jvp={jvp}")
            # a different ys changes the result
            lp2, _ = cnf.inference(ic, cnf.TrainMode(), _dev(xs), _dev(ys + 1.0), flat, {}, eps=_dev(eps))
            assert float((lp2 - logpx).abs().max()) > 1e-3
            ic.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_call_matrix_of_the_reference(kernel):
    """The loops of test/call_tests.jl:1-252 at its own sizes (nvars = 2, ndata = 4): model types x modes x augmentation and
    steering x in-place flag x compute modes (here: the two HIP matrix modes), and for each the calls it makes --
    inference, generate, loss, the Lux-layer call, (Cond)ICNFDist logpdf / pdf / rand(d) / rand(d, n), and the gradients
    of the loss w.r.t. ps and w.r.t. the data.  The reference asserts `!isnothing`; here every result is finite as well,
    and inference is compared with the float64 oracle."""
    nvars, ndata = 2, 4
    rng = np.random.default_rng(2024)
    f64 = lambda a: None if a is None else a.astype(np.float64)
    for mt in (cnf.RNODE, cnf.FFJORD, cnf.Planar, cnf.CondRNODE, cnf.CondFFJORD, cnf.CondPlanar):
        cond = mt in (cnf.CondRNODE, cnf.CondFFJORD, cnf.CondPlanar)
        planar = mt in (cnf.Planar, cnf.CondPlanar)
        for aug_steer in (False, True):
            naugs = nvars if aug_steer else 0
            n_in = nvars + naugs
            n_cond = nvars if cond else 0
            if planar:               # Lux.Chain(PlanarLayer(n_in, tanh; n_cond)): the MLP (n_in + n_cond) -> 1 -> n_in to the kernels
                chain = cnf.Chain(cnf.PlanarLayer(n_in, "tanh", n_cond=n_cond))
                dims = chain.dims
                net = O.Net(dims, (O.ACT_TANH, O.ACT_IDENTITY))
                layers = None
                flat_ext = cnf.setup(int(rng.integers(1 << 30)), chain)[0]
                flat_int = chain.to_internal(flat_ext)
            else:
                dims = (n_in + n_cond, 3 * n_in, n_in)
                net = O.Net(dims, (O.ACT_TANH, O.ACT_TANH))
                layers = [cnf.Dense(i, o, "tanh") for i, o in zip(dims[:-1], dims[1:])]
                chain = cnf.Chain(*layers)
                flat_ext = flat_int = O.glorot_params(net, rng, np.float32, 0.1)
            flat = flat_ext
            r = rng.standard_normal((nvars, ndata)).astype(np.float32)
            r2 = rng.standard_normal((nvars, ndata)).astype(np.float32) if cond else None
            eps = rng.standard_normal((n_in, ndata)).astype(np.float32)
            for inplace in (False, True):
                for cm in (cnf.HIPVecJacMatrixMode(kernel), cnf.HIPJacVecMatrixMode(kernel)):
                    jvp = isinstance(cm, cnf.HIPJacVecMatrixMode)
                    lam = dict(lambda1=1e-2, lambda2=1e-2) if mt in (cnf.RNODE, cnf.CondRNODE) else {}
                    icnf = cnf.construct(mt, chain, nvars, naugs, compute_mode=cm, inplace=inplace,
                                         steer_rate=1e-1 if aug_steer else 0.0, lambda3=1e-2 if aug_steer else 0.0,
                                         sol_kwargs=dict(adaptive=False, dt=1 / 8), **lam)
                    for omode in (cnf.TrainMode(), cnf.TestMode()):
                        train = isinstance(omode, cnf.TrainMode)
                        if not _supported(icnf, omode, ndata):
                            continue
                        args = (r2, flat, {}) if cond else (flat, {})
                        # (steering draws the end time from icnf.rng in TrainMode: fixed here so that the oracle runs the same span)
                        icnf.steer_rate = 0.0
                        logpx, regs = cnf.inference(icnf, omode, r, *args, eps=eps if train else None)
                        cfg = O.Cfg(net, nvars, naugs, lam.get("lambda1", 0.0), lam.get("lambda2", 0.0),
                                    1e-2 if aug_steer else 0.0, jvp)
                        _, ref_lp, _, _ = O.inference(cfg, f64(flat_int), f64(r), f64(eps) if train else None, train, f64(r2),
                                                      dt=1 / 8, adaptive=False)
                        assert_parity(logpx, ref_lp, f"call matrix {mt.__name__} aug={aug_steer} ip={inplace} jvp={jvp} train={train}")
                        icnf.steer_rate = 1e-1 if aug_steer else 0.0
                        assert np.isfinite(cnf.inference(icnf, omode, r, *args)[0]).all()
                        g = cnf.generate(icnf, omode, flat, {}, ndata, ys=r2)
                        assert g.shape == (nvars, ndata) and np.isfinite(g).all()
                        assert np.isfinite(cnf.loss(icnf, omode, r, *args))
                        out, st_ = icnf((r, r2), flat, {}) if cond else icnf(r, flat, {})
                        assert np.isfinite(out).all() and st_ == {}
                        d = cnf.CondICNFDist(icnf, omode, r2, flat, {}) if cond else cnf.ICNFDist(icnf, omode, flat, {})
                        assert np.isfinite(cnf.logpdf(d, r)).all() and np.isfinite(cnf.pdf(d, r)).all()
                        assert cnf.rand(d).shape == (nvars,) and cnf.rand(d, ndata).shape == (nvars, ndata)
                        # (src/exts/dist_ext/core.jl:6-12, core_icnf.jl:34-58: length, eltype, rand!)
                        assert len(d) == nvars and d.eltype == np.float32
                        buf = np.full((nvars, ndata), np.nan, np.float32)
                        assert cnf.rand_(d, buf) is buf and np.isfinite(buf).all()
                        assert np.isfinite(cnf.rand_(d, np.full(nvars, np.nan, np.float32))).all()
                        if not train and kernel != "generic":
                            # (TestMode gradients: every network of this matrix runs on k_solve_wave<TEST, GRAD>)
                            val, gps, gx = cnf.loss_and_grad(icnf, omode, r, *args, with_x=True)
                            assert np.isfinite(val) and np.isfinite(gps).all() and gps.shape == flat.shape
                            assert gx.shape == r.shape and np.isfinite(gx).all()
                        if train:            # (every model type: the mode the reference trains in)
                            val, gps, gx = cnf.loss_and_grad(icnf, omode, r, *args, with_x=True)
                            assert np.isfinite(val) and np.isfinite(gps).all() and gps.shape == flat.shape
                            assert gx.shape == r.shape and np.isfinite(gx).all()
                            if planar:           # the (u, w, b) gradient against the oracle's, mapped from the MLP layout
                                from oracle import cnf_grad_oracle as G
                                icnf.steer_rate = 0.0
                                v2, g2 = cnf.loss_and_grad(icnf, omode, r, *args, eps=eps)
                                rv, rg, _ = G.loss_and_grad(cfg, f64(flat_int), f64(r), f64(eps), f64(r2), adaptive=False, dt=1 / 8)
                                assert abs(v2 - rv) <= 1e-5 * max(1.0, abs(rv))
                                _assert_grad(g2, chain.grad_to_external(rg), f"planar gradient aug={aug_steer} jvp={jvp}", rtol=2e-4)
                    icnf.close()


def test_full_size_cfg5_properties():
    """BASELINE config 5 at full size (RNODE 64+64, MLP 128-384-128, B = 2048): weights stay in
    HBM/L2, exact trace in closed form.  Size-independent checks + sampled columns vs the oracle."""
    cfg, B, _ = O.baseline_cfg(5)
    rng = np.random.default_rng(55)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    icnf = make_icnf(cnf, cfg, kernel="mfma")
    u = rng.standard_normal((cfg.D(True), B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    ud, ed = _dev(u), _dev(eps)
    tr = cnf.augmented_f(ud, flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, ed)
    te = cnf.augmented_f(ud[: cfg.n_in + 1].contiguous(), flat, 0.0, icnf, cnf.TestMode(), icnf.nn, {}, None)
    assert torch.isfinite(tr).all() and torch.isfinite(te).all()
    # both modes compute the same zdot; permutation of the columns permutes the output bit for bit
    assert torch.allclose(tr[: cfg.n_in], te[: cfg.n_in], rtol=0, atol=0)
    perm = torch.from_numpy(rng.permutation(B)).cuda()
    assert torch.equal(cnf.augmented_f(ud[:, perm].contiguous(), flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {},
                                       ed[:, perm].contiguous()), tr[:, perm])
    # the Hutchinson row is an unbiased estimate of the exact-trace row: batch means agree statistically
    d = (tr[cfg.n_in] - te[cfg.n_in]).cpu().numpy()
    assert abs(d.mean()) < 5 * d.std() / np.sqrt(B)
    idx = rng.choice(B, 64, replace=False)
    ref_te = cfg.rhs(flat.astype(np.float64), None, False)(u[: cfg.n_in + 1, idx].astype(np.float64))
    assert_parity(te[:, torch.from_numpy(idx).cuda()].cpu().numpy(), ref_te, "cfg5 exact trace, sampled columns", trace_row=cfg.n_in)
    ref_tr = cfg.rhs(flat.astype(np.float64), eps[:, idx].astype(np.float64), True)(u[:, idx].astype(np.float64))
    assert_parity(tr[:, torch.from_numpy(idx).cuda()].cpu().numpy(), ref_tr, "cfg5 Hutchinson, sampled columns", trace_row=cfg.n_in)
    # exact logpdf (README usage: ICNFDist(icnf, TestMode(), ps, st)) vs a float64 solve on a few columns
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    lp = cnf.logpdf(cnf.ICNFDist(ic, cnf.TestMode(), flat, {}), _dev(xs)).cpu().numpy()
    _, ref_lp, _, _ = O.inference(cfg, flat.astype(np.float64), xs[:, :24].astype(np.float64), None, False,
                                  dt=1 / 8, adaptive=False)
    assert_parity(lp[:24], ref_lp, "cfg5 exact logpdf")


@pytest.mark.parametrize("kernel", KERNELS)
def test_lockstep_shards_follow_the_unsharded_solve(kernel):
    """SURVEY 8(e) option 2 through cnf_set_shard_reduce: two shards (two handles driven from two
    threads; the callback is an in-process all-reduce) take exactly the step sequence of the
    unsharded adaptive solve, whereas independent shard solves do not."""
    import threading
    cfg, _, _ = O.baseline_cfg(2)
    B, cut = 300, 130                         # ragged shards on purpose
    rng = np.random.default_rng(77)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.3)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    xs[:, cut:] *= 2.5                        # make the shards differ in stiffness
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    full = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw)
    _skip_if_unsupported(full, cnf.TrainMode(), B)
    prob = cnf.inference_prob(full, cnf.TrainMode(), xs, flat, {}, eps=eps)
    ref = cnf.base_sol(full, prob).view().copy()
    ref_st = dict(prob.stats)

    shards = [(0, cut), (cut, B)]

    def run(lock):
        icnfs = [make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=kw) for _ in shards]
        bar, bufs = threading.Barrier(len(shards)), [None] * len(shards)
        calls = [0] * len(shards)

        def reducer(r):
            def f(v):
                calls[r] += 1
                bufs[r] = v.copy()
                bar.wait()
                tot = bufs[0] + bufs[1]
                bar.wait()
                v[:] = tot
            return f
        out, stats, errs = [None] * 2, [None] * 2, []

        def work(r):
            try:
                lo, hi = shards[r]
                if lock:
                    icnfs[r].set_shard_reduce(reducer(r))
                p = cnf.inference_prob(icnfs[r], cnf.TrainMode(), np.ascontiguousarray(xs[:, lo:hi]), flat, {},
                                       eps=np.ascontiguousarray(eps[:, lo:hi]))
                out[r] = cnf.base_sol(icnfs[r], p).view().copy()
                stats[r] = dict(p.stats)
            except Exception as e:            # pragma: no cover
                errs.append(e)
                bar.abort()
        th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(120)
        assert not errs, errs
        return np.concatenate(out, axis=1), stats, calls

    got, st, calls = run(True)
    # every shard takes bit-identical decisions; against the unsharded solve the global sums differ
    # only by the association order of the float additions (1 ulp), which the controller follows
    # smoothly except at its dead zone (1 <= q <= 1.2 -> q = 1), so dt is compared loosely
    assert st[0]["dt_last"] == st[1]["dt_last"] and st[0]["naccept"] == st[1]["naccept"]
    assert abs(st[0]["naccept"] - ref_st["naccept"]) <= 1 and st[0]["nreject"] == ref_st["nreject"], (st, ref_st)
    # (the unsharded solve of a small network runs on another kernel -- k_solve_wave -- than the lock-step shards -- step
    # launches of k_mfma: where the two take a different number of steps, the last, clipped step differs as well)
    if st[0]["naccept"] == ref_st["naccept"]:
        assert abs(st[0]["dt_last"] - ref_st["dt_last"]) <= 0.25 * ref_st["dt_last"]
    assert calls[0] == calls[1] == 2 + st[0]["naccept"] + st[0]["nreject"]
    assert_parity(got, ref, "lock-step shards vs unsharded", rtol=1e-4, trace_row=cfg.n_in)

    ind, st_i, _ = run(False)
    assert st_i[0]["dt_last"] != st_i[1]["dt_last"]                   # the coupling the lock-step removes
    assert_parity(ind, ref, "independent shards vs unsharded", rtol=5e-3, trace_row=cfg.n_in)   # still within solver tolerance

    # switching the callback off again restores independent solves; fixed-dt never calls it
    ic = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=dict(adaptive=False, dt=1 / 16))
    hit = []
    ic.set_shard_reduce(lambda v: hit.append(1))
    cnf.inference(ic, cnf.TrainMode(), xs[:, :64].copy(), flat, {}, eps=eps[:, :64].copy())
    assert not hit


# ---------------------------------------------------------------------------------------
# Row f3: gradient of the training loss w.r.t. the parameters (cnf_loss_grad)
# ---------------------------------------------------------------------------------------
def _grad_case(cfg, B, seed, kernel, sol_kw, ora_kw, jvp=False, n_cond=0, scale=0.2, host=False):
    from oracle import cnf_grad_oracle as G
    rng = np.random.default_rng(seed)
    net = cfg.net
    if n_cond:
        net = O.Net((cfg.net.dims[0] + n_cond,) + tuple(cfg.net.dims[1:]), cfg.net.acts)
    flat = O.glorot_params(net, rng, np.float32, scale)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    ys = rng.standard_normal((n_cond, B)).astype(np.float32) if n_cond else None
    if n_cond:
        layers = [cnf.Dense(a, b, helpers.ACT_NAME[k]) for a, b, k in zip(net.dims[:-1], net.dims[1:], net.acts)]
        cm = cnf.HIPJacVecMatrixMode(kernel) if jvp else cnf.HIPVecJacMatrixMode(kernel)
        icnf = cnf.construct(cnf.CondRNODE, cnf.Chain(*layers), cfg.nvars, cfg.naugs, compute_mode=cm, tspan=cfg.tspan,
                             lambda1=cfg.lam1, lambda2=cfg.lam2, lambda3=cfg.lam3, sol_kwargs=sol_kw)
    else:
        icnf = make_icnf(cnf, cfg, jvp=jvp, kernel=kernel, sol_kwargs=sol_kw)
    args = (ys, flat, {}) if n_cond else (flat, {})
    conv = (lambda a: a) if host else _dev
    cargs = tuple(conv(a) if isinstance(a, np.ndarray) and a is not flat else a for a in args)
    val, grad, gx = cnf.loss_and_grad(icnf, cnf.TrainMode(), conv(xs), *cargs, eps=conv(eps), with_x=True)
    grad = grad.cpu().numpy() if hasattr(grad, "cpu") else grad
    gx = gx.cpu().numpy() if hasattr(gx, "cpu") else gx
    c64 = O.Cfg(net, cfg.nvars, cfg.naugs, cfg.lam1, cfg.lam2, cfg.lam3, use_jvp=jvp, tspan=cfg.tspan)
    if ora_kw == "replay":            # differentiate exactly the steps the device took
        ora_kw = dict(dts=[float(d) for d in icnf.last_steps])
    rval, rgrad, st = G.loss_and_grad(c64, flat.astype(np.float64), xs.astype(np.float64), eps.astype(np.float64),
                                      None if ys is None else ys.astype(np.float64), **ora_kw)
    # the gradient w.r.t. the data (test/call_tests.jl differentiates the loss w.r.t. x as well as ps): the adjoint state at t0
    if "dts" in ora_kw or not sol_kw.get("adaptive", True):        # (same discrete map on both sides)
        assert gx.shape == xs.shape
        _assert_grad(gx, st.grad_x, "d loss / d xs", rtol=2e-4)
    return val, grad, rval, rgrad, icnf.last_stats, st


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("i", [1, 2, 3])
def test_loss_grad_fixed_dt_matches_oracle(i, kernel):
    """Fixed dt: the discrete adjoint on the device vs the float64 oracle (itself pinned by torch
    autograd, tests/test_grad_oracle.py).  Tolerance: 1e-4 of the gradient's scale."""
    cfg, _, _ = O.baseline_cfg(i)
    cfg.tspan = (0.0, 1.0)
    B = 96
    val, grad, rval, rgrad, st, _ = _grad_case(cfg, B, 300 + i, kernel, dict(adaptive=False, dt=1 / 8),
                                               dict(adaptive=False, dt=1 / 8))
    assert st["naccept"] == 8
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    assert np.isfinite(grad).all()
    scale = np.sqrt(np.mean(rgrad ** 2))
    assert np.abs(grad - rgrad).max() <= 1e-4 * (np.abs(rgrad).max() + scale), (np.abs(grad - rgrad).max(), scale)


def _assert_grad(grad, rgrad, what, rtol=1e-4):
    assert np.isfinite(grad).all(), what
    scale = np.sqrt(np.mean(rgrad ** 2))
    err = np.abs(grad - rgrad).max()
    assert err <= rtol * (np.abs(rgrad).max() + scale), (what, err, np.abs(rgrad).max(), scale)


@pytest.mark.parametrize("kernel", KERNELS)
def test_loss_grad_adaptive_jvp_cond_and_host(kernel):
    """Adaptive solve, JVP compute mode, a conditional model, host arrays, ragged batches."""
    from oracle import cnf_grad_oracle as G
    cfg, _, _ = O.baseline_cfg(2)
    cfg.tspan = (0.0, 1.0)
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    # adaptive: float32 and float64 controllers take different steps at these tolerances (the error
    # estimate of the rows that start at 0 is rounding noise against abstol = eps), so the oracle
    # replays the steps the device took (cnf_grad_steps) and differentiates the same discrete map
    val, grad, rval, rgrad, st, ost = _grad_case(cfg, 77, 311, kernel, dict(tol), "replay")
    assert st["naccept"] == ost.naccept and st["naccept"] > 4 and st["nf"] == 2 + 6 * (st["naccept"] + st["nreject"])
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad, rgrad, "adaptive")
    # JVP compute mode (n-dot = |J eps|), fixed dt
    val, grad, rval, rgrad, _, _ = _grad_case(cfg, 50, 312, kernel, dict(adaptive=False, dt=1 / 6),
                                              dict(adaptive=False, dt=1 / 6), jvp=True)
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad, rgrad, "jvp")
    # conditional model: gradient also w.r.t. the conditioning columns of W1
    val, grad, rval, rgrad, _, _ = _grad_case(cfg, 40, 313, kernel, dict(adaptive=False, dt=1 / 6),
                                              dict(adaptive=False, dt=1 / 6), n_cond=5)
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad, rgrad, "conditional")
    # host arrays in, host gradient out
    val, grad, rval, rgrad, _, _ = _grad_case(cfg, 33, 314, kernel, dict(adaptive=False, dt=1 / 6),
                                              dict(adaptive=False, dt=1 / 6), host=True)
    assert isinstance(grad, np.ndarray)
    _assert_grad(grad, rgrad, "host")


def test_loss_grad_headline_shape_variants():
    """The resident-fragment pullback k_adj3 (shapes that pad to 32-128-128-32, 32 samples per workgroup): other
    activations, a conditional model, widths below the padding, ragged and tiny batches, FFJORD (no norm rows)."""
    cases = [
        (O.Cfg(O.Net((32, 128, 128, 32), (O.ACT_SOFTPLUS, O.ACT_SIGMOID, O.ACT_TANH)), 32, 0, 0.01, 0.01), 0, 45),
        (O.Cfg(O.Net((28, 128, 128, 28), (O.ACT_TANH,) * 3), 28, 0, 0.01, 0.01), 4, 70),
        (O.Cfg(O.Net((30, 120, 116, 30), (O.ACT_TANH, O.ACT_ELU, O.ACT_TANH)), 20, 10, 0.01, 0.01, 0.01), 0, 33),
        (O.Cfg(O.Net((32, 128, 128, 32), (O.ACT_TANH,) * 3), 32, 0), 0, 1),
        (O.Cfg(O.Net((32, 128, 128, 32), (O.ACT_TANH,) * 3), 32, 0, 0.01, 0.01), 0, 300),
    ]
    for k, (cfg, n_cond, B) in enumerate(cases):
        cfg.tspan = (0.0, 0.5)
        val, grad, rval, rgrad, _, _ = _grad_case(cfg, B, 340 + k, "mfma", dict(adaptive=False, dt=1 / 4),
                                                  dict(adaptive=False, dt=1 / 4), n_cond=n_cond, scale=0.1)
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval)), k
        _assert_grad(grad, rgrad, f"headline-shape case {k}")
    # JVP handle of the same shape: the forward steps of the recorded solve leave k_step3j (it files no stage states) for
    # the run-time-layout kernel, whose grid differs -- the partial count of the controller must follow
    cfg = O.Cfg(O.Net((32, 128, 128, 32), (O.ACT_TANH,) * 3), 32, 0, 0.01, 0.01)
    cfg.tspan = (0.0, 0.5)
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    val, grad, rval, rgrad, st, ost = _grad_case(cfg, 90, 349, "mfma", dict(tol), "replay", jvp=True, scale=0.1)
    assert st["naccept"] == ost.naccept
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad, rgrad, "headline shape, JVP handle, adaptive")


def test_ab_switches_take_the_other_kernels_and_stay_parity_green():
    """CNF_PERSISTENT=0 (the streamed step launches instead of the one-launch solve), CNF_SOLVE_POLL_LIMIT=1 (every wait of the one-launch solve runs out: the streamed
    fallback), CNF_STEP_FP32 (the fp32-MFMA step
    kernels k_step3 / k_step3j instead of the split-bf16 ones), CNF_STEP_V1 /
    CNF_TRACE_GENERIC / CNF_ADJ_GENERIC (the first-generation kernels), CNF_WGRAD_LDS (the contraction with LDS images,
    k_wgrad_mfma_b, instead of the wave-local k_wgrad_wave), CNF_WAVE_GRAD=0 (the streamed gradient path for the small networks
    k_solve_wave<GRAD> otherwise takes), CNF_WAVE_WG=0 (one workgroup per wave at small batches), CNF_WAVE_RICH=0 (the gradient
    recomputes the forward half of its stages): read once per process; each
    route runs its parity tests in a child process."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for var, sel in (("CNF_PERSISTENT=0", "test_adaptive_solve_vs_oracles and 3-mfma or test_full_size_cfg3 or "
                                          "test_one_launch_solve or test_headline_kernels_strict or test_submitted_inferences or "
                                          "test_conditional_model_of_the_headline"),
                     ("CNF_SOLVE_POLL_LIMIT=1", "test_one_launch_solve_falls_back or test_submitted_inferences"),
                     ("CNF_BCAST=0", "test_config5_one_launch_solve_strict and 1000"),
                     ("CNF_WAVE_GRAD=0", "test_loss_grad_fixed_dt_matches_oracle or test_loss_grad_adaptive_jvp_cond or "
                                         "test_loss_grad_more_steps_than_the_trajectory_store or test_fit_readme_example"),
                     ("CNF_WAVE_WG=0", "test_wave_local_solve_small_networks or test_one_layer_network"),
                     ("CNF_WAVE_RICH=0", "test_loss_grad_wave_local_small or test_submitted_gradients or test_one_layer_network"),
                     ("CNF_TRACE_SOLVE=0", "test_exact_trace_mfma_deep_networks or test_testmode_headline_network_is_three_launches"),
                     ("CNF_STEP_FP32", "test_adaptive_solve_vs_oracles and 3-mfma or test_full_size_cfg3_solve or "
                                       "test_jvp_mode_headline_shape_step_kernel or ragged or test_headline_kernels_strict"),
                     ("CNF_STEP_V1", "test_adaptive_solve_vs_oracles and 3-mfma or test_full_size_cfg3_solve"),
                     ("CNF_TRACE_GENERIC", "test_exact_trace_mfma_deep_networks"),
                     ("CNF_TRACE_FP32", "test_exact_trace_mfma_deep_networks"),
                     ("CNF_ADJ_GENERIC", "test_loss_grad_fixed_dt_matches_oracle and 3-mfma or test_loss_grad_headline"),
                     ("CNF_WGRAD_LDS", "test_loss_grad_fixed_dt_matches_oracle and 3-mfma or test_loss_grad_headline")):
        name, _, val = var.partition("=")
        env = dict(os.environ, **{name: val or "1", "CNF_NO_PARITY_REPORT": "1"})
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x",
                            "-m", "gpu", "-p", "no:cacheprovider", "-k", sel], env=env, cwd=root, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, (var, r.stdout[-2000:], r.stderr[-1000:])
        assert " passed" in r.stdout and "failed" not in r.stdout, (var, r.stdout[-500:])


def test_submitted_inferences_equal_the_synchronous_ones():
    """cnf_inference_submit / cnf_inference_collect: up to three inferences enqueued back to back on one stream, collected
    oldest first -- outputs, loss sums and statistics equal those of the synchronous call bit for bit (the same launch,
    only waited for later), on every route (one launch, several tiles per workgroup, streamed, the forced fallback of
    the CNF_SOLVE_POLL_LIMIT=1 child run); a fourth submission is refused, any other call on the handle collects first,
    a submission that fails reports at its collect."""
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(911)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    tol = dict(reltol=3.45e-4, abstol=1.19e-7)
    forced = os.environ.get("CNF_SOLVE_POLL_LIMIT") == "1"
    for kernel, Bs in (("mfma", (8192, 1000, 8224)), ("generic", (64, 33, 7))):
        ic = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=tol)
        ins = [(_dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))) for B in Bs]
        want = []
        for xs, eps in ins:
            lp, regs, sums = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True)
            want.append((lp.clone(), [r.clone() for r in regs], sums.clone(), dict(ic.last_stats)))
        fb0 = ic.solve_fallbacks()
        outs = [cnf.inference_submit(ic, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True) for xs, eps in ins]
        assert _lib.lib().cnf_inference_pending(ic.handle()) == 3
        with pytest.raises(cnf.CNFError):
            cnf.inference_submit(ic, cnf.TrainMode(), ins[0][0], flat, {}, eps=ins[0][1])
        for (lp, regs, sums), (wlp, wregs, wsums, wst) in zip(outs, want):
            st = cnf.inference_collect(ic)
            torch.cuda.synchronize()
            assert torch.equal(lp, wlp) and all(torch.equal(a, b) for a, b in zip(regs, wregs)) and torch.equal(sums, wsums)
            for key in ("nf", "naccept", "nreject", "kernel_used"):
                assert st[key] == wst[key], (kernel, key, st, wst)
            if kernel == "mfma" and _one_launch_expected() and not forced and wst["launches"] == 1:
                assert st["launches"] == 1, st
        assert _lib.lib().cnf_inference_pending(ic.handle()) == 0
        if forced and kernel == "mfma" and _one_launch_expected():
            assert ic.solve_fallbacks() - fb0 == 3, ic.solve_fallbacks() - fb0
        with pytest.raises(cnf.CNFError):
            cnf.inference_collect(ic)                                   # nothing submitted
        # any other call on the handle completes what is submitted first
        o1 = cnf.inference_submit(ic, cnf.TrainMode(), ins[1][0], flat, {}, eps=ins[1][1])
        lp2, _ = cnf.inference(ic, cnf.TrainMode(), ins[2][0], flat, {}, eps=ins[2][1])
        torch.cuda.synchronize()
        assert _lib.lib().cnf_inference_pending(ic.handle()) == 0
        assert torch.equal(o1[0], want[1][0]) and torch.equal(lp2, want[2][0])
        ic.close()
    # every submission runs with ITS conditioning and ITS parameters (mini-batches of a conditional model; an optimiser
    # step between two submissions): the library settles what is queued before either changes
    n_cond = 8
    net = O.Net((32 + n_cond, 128, 128, 32), (O.ACT_TANH,) * 3)
    layers = [cnf.Dense(a, b, "tanh") for a, b in zip(net.dims[:-1], net.dims[1:])]
    flats = [O.glorot_params(net, rng, np.float32, 0.1) for _ in range(2)]
    ic = cnf.construct(cnf.CondRNODE, cnf.Chain(*layers), 32, 0, compute_mode=cnf.HIPVecJacMatrixMode("mfma"), sol_kwargs=tol)
    batches = [(_dev(rng.standard_normal((32, 512))), _dev(rng.standard_normal((n_cond, 512))), _dev(rng.standard_normal((32, 512))),
                flats[i // 2]) for i in range(3)]                       # ys differs per batch, ps changes before the third
    want = [cnf.inference(ic, cnf.TrainMode(), xs, ys, fl, {}, eps=eps)[0].clone() for xs, ys, eps, fl in batches]
    outs = [cnf.inference_submit(ic, cnf.TrainMode(), xs, ys, fl, {}, eps=eps)[0] for xs, ys, eps, fl in batches]
    assert _lib.lib().cnf_inference_pending(ic.handle()) == 3 and len(ic._submitted) == 3
    for _ in batches:
        cnf.inference_collect(ic)
    torch.cuda.synchronize()
    assert not ic._submitted
    for k, (got, w) in enumerate(zip(outs, want)):
        assert torch.equal(got, w), ("conditional submission", k, float((got - w).abs().max()))
    assert not torch.equal(want[0], want[1])
    ic.close()
    # an inference that fails (maxiters) reports at ITS collect, not at the submit; the handle goes on working
    ic = make_icnf(cnf, cfg, sol_kwargs=dict(tol, maxiters=2))
    xs, eps = ins[1] if False else (_dev(rng.standard_normal((cfg.nvars, 512))), _dev(rng.standard_normal((cfg.n_in, 512))))
    cnf.inference_submit(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    with pytest.raises(cnf.CNFError):
        cnf.inference_collect(ic)
    ic.sol_kwargs = dict(tol)
    ic._opts_cache = None
    a = cnf.inference_submit(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    cnf.inference_collect(ic)
    b, _ = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b)
    ic.close()


def test_bare_solve_reads_and_writes_the_callers_columns():
    """cnf_solve_tsit5 on caller-owned columns (src/base_icnf.jl:137-143: `base_sol` hands over `prob.u0` and takes the last
    state): the one-launch solve reads u0 where it is and writes the final columns where they are wanted -- ONE launch,
    no copies around it -- out of place, in place (u_out = u0), and beyond one tile per workgroup (state in the
    integrator's buffers: a copy out); all equal to the solve through the handle's own buffers (inference on the same
    columns is the reference)."""
    import ctypes as C
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(5)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    persistent = os.environ.get("CNF_PERSISTENT") != "0" and os.environ.get("CNF_STEP_FP32") != "1" \
        and os.environ.get("CNF_STEP_V1") != "1"
    for jvp in (False, True):
        for B in (1000, 8224):
            D = cfg.n_in + 3
            ic = make_icnf(cnf, cfg, jvp=jvp, sol_kwargs=dict(adaptive=False, dt=1 / 8))
            ic.set_params(flat)
            l, h = _lib.lib(), ic.handle()
            u0 = _dev(rng.standard_normal((B, D))).contiguous()
            u0[:, cfg.n_in:] = 0
            eps = _dev(rng.standard_normal((B, cfg.n_in))).contiguous()
            sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            opts = _lib.cnf_solve_opts(0.0, 1.0, 0.0, 0.0, 1 / 8, 0, 1 << 20, 0)
            stats = _lib.cnf_solve_stats()
            keep = u0.clone()
            out = torch.full_like(u0, float("nan"))
            _lib.check(l.cnf_solve_tsit5(h, 1, u0.data_ptr(), eps.data_ptr(), out.data_ptr(), B, C.byref(opts), C.byref(stats), sp), h)
            torch.cuda.synchronize()
            if persistent:
                assert stats.launches == (1 if B <= 8192 else 2), (jvp, B, stats.launches)
            assert torch.equal(u0, keep) and bool(torch.isfinite(out).all())
            inpl = u0.clone()
            _lib.check(l.cnf_solve_tsit5(h, 1, inpl.data_ptr(), eps.data_ptr(), inpl.data_ptr(), B, C.byref(opts), C.byref(stats), sp), h)
            torch.cuda.synchronize()
            assert torch.equal(inpl, out), (jvp, B, float((inpl - out).abs().max()))
            # the same columns through the generic kernel's streamed solve
            ig = make_icnf(cnf, cfg, jvp=jvp, kernel="generic", sol_kwargs=dict(adaptive=False, dt=1 / 8))
            ig.set_params(flat)
            sub = 96
            og = torch.empty(sub, D, device=u0.device)
            _lib.check(l.cnf_solve_tsit5(ig.handle(), 1, u0[:sub].contiguous().data_ptr(), eps[:sub].contiguous().data_ptr(),
                                         og.data_ptr(), sub, C.byref(opts), C.byref(stats), sp), ig.handle())
            torch.cuda.synchronize()
            assert torch.allclose(out[:sub], og, rtol=2e-5, atol=2e-5), (jvp, B, float((out[:sub] - og).abs().max()))


def test_one_launch_solve_takes_the_headline_shape_and_agrees_with_the_streamed_launches():
    """k_solve3b: the adaptive solve of the headline shape (VJP, |eps^T J| row) is ONE launch -- one 32-column tile per
    workgroup up to 8192 columns, several tiles per workgroup beyond (the state then lives in the integrator's buffers);
    CNF_PERSISTENT=0 streams step launches.  Same arithmetic, separately compiled: agreement to the solver tolerance,
    identical step counts on this well-conditioned case; fixed steps, ragged tiles, backward time and maxiters go through
    the same launch."""
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(77)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    persistent = os.environ.get("CNF_PERSISTENT") != "0" and os.environ.get("CNF_STEP_FP32") != "1" \
        and os.environ.get("CNF_STEP_V1") != "1"
    res = {}
    for B in (8192, 1000, 8224):
        xs, eps = _dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))
        for name, kw in (("adaptive", dict(reltol=3.45e-4, abstol=1.19e-7)), ("fixed", dict(adaptive=False, dt=1 / 8))):
            ic = make_icnf(cnf, cfg, sol_kwargs=kw)
            logpx, regs = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
            stt = ic.last_stats
            one = persistent                  # (B = 8224: two tiles per workgroup of the same launch)
            assert (stt["launches"] <= 3) == one, (B, name, stt)
            assert stt["nf"] == (2 if name == "adaptive" else 1) + 6 * (stt["naccept"] + stt["nreject"])
            # the columns do not interact: a sub-batch through the other driver gives the same columns (to the tolerance
            # of the solve when the step sequences differ: the error norm is over the batch)
            sub = slice(0, 512)
            lp2, _ = cnf.inference(make_icnf(cnf, cfg, sol_kwargs=kw), cnf.TrainMode(), xs[:, sub].contiguous(), flat, {},
                                   eps=eps[:, sub].contiguous())
            tol = 2e-3 if name == "adaptive" else 2e-5
            assert torch.allclose(logpx[sub], lp2, rtol=tol, atol=tol), (B, name, float((logpx[sub] - lp2).abs().max()))
            res[(B, name)] = logpx
    # the JVP compute mode and handles without the |eps^T J| row (FFJORD) take k_solve3jb; same agreement with the
    # C oracle as the streamed launches (they share the bar of test_adaptive_solve_vs_oracles)
    cfg4, _, _ = O.baseline_cfg(4)
    for c, jvp in ((cfg, True), (cfg4, False)):
        B = 2048
        xs, eps = _dev(rng.standard_normal((c.nvars, B))), _dev(rng.standard_normal((c.n_in, B)))
        fl = O.glorot_params(c.net, rng, np.float32, 0.1)
        ic = make_icnf(cnf, c, jvp=jvp, sol_kwargs=dict(adaptive=False, dt=1 / 8))
        logpx, regs = cnf.inference(ic, cnf.TrainMode(), xs, fl, {}, eps=eps)
        assert (ic.last_stats["launches"] <= 3) == persistent, (jvp, ic.last_stats)
        sub = slice(100, 164)
        lp2, _ = cnf.inference(make_icnf(cnf, c, jvp=jvp, kernel="generic", sol_kwargs=dict(adaptive=False, dt=1 / 8)),
                               cnf.TrainMode(), xs[:, sub].contiguous(), fl, {}, eps=eps[:, sub].contiguous())
        assert torch.allclose(logpx[sub], lp2, rtol=2e-5, atol=2e-5), (jvp, float((logpx[sub] - lp2).abs().max()))
    # cnf_solve_kernel_time: the kernel's own clock, summed on the device while enabled
    import ctypes as C
    ic = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=1 / 8))
    xs, eps = _dev(rng.standard_normal((cfg.nvars, 512))), _dev(rng.standard_normal((cfg.n_in, 512)))
    cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    l, hh = _lib.lib(), ic.handle()
    _lib.check(l.cnf_solve_kernel_time(hh, 1, None, None), hh)
    for _ in range(3):
        cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    us, k = C.c_float(), C.c_int()
    _lib.check(l.cnf_solve_kernel_time(hh, 0, C.byref(us), C.byref(k)), hh)
    assert k.value == (3 if persistent else 0) and (20.0 < us.value < 5000.0 if persistent else us.value == 0.0), (k.value, us.value)
    cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    _lib.check(l.cnf_solve_kernel_time(hh, 0, C.byref(us), C.byref(k)), hh)
    assert k.value == 0                                    # switched off: nothing is counted
    # maxiters inside the launch
    ic = make_icnf(cnf, cfg, sol_kwargs=dict(reltol=3.45e-4, abstol=1.19e-7, maxiters=3))
    xs, eps = _dev(rng.standard_normal((cfg.nvars, 256))), _dev(rng.standard_normal((cfg.n_in, 256)))
    with pytest.raises(_lib.CNFError):
        cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    # and the handle still works afterwards
    ic2 = make_icnf(cnf, cfg, sol_kwargs=dict(reltol=3.45e-4, abstol=1.19e-7))
    lp, _ = cnf.inference(ic2, cnf.TrainMode(), xs, flat, {}, eps=eps)
    assert torch.isfinite(lp).all()


def test_loss_grad_larger_batches_and_backward_time():
    """The 2- and 4-samples-per-workgroup variants of the pullback kernel (B >= 512, B >= 2048), the
    K-split of the weight-gradient GEMM, and a solve in reverse time."""
    cfg, _, _ = O.baseline_cfg(2)
    for B, tspan in ((600, (0.0, 1.0)), (2100, (0.0, 0.5)), (70, (1.0, 0.0))):
        cfg.tspan = tspan
        val, grad, rval, rgrad, _, _ = _grad_case(cfg, B, 320 + B, "auto", dict(adaptive=False, dt=1 / 4),
                                                  dict(adaptive=False, dt=1 / 4))
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
        _assert_grad(grad, rgrad, f"B={B} tspan={tspan}")


def test_loss_grad_is_reproducible_and_descends():
    """Bit-reproducible (no atomics), and a small step along -grad lowers the loss."""
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(9)
    B = 256
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs, eps = _dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))
    icnf = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=1 / 8))
    l0, g0 = cnf.loss_and_grad(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
    l1, g1 = cnf.loss_and_grad(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
    assert l0 == l1 and torch.equal(g0, g1)
    assert abs(l0 - cnf.loss(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)) <= 1e-6 * abs(l0)
    g = g0.cpu().numpy()
    step = 1e-2 / np.linalg.norm(g)
    l2 = cnf.loss(icnf, cnf.TrainMode(), xs, (flat - step * g).astype(np.float32), {}, eps=eps)
    assert l2 < l0 and abs((l0 - l2) - step * np.dot(g, g)) <= 0.05 * step * np.dot(g, g)
    # (the TestMode loss has a gradient on every network as well: test_testmode_loss_gradient_beyond_the_wave_kernels)
    lt, gt = cnf.loss_and_grad(icnf, cnf.TestMode(), xs, flat, {})
    assert np.isfinite(lt) and torch.isfinite(gt).all() and float(gt.abs().max()) > 0


def test_fit_transform_front_end():
    """The training loop of the reference's MLJ adapter (src/exts/mlj_ext/core_icnf.jl:31-123) on the
    README model without augmentation: a few epochs lower the mean batch loss towards the entropy of the
    data distribution, `transform` returns TestMode log-densities, parameters survive a file round trip."""
    import tempfile
    nn = cnf.Chain(cnf.Dense(1, 3, "tanh"), cnf.Dense(3, 1, "tanh"))
    e32 = float(np.finfo(np.float32).eps)
    icnf = cnf.construct(cnf.RNODE, nn, 1, 0, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0), steer_rate=0.1,
                         lambda1=1e-2, lambda2=1e-2, lambda3=1e-2, sol_kwargs=dict(reltol=float(np.sqrt(e32)), abstol=e32),
                         rng=1)
    r = np.random.default_rng(1).beta(2.0, 4.0, size=(256, 1)).astype(np.float32)
    model = cnf.ICNFModel(icnf, optimizers=(cnf.Lion(eta=1e-2),), n_epochs=12, batch_size=32)
    fitresult, cache, report = cnf.fit(model, 0, r)
    losses = report["losses"]
    assert report["stats"]["iterations"] == 12 * 8 and cache is None
    assert np.mean(losses[-8:]) < np.mean(losses[:8]) - 1.0          # 4.x at initialisation
    assert np.mean(losses[-8:]) > -0.6                                # h(Beta(2,4)) = -0.362 bounds the NLL below
    logpx = cnf.transform(model, fitresult, r)
    assert logpx.shape == (256,) and np.isfinite(logpx).all()
    assert abs(-logpx.mean() - np.mean(losses[-8:])) < 0.5            # exact-trace NLL ~ the training loss
    ps = cnf.fitted_params(model, fitresult)["learned_parameters"]
    with tempfile.TemporaryDirectory() as d:
        cnf.save_params(os.path.join(d, "p.cnfp"), icnf, ps)
        assert np.array_equal(cnf.load_params(os.path.join(d, "p.cnfp"), icnf), ps)


def test_fit_reaches_the_closed_form_optimum_of_a_linear_field():
    """a fit whose optimum is KNOWN.  Data x ~ N(mu, s^2 I_2); field f(z) = W2 (W1 z + b1) + b2
    (identity activations: an affine flow); the maximum-likelihood flow carries the data to N(0, I), so the optimal NLL is the
    data's entropy, n/2 log(2 pi e s^2), and the fitted density is the Gaussian itself.  `mlj.fit` (the reference's loop,
    src/exts/mlj_ext/core_icnf.jl:31-94: shuffled mini-batches of 32, `loss` in TrainMode with fresh Hutchinson probes per
    call, the device adjoint for the gradient) must get there -- which validates loss, gradient, probe draws and the loop
    TOGETHER against something none of them was derived from.  Asserted: exact-trace NLL on the sample within 0.03 of the
    true density's NLL on it, and the reference's three regression distances (test/regression_tests.jl:42-48) of
    pdf(ICNFDist(TestMode)) against the true pdf each <= 0.1."""
    from continuousnf.jl_amd import mlj
    from scipy import stats
    nvars, n, mu, sg = 2, 1024, 0.5, 0.5
    rng = np.random.default_rng(5)
    r = (mu + sg * rng.standard_normal((nvars, n))).astype(np.float32)
    nn = cnf.Chain(cnf.Dense(nvars, 3 * nvars, "identity"), cnf.Dense(3 * nvars, nvars, "identity"))
    icnf = cnf.construct(cnf.FFJORD, nn, nvars, 0, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 1.0), rng=5)
    # (two optimisers one after the other, as the reference's `optimizers` tuple allows: core_icnf.jl:64-73)
    model = mlj.ICNFModel(icnf, optimizers=(mlj.Adam(eta=1e-2), mlj.Adam(eta=1e-3)), n_epochs=40, batch_size=32)
    fitresult, _, report = mlj.fit(model, 0, r.T)
    ps, st = fitresult
    d = cnf.ICNFDist(icnf, cnf.TestMode(), ps, st)
    est_lp = np.asarray(cnf.logpdf(d, r)).reshape(-1).astype(np.float64)
    act_lp = stats.norm(mu, sg).logpdf(r.astype(np.float64)).sum(0)
    entropy = nvars * float(stats.norm(mu, sg).entropy())
    nll, nll_true = float(-est_lp.mean()), float(-act_lp.mean())
    est, act = np.exp(est_lp), np.exp(act_lp)
    mad, msd, tv = float(np.mean(np.abs(est - act))), float(np.mean((est - act) ** 2)), float(np.sum(np.abs(est - act)) / 2 / n)
    helpers.note(f"linear field on N({mu}, {sg}^2 I_2): fitted exact-trace NLL {nll:.4f}, true density's NLL on the sample {nll_true:.4f} "
                 f"(entropy {entropy:.4f}); mad {mad:.4f} msd {msd:.4f} tv {tv:.4f}; first / last batch loss "
                 f"{float(np.mean(report['losses'][:32])):.3f} / {float(np.mean(report['losses'][-32:])):.3f}")
    assert float(np.mean(report["losses"][:32])) > nll_true + 0.5                       # it started far from the optimum
    assert abs(nll - nll_true) <= 0.03, (nll, nll_true, entropy)
    assert mad <= 0.1 and msd <= 0.1 and tv <= 0.1, (mad, msd, tv)
    # the flow it found is the whitening map: z(t1) of the data is standard normal
    prob = cnf.inference_prob(icnf, cnf.TestMode(), r, ps, st)
    zT = np.asarray(cnf.base_sol(icnf, prob).view())[:nvars]
    assert np.abs(zT.mean(1)).max() <= 0.15 and np.abs(np.cov(zT) - np.eye(nvars)).max() <= 0.2, (zT.mean(1), np.cov(zT))
    icnf.close()


def test_fit_readme_example_without_augmentation_reaches_the_entropy_bound():
    """VERDICT round 3, item 1(a), second target: the README example (README.md:31-110) with its commented-out
    `n_in = nvars` line -- RNODE, nvars = 1, Dense(1 => 3, tanh), Dense(3 => 1, tanh), tspan (0, 13), steer_rate 0.1,
    1024 draws of Beta(2, 4).  Without augmentation pdf(ICNFDist) is a normalised density, so the exact-trace NLL cannot go
    below the sample's true NLL (~ the entropy, -0.362) and a working trainer gets close to it.  Lion at the reference's
    default step and then at a tenth of it (`optimizers` is a tuple in the reference too).  Asserted: NLL gap <= 0.04;
    msd and tv <= 0.1 (test/regression_tests.jl:47-48); mad -- which sits at 0.10-0.13 for this 10-parameter field in an
    independent float64 implementation as well (tools/train_lab.py, profiles/round4_training_ablation.md) -- <= 0.2, reported."""
    from continuousnf.jl_amd import mlj
    from scipy import stats
    n = 1024
    rng = np.random.default_rng(1)
    r = rng.beta(2.0, 4.0, size=(1, n)).astype(np.float32)
    nn = cnf.Chain(cnf.Dense(1, 3, "tanh"), cnf.Dense(3, 1, "tanh"))
    icnf = cnf.construct(cnf.RNODE, nn, 1, 0, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0), steer_rate=0.1, rng=1)
    model = mlj.ICNFModel(icnf, optimizers=(mlj.Lion(eta=1e-3), mlj.Lion(eta=1e-4)), n_epochs=100, batch_size=32)
    fitresult, _, report = mlj.fit(model, 0, r.T)
    ps, st = fitresult
    d = cnf.ICNFDist(icnf, cnf.TestMode(), ps, st)
    est_lp = np.asarray(cnf.logpdf(d, r)).reshape(-1).astype(np.float64)
    act_lp = stats.beta(2.0, 4.0).logpdf(r.astype(np.float64)).sum(0)
    nll, nll_true = float(-est_lp.mean()), float(-act_lp.mean())
    est, act = np.exp(est_lp), np.exp(act_lp)
    mad, msd, tv = float(np.mean(np.abs(est - act))), float(np.mean((est - act) ** 2)), float(np.sum(np.abs(est - act)) / 2 / n)
    helpers.note(f"README example, nvars = 1 without augmentation: fitted exact-trace NLL {nll:.4f}, true density's NLL on the sample "
                 f"{nll_true:.4f}; mad {mad:.4f} msd {msd:.4f} tv {tv:.4f}; {report['stats']['iterations']} gradient steps in "
                 f"{report['stats']['time']:.1f} s")
    assert nll >= nll_true - 0.02                                         # a normalised density cannot beat the truth by more than sampling noise
    assert nll - nll_true <= 0.04, (nll, nll_true)
    assert msd <= 0.1 and tv <= 0.1 and mad <= 0.2, (mad, msd, tv)
    icnf.close()


def test_default_fit_trains_the_unaugmented_regression_configuration():
    """`ICNFModel` defaults -- Optimisers.Lion as Optimisers.jl states its rule (the state refreshed
    first), eta = 1e-3, batch 32 (src/exts/mlj_ext/core_icnf.jl:14-28) -- must TRAIN the reference's regression model without
    augmentation (test/regression_tests.jl:1-30 with naugs = 0: RNODE nvars = 8, Dense(8 => 24, tanh), Dense(24 => 8, tanh), tspan
    (0, 13), steer 0.1, 8 x 1024 draws of Beta(2, 4)).  The fitted pdf is a normalised density, so its exact-trace NLL on the sample
    cannot go below the true density's (-2.92) and a working default gets well below the starting point (+11.6, the standard
    normal's NLL of the data) within 60 of the 300 epochs; the paper's Lion rule -- last round's default -- ends ABOVE +5 on
    this configuration after all 300 (profiles/round4_training_ablation.md)."""
    from continuousnf.jl_amd import mlj
    from scipy import stats
    nvars, n = 8, 1024
    rng = np.random.default_rng(1)
    r = rng.beta(2.0, 4.0, size=(nvars, n)).astype(np.float32)
    nn = cnf.Chain(cnf.Dense(nvars, 3 * nvars, "tanh"), cnf.Dense(3 * nvars, nvars, "tanh"))
    icnf = cnf.construct(cnf.RNODE, nn, nvars, 0, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0), steer_rate=0.1, rng=1)
    model = mlj.ICNFModel(icnf, n_epochs=60)                              # every other field at its default
    assert model.optimizers[0].rule == "optimisers" and model.batch_size == 32
    fitresult, _, report = mlj.fit(model, 0, r.T)
    ps, st = fitresult
    d = cnf.ICNFDist(icnf, cnf.TestMode(), ps, st)
    est_lp = np.asarray(cnf.logpdf(d, r)).reshape(-1).astype(np.float64)
    act_lp = stats.beta(2.0, 4.0).logpdf(r.astype(np.float64)).sum(0)
    nll, nll_true = float(-est_lp.mean()), float(-act_lp.mean())
    helpers.note(f"default fit (Lion, Optimisers.jl rule), regression model 8 + 0, 60 epochs: exact-trace NLL {nll:.3f}, true density's "
                 f"{nll_true:.3f}; {report['stats']['iterations']} gradient steps in {report['stats']['time']:.1f} s, "
                 f"pipelined = {report['stats']['pipelined']}")
    assert np.isfinite(report["losses"]).all()
    assert nll >= nll_true - 0.05                                         # a normalised density cannot beat the truth
    assert nll <= 0.0, (nll, nll_true)                                    # ... and the default rule has trained it
    icnf.close()


def test_instability_config_of_the_reference():
    """test/instability_tests.jl:9-45: RNODE 8 + 8, one Dense(16 => 16, tanh), tspan (0, 13), steer_rate 0.1, lambda3 1e-2, 64
    columns of rand(Float32): `loss(icnf, TrainMode(), r, ps, st)` at the package's default solver tolerances (the call
    the reference type-checks with JET).  Finite here, and equal to the float64 oracle's loss at the solver tolerance when
    the probes and the steered end time are pinned."""
    nvars = naugs = 8
    n_in, n = nvars + naugs, 64
    nn = cnf.Chain(cnf.Dense(n_in, n_in, "tanh"))
    icnf = cnf.construct(cnf.RNODE, nn, nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0),
                         steer_rate=1e-1, lambda3=1e-2, rng=3)
    ps, st = cnf.setup(icnf.rng, icnf.nn)
    r = np.random.default_rng(3).random((nvars, n), dtype=np.float32)
    L = cnf.loss(icnf, cnf.TrainMode(), r, ps, st)
    assert np.isfinite(L)
    icnf.steer_rate = 0.0
    eps = np.random.default_rng(4).standard_normal((n_in, n)).astype(np.float32)
    L2 = cnf.loss(icnf, cnf.TrainMode(), r, ps, st, eps=eps)
    net = O.Net((n_in, n_in), (O.ACT_TANH,))
    cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, 1e-2, False, tspan=(0.0, 13.0))
    _, lp, regs, _ = O.inference(cfg, ps.astype(np.float64), r.astype(np.float64), eps.astype(np.float64), True,
                                 reltol=1e-9, abstol=1e-9)
    ref = O.loss(cfg, lp, regs, True)
    assert abs(L2 - ref) <= 5e-3 * max(1.0, abs(ref)), (L2, ref)
    icnf.close()


def test_fit_matrix_of_the_reference():
    """The loops of test/fit_tests.jl at its own sizes (nvars = 2, ndata = 4, n_epochs = 2, one Dense layer): model types x
    augmentation and steering x in-place flag x compute modes; machine -> fit! -> transform -> fitted_params, and the
    distributions built from the fitted machine in both modes.  The reference asserts `!isnothing`; here everything is
    finite as well and the parameters have moved."""
    nvars, ndata, n_epochs = 2, 4, 2
    rng = np.random.default_rng(77)
    for mt in (cnf.RNODE, cnf.FFJORD, cnf.Planar, cnf.CondRNODE, cnf.CondFFJORD, cnf.CondPlanar):
        cond = mt in (cnf.CondRNODE, cnf.CondFFJORD, cnf.CondPlanar)
        for aug_steer in (False, True):
            naugs = nvars if aug_steer else 0
            n_in = nvars + naugs
            if mt in (cnf.Planar, cnf.CondPlanar):                                      # fit_tests.jl:89-131
                nn = cnf.Chain(cnf.PlanarLayer(n_in, "tanh", n_cond=nvars if cond else 0))
            else:
                nn = cnf.Chain(cnf.Dense(n_in + (nvars if cond else 0), n_in, "tanh"))
            df = rng.beta(2.0, 4.0, size=(ndata, nvars)).astype(np.float32)             # rows = observations (DataFrame)
            df2 = rng.beta(4.0, 2.0, size=(ndata, nvars)).astype(np.float32)
            for inplace in (False, True):
                for cm in (cnf.HIPVecJacMatrixMode(), cnf.HIPJacVecMatrixMode()):
                    kw = dict(steer_rate=1e-1, lambda3=1e-2) if aug_steer else {}
                    icnf = cnf.construct(mt, nn, nvars, naugs, compute_mode=cm, inplace=inplace, rng=5, **kw)
                    # (the reference's positional `loss` and its `adtype` keyword are accepted: core_icnf.jl:14-28)
                    model = (cnf.CondICNFModel if cond else cnf.ICNFModel)(icnf, cnf.loss, n_epochs=n_epochs, adtype="AutoEnzyme")
                    mach = cnf.machine(model, (df, df2) if cond else df)
                    ps0, _ = cnf.setup(5, icnf.nn)
                    assert cnf.fit_(mach) is mach and mach.report["stats"]["iterations"] == n_epochs
                    lp = cnf.transform(mach, (df, df2) if cond else df)
                    assert lp.shape == (ndata,) and np.isfinite(lp).all()
                    fp = cnf.fitted_params(mach)
                    assert np.isfinite(fp["learned_parameters"]).all() and fp["learned_parameters"].shape == ps0.shape
                    assert np.abs(fp["learned_parameters"] - ps0).max() > 0        # Lion moved every parameter it saw a sign for
                    for mode in (cnf.TrainMode(), cnf.TestMode()):
                        d = cnf.CondICNFDist(mach, mode, df2.T) if cond else cnf.ICNFDist(mach, mode)
                        assert np.isfinite(cnf.logpdf(d, df.T)).all()
                    icnf.close()


def test_loss_grad_ragged_contraction_after_a_larger_batch():
    """The weight-gradient contraction reads whole 32-row chunks of the factor arrays; rows past the end of a K-split must
    count as zeros whatever an earlier, larger call left in the arena (k_wgrad_wave: the row offset is part of the
    range-checked buffer offset).  One handle: a large batch first, then ragged ones (K = 6 x steps x B not a multiple
    of 32, several contractions per call) against the float64 oracle."""
    from oracle import cnf_grad_oracle as G
    for cfg, Bbig, Bs in ((O.baseline_cfg(3)[0], 1100, (17, 70, 333)), (O.baseline_cfg(2)[0], 2100, (17, 45))):
        cfg.tspan = (0.0, 0.5)
        rng = np.random.default_rng(77)
        flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
        icnf = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 10))       # 5 steps
        for B in (Bbig,) + tuple(Bs):
            xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32) * 3.0       # (large values first: a stale row would show)
            eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
            val, grad = cnf.loss_and_grad(icnf, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
            if B == Bbig:
                continue
            rval, rgrad, _ = G.loss_and_grad(cfg, flat.astype(np.float64), xs.astype(np.float64), eps.astype(np.float64), None,
                                             adaptive=False, dt=1 / 10)
            assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
            _assert_grad(grad.cpu().numpy(), rgrad, f"ragged contraction B={B} after B={Bbig}")
        icnf.close()


def test_loss_grad_tiny_batches():
    """One sample and 17 samples (one full + one nearly empty MFMA column tile)."""
    cfg, _, _ = O.baseline_cfg(3)
    cfg.tspan = (0.0, 0.5)
    for B in (1, 17):
        for kernel in KERNELS:
            val, grad, rval, rgrad, _, _ = _grad_case(cfg, B, 400 + B, kernel, dict(adaptive=False, dt=1 / 4),
                                                      dict(adaptive=False, dt=1 / 4))
            assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
            _assert_grad(grad, rgrad, f"B={B} {kernel}")


def test_loss_grad_wave_local_small_networks(monkeypatch):
    """loss_and_grad of small two-layer tanh networks (the README / regression networks, n_in <= 16, up to 64 hidden units) at
    small batches: solve, loss sums and the whole discrete adjoint in ONE launch, one wave per 16 samples (k_solve_wave<GRAD>),
    + the sum of the waves' partials.  Against the float64 oracle differentiating the same accepted steps: every hidden tile
    count, widths below the padding, ragged and one-sample batches, the regression span (0, 13), lambda3 with augmentation
    and without; bit-reproducible; and A/B against the streamed gradient path (CNF_WAVE_GRAD=0 is read once per process, so
    the A/B runs on the GENERIC kernel choice, which never takes the wave path)."""
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    cases = [
        (O.Cfg(O.Net((16, 48, 16), (O.ACT_TANH,) * 2), 8, 8, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 32, "replay", dict()),   # regression_tests.jl
        (O.Cfg(O.Net((2, 6, 2), (O.ACT_TANH,) * 2), 1, 1, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 32, "replay", dict()),      # README.md:47
        (O.Cfg(O.Net((16, 48, 16), (O.ACT_TANH,) * 2), 8, 8, 1e-2, 1e-2, 1e-2), 77, "replay", dict(tol)),
        (O.Cfg(O.Net((12, 20, 12), (O.ACT_TANH,) * 2), 12, 0, 1e-2, 0.0, 0.0), 5, dict(adaptive=False, dt=1 / 6), dict(adaptive=False, dt=1 / 6)),
        (O.Cfg(O.Net((16, 64, 16), (O.ACT_TANH,) * 2), 10, 6, 0.0, 1e-2, 5e-2), 300, dict(adaptive=False, dt=1 / 5), dict(adaptive=False, dt=1 / 5)),
        (O.Cfg(O.Net((7, 13, 7), (O.ACT_TANH,) * 2), 4, 3, 1e-2, 1e-2, 1e-2, tspan=(1.0, 0.0)), 1, dict(adaptive=False, dt=1 / 4), dict(adaptive=False, dt=1 / 4)),
        (O.Cfg(O.Net((16, 32, 16), (O.ACT_TANH,) * 2), 16, 0, 0.0, 0.0, 0.0), 2048, "replay", dict()),                      # FFJORD, 128 waves
        (O.Cfg(O.Net((16, 48, 16), (O.ACT_TANH,) * 2), 8, 8, 1e-2, 1e-2, 1e-2), 4099, "replay", dict()),                    # config 2's batch + 3: 257 waves, their own partials
        (O.Cfg(O.Net((6, 1, 6), (O.ACT_TANH, O.ACT_IDENTITY)), 6, 0, 1e-2, 1e-2, 0.0), 40, dict(adaptive=False, dt=1 / 5), dict(adaptive=False, dt=1 / 5)),   # a planar flow's MLP form
    ]
    for ci, (cfg, B, ora_kw, sol_kw) in enumerate(cases):
        val, grad, rval, rgrad, st, ost = _grad_case(cfg, B, 900 + ci, "mfma", dict(sol_kw), ora_kw, scale=0.5)
        what = f"wave-local gradient {cfg.net.dims} B={B}"
        assert st["launches"] <= 2, (what, st)
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval)), (what, val, rval)
        _assert_grad(grad, rgrad, what)
        helpers.note(f"{what}: {st['naccept']} steps, {st['launches']} launches, |grad| {np.abs(rgrad).max():.3g}")
    # rejected attempts (a first step of half the span): their stage states, filed in the slot of the step still to be accepted, are
    # overwritten by the attempt that is
    cfg = O.Cfg(O.Net((16, 48, 16), (O.ACT_TANH,) * 2), 8, 8, 1e-2, 1e-2, 1e-2, tspan=(0.0, 6.0))
    val, grad, rval, rgrad, st, _ = _grad_case(cfg, 40, 925, "mfma", dict(dt=3.0, reltol=1e-4, abstol=1e-6), "replay", scale=3.0)
    assert st["nreject"] >= 2 and st["launches"] <= 2, st
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad, rgrad, "wave-local gradient with rejected attempts")
    # conditional models ([z; ys] within the input tile): the conditioning columns of W_1 get their gradient from the same contraction
    for ci, (dims, nvars, naugs, n_cond, B) in enumerate((((6, 18, 6), 4, 2, 5, 40), ((2, 6, 2), 2, 0, 2, 33))):
        cfg = O.Cfg(O.Net(dims, (O.ACT_TANH,) * 2), nvars, naugs, 1e-2, 1e-2, 1e-2 if naugs else 0.0)
        val, grad, rval, rgrad, st, _ = _grad_case(cfg, B, 930 + ci, "mfma", dict(adaptive=False, dt=1 / 6), dict(adaptive=False, dt=1 / 6),
                                                   n_cond=n_cond, scale=0.5)
        assert st["launches"] <= 2, st
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
        _assert_grad(grad, rgrad, f"wave-local gradient, conditional {dims} + {n_cond}")
    # the JVP compute mode (src/icnf.jl:384-420: omega = -c_l eps + c_n J eps / |J eps|, tau = eps), incl. a conditional model
    for ci, (dims, nvars, naugs, n_cond, B, kw) in enumerate((((16, 48, 16), 8, 8, 0, 32, "replay"), ((5, 9, 5), 3, 2, 0, 50, dict(adaptive=False, dt=1 / 6)),
                                                               ((6, 18, 6), 4, 2, 3, 20, dict(adaptive=False, dt=1 / 6)))):
        cfg = O.Cfg(O.Net(dims, (O.ACT_TANH,) * 2), nvars, naugs, 1e-2, 1e-2, 1e-2, tspan=(0.0, 2.0))
        sol_kw = dict() if kw == "replay" else dict(kw)
        val, grad, rval, rgrad, st, _ = _grad_case(cfg, B, 940 + ci, "mfma", sol_kw, kw, jvp=True, n_cond=n_cond, scale=0.5)
        assert st["launches"] <= 2, st
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
        _assert_grad(grad, rgrad, f"wave-local gradient, JVP mode {dims} + {n_cond}")
    # bit-reproducible, and the same gradient as the streamed path to rounding
    cfg = cases[0][0]
    rng = np.random.default_rng(5)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.5)
    xs = rng.standard_normal((cfg.nvars, 32)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, 32)).astype(np.float32)
    out = []
    for kernel in ("mfma", "mfma", "generic"):
        ic = make_icnf(cnf, cfg, kernel=kernel)
        v, g = cnf.loss_and_grad(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
        out.append((v, g.cpu().numpy(), ic.last_stats["launches"]))
        ic.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    assert out[0][2] <= 2 < out[2][2]
    _assert_grad(out[0][1], out[2][1].astype(np.float64), "wave-local vs streamed gradient", rtol=5e-4)


def test_exact_trace_mfma_deep_networks():
    """TestMode for networks with three or more layers runs the MFMA exact-trace kernel
    (cnf_trace.hip: tr J = sum_i [D_L W_L T_{L-1}]_ii with the tangent columns of a group of samples sharing
    every weight fragment).  Checked against the oracle's jacobian_batched: config 3 (3 layers), a 4-layer
    network (tangent ping-pong), mixed activations, a conditional model, ragged batches; then a solve."""
    l = _lib.lib()
    cases = [
        (O.baseline_cfg(3)[0], 0, 77),
        (O.Cfg(O.Net((16, 64, 48, 32, 16), (O.ACT_TANH,) * 4), 12, 4), 0, 50),
        (O.Cfg(O.Net((32, 96, 64, 32), (O.ACT_SOFTPLUS, O.ACT_TANH, O.ACT_IDENTITY)), 32, 0), 0, 33),
        (O.Cfg(O.Net((10, 40, 24, 10), (O.ACT_TANH, O.ACT_SIGMOID, O.ACT_TANH)), 8, 2), 0, 19),   # unaligned widths
        (O.Cfg(O.Net((32, 64, 64, 32), (O.ACT_TANH,) * 3), 32, 0), 5, 40),                          # conditional
        (O.Cfg(O.Net((40, 64, 64, 40), (O.ACT_TANH,) * 3), 40, 0), 0, 21),      # 3 column tiles per sample
        (O.Cfg(O.Net((72, 96, 80, 72), (O.ACT_TANH,) * 3), 60, 12), 0, 18),     # 5 column tiles per sample
        (O.Cfg(O.Net((100, 64, 64, 100), (O.ACT_TANH,) * 3), 100, 0), 0, 17),   # 7 column tiles per sample
        # the shape of the resident-fragment kernel k_trace3 (pads to 32-128-128-32): other activations, a conditional
        # model (28 + 4 input rows), widths below the padding, one sample, a ragged last workgroup
        (O.Cfg(O.Net((32, 128, 128, 32), (O.ACT_SOFTPLUS, O.ACT_SIGMOID, O.ACT_TANH)), 32, 0), 0, 45),
        (O.Cfg(O.Net((28, 128, 128, 28), (O.ACT_TANH,) * 3), 28, 0), 4, 35),
        (O.Cfg(O.Net((30, 120, 116, 30), (O.ACT_TANH, O.ACT_ELU, O.ACT_TANH)), 20, 10), 0, 1),
        (O.baseline_cfg(3)[0], 0, 1000),
    ]
    for k, (cfg, n_cond, B) in enumerate(cases):
        rng = np.random.default_rng(500 + k)
        net = cfg.net if not n_cond else O.Net((cfg.net.dims[0] + n_cond,) + tuple(cfg.net.dims[1:]), cfg.net.acts)
        flat = O.glorot_params(net, rng, np.float32, 0.2)
        u = rng.standard_normal((cfg.n_in + 1, B)).astype(np.float32)
        ys = rng.standard_normal((n_cond, B)).astype(np.float32) if n_cond else None
        layers = [cnf.Dense(a, b, helpers.ACT_NAME[c]) for a, b, c in zip(net.dims[:-1], net.dims[1:], net.acts)]
        tag = cnf.CondRNODE if n_cond else cnf.RNODE
        icnf = cnf.construct(tag, cnf.Chain(*layers), cfg.nvars, cfg.naugs, compute_mode=cnf.HIPVecJacMatrixMode("mfma"))
        nn = cnf.CondLayer(icnf.nn, _dev(ys)) if n_cond else icnf.nn
        if n_cond:
            icnf.set_params(flat)
            icnf.set_cond(_dev(ys), B)
        assert l.cnf_kernel_for(icnf.handle(), _lib.MODE_TEST, B) == _lib.KERNEL_MFMA, k
        du = cnf.augmented_f(_dev(u), flat, 0.0, icnf, cnf.TestMode(), nn, {}, None).cpu().numpy()
        c64 = O.Cfg(net, cfg.nvars, cfg.naugs)
        ref = O.augmented_f_test(net, flat.astype(np.float64), u.astype(np.float64), False,
                                 None if ys is None else ys.astype(np.float64))
        assert_parity(du, ref, f"exact trace case {k}", trace_row=cfg.n_in)
    # a full adaptive TestMode inference on config 3 (what pdf(ICNFDist(TestMode)) runs) vs the float64 oracle
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(9)
    B = 200
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    kw = dict(reltol=1e-5, abstol=1e-6)
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=kw)
    logpx, _ = cnf.inference(ic, cnf.TestMode(), _dev(xs), flat, {})
    assert ic.last_stats["kernel_used"] == _lib.KERNEL_MFMA
    _, ref_lp, _, _ = O.inference(cfg, flat.astype(np.float64), xs.astype(np.float64), None, False, reltol=1e-9, abstol=1e-9)
    assert_parity(logpx.cpu().numpy(), ref_lp, "config 3 TestMode logpx", rtol=2e-4)


def test_jvp_mode_headline_shape_step_kernel():
    """JVP compute mode on the headline network runs the fused step kernel k_step3j (one forward sweep of state and
    tangent columns per evaluation): fixed-dt and adaptive solves against the oracles, ragged and multi-tile batches,
    RNODE (all three scalar rows) and FFJORD (no norm rows)."""
    # (B = 8224: 257 tiles over 129 workgroups of the one-launch solve -- two tiles per workgroup, one workgroup with one)
    for i, B in ((3, 77), (3, 1000), (4, 40), (3, 8224)):
        cfg, _, _ = O.baseline_cfg(i)
        cfg.use_jvp = True
        rng = np.random.default_rng(720 + B)
        flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
        xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
        eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
        ic = make_icnf(cnf, cfg, jvp=True, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
        prob = cnf.inference_prob(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
        fsol = cnf.base_sol(ic, prob).view().cpu().numpy()
        assert prob.stats["kernel_used"] == _lib.KERNEL_MFMA and prob.stats["launches"] <= 8 + 4
        if _one_launch_expected():
            assert prob.stats["launches"] <= 3, prob.stats            # also beyond one tile per workgroup
        u0 = O.inference_u0(cfg, xs, True)
        ref, _ = O.tsit5_solve(cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), True), u0.astype(np.float64),
                               *cfg.tspan, dt=1 / 8, adaptive=False)
        assert_parity(fsol, ref, f"cfg{i} JVP fixed-dt fsol B={B}", trace_row=cfg.n_in)
        kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
        ia = make_icnf(cnf, cfg, jvp=True, kernel="mfma", sol_kwargs=kw)
        pa = cnf.inference_prob(ia, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
        fa = cnf.base_sol(ia, pa).view().cpu().numpy()
        cref, cst = CO.solve(cfg, flat, u0, eps, True, **kw)
        assert abs(pa.stats["naccept"] - cst["naccept"]) <= 2, (pa.stats, cst)
        assert_parity(fa, cref, f"cfg{i} JVP adaptive vs C oracle B={B}", rtol=5e-3, trace_row=cfg.n_in)


def test_jvp_mode_large_network_on_mfma():
    """TrainMode with the JVP compute mode (src/icnf.jl:384-420) on config 5's 128-384-128 network, whose
    tangent images do not fit beside the weights in the fused step kernel: the forward sweep with the
    tangents as a second column tile (cnf_trace.hip: k_jvp_mfma) behind the generic Tsit5 driver."""
    cfg, _, _ = O.baseline_cfg(5)
    B = 150
    rng = np.random.default_rng(700)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    u = rng.standard_normal((cfg.D(True), B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    icnf = make_icnf(cnf, cfg, jvp=True, kernel="mfma")
    assert _lib.lib().cnf_kernel_for(icnf.handle(), _lib.MODE_TRAIN, B) == _lib.KERNEL_MFMA
    du = cnf.augmented_f(_dev(u), flat, 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, _dev(eps)).cpu().numpy()
    ref = O.augmented_f_train(cfg.net, flat.astype(np.float64), u.astype(np.float64), eps.astype(np.float64),
                              cfg.lam1 != 0, cfg.lam2 != 0, use_jvp=True)
    assert_parity(du, ref, "cfg5 JVP RHS", trace_row=cfg.n_in)
    # and through a fixed-dt solve
    ic = make_icnf(cnf, cfg, jvp=True, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    logpx, regs = cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
    assert ic.last_stats["kernel_used"] == _lib.KERNEL_MFMA
    c64 = O.Cfg(cfg.net, cfg.nvars, cfg.naugs, cfg.lam1, cfg.lam2, cfg.lam3, use_jvp=True)
    _, ref_lp, ref_regs, _ = O.inference(c64, flat.astype(np.float64), xs.astype(np.float64), eps.astype(np.float64),
                                         True, dt=1 / 8, adaptive=False)
    assert_parity(logpx.cpu().numpy(), ref_lp, "cfg5 JVP logpx")
    assert_parity(regs[1].cpu().numpy(), ref_regs[1], "cfg5 JVP n-dot integral")


def test_loss_grad_config5_network():
    """The gradient on config 5's 128-384-128 network (weights streamed from L2 in the forward kernel, MFMA
    pullback with ahat sharing the tbar_L slot to fit 160 KB of LDS)."""
    cfg, _, _ = O.baseline_cfg(5)
    cfg.tspan = (0.0, 0.5)
    for kernel in KERNELS:
        val, grad, rval, rgrad, _, _ = _grad_case(cfg, 40, 810, kernel, dict(adaptive=False, dt=1 / 4),
                                                  dict(adaptive=False, dt=1 / 4), scale=0.1)
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
        _assert_grad(grad, rgrad, f"config 5 {kernel}")


@pytest.mark.parametrize("which", ["headline", "config5", "conditional-3-layer", "headline-jvp", "config5-jvp", "conditional-3-layer-jvp"])
def test_loss_grad_one_launch_and_two_launch_pullbacks_agree(which):
    """The pullback kernels' two forms (cnf_set_grad_split: one launch per run of steps, or the adjoint-independent sweeps of all
    stages side by side + the hbar chains in turn) against the float64 adjoint and against each other -- k_adj3b at the
    headline shape, k_adj_mfma at config 5's network and on a conditional three-layer model, ragged batches, adaptive steps."""
    l = _lib.lib()
    jvp = which.endswith("-jvp")               # the JVP compute mode (src/icnf.jl:384-456): k_adj_mfma<.., JM> on every shape
    which = which[:-4] if jvp else which
    if which == "headline":
        cfg = O.Cfg(O.Net((32, 128, 128, 32), (O.ACT_TANH,) * 3), 32, 0, 0.01, 0.01)
        n_cond, B, scale = 0, 77, 0.1
    elif which == "config5":
        cfg, _, _ = O.baseline_cfg(5)
        n_cond, B, scale = 0, 40, 0.1
    else:
        cfg = O.Cfg(O.Net((12, 64, 48, 12), (O.ACT_TANH, O.ACT_SOFTPLUS, O.ACT_TANH)), 8, 4, 0.01, 0.01, 0.01)
        n_cond, B, scale = 3, 50, 0.3
    cfg.tspan = (0.0, 0.5)
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    got = {}
    was = l.cnf_set_grad_split(-1)
    try:
        for mode in (0, 1):
            l.cnf_set_grad_split(mode)
            for tag, sol_kw, ora_kw in (("fixed", dict(adaptive=False, dt=1 / 4), dict(adaptive=False, dt=1 / 4)),
                                        ("adaptive", dict(tol), "replay")):
                val, grad, rval, rgrad, st, _ = _grad_case(cfg, B, 870, "mfma", sol_kw, ora_kw, n_cond=n_cond, scale=scale, jvp=jvp)
                assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval)), (which, jvp, mode, tag)
                _assert_grad(grad, rgrad, f"{which} jvp={jvp} split={mode} {tag}")
                got[mode, tag] = (val, grad, st["naccept"])
    finally:
        l.cnf_set_grad_split(was)
    for tag in ("fixed", "adaptive"):
        (v0, g0, n0), (v1, g1, n1) = got[0, tag], got[1, tag]
        assert n0 == n1 and v0 == v1, (which, tag)
        assert np.abs(g0 - g1).max() <= 2e-6 * (np.abs(g0).max() + 1e-30), (which, tag, np.abs(g0 - g1).max(), np.abs(g0).max())


@pytest.mark.parametrize("which", ["headline", "three-layer"])
def test_loss_grad_more_steps_than_one_run_of_the_pullback(which):
    """40 fixed steps: more than the 32 steps one launch (pair) of the pullback kernels takes -- two runs, each with its own
    contraction, in both forms of the pullback (k_adj3b at the headline shape, k_adj_mfma_run elsewhere)."""
    l = _lib.lib()
    if which == "headline":
        cfg = O.Cfg(O.Net((32, 128, 128, 32), (O.ACT_TANH,) * 3), 32, 0, 0.01, 0.01)
        B, scale = 40, 0.1
    else:
        cfg = O.Cfg(O.Net((12, 64, 48, 12), (O.ACT_TANH, O.ACT_TANH, O.ACT_TANH)), 8, 4, 0.01, 0.01, 0.01)
        B, scale = 20, 0.3
    cfg.tspan = (0.0, 0.5)
    was = l.cnf_set_grad_split(-1)
    try:
        for mode in (0, 1):
            l.cnf_set_grad_split(mode)
            val, grad, rval, rgrad, st, _ = _grad_case(cfg, B, 880, "mfma", dict(adaptive=False, dt=0.5 / 40),
                                                       dict(adaptive=False, dt=0.5 / 40), scale=scale)
            assert st["naccept"] == 40
            assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval)), (which, mode)
            _assert_grad(grad, rgrad, f"{which} 40 steps split={mode}")
    finally:
        l.cnf_set_grad_split(was)


@pytest.mark.parametrize("i", [3, 5])
def test_loss_grad_recording_solve_that_gives_up_falls_back_to_step_launches(i):
    """The gradient's forward is a recording ONE-launch solve (k_solve3b<RECORD>, k_solve_bcast<RECORD>) that also assembles u0 and
    forms the loss sums; when its waits run out (here: after one poll) the call records with step launches instead, forms the
    post-processing behind them, and returns the same loss and gradient."""
    if not _one_launch_expected():
        pytest.skip("the one-launch solve is switched off in this process")
    cfg, _, _ = O.baseline_cfg(i)
    cfg.tspan = (0.0, 0.5)
    rng = np.random.default_rng(60 + i)
    B = 300 if i == 3 else 120                                   # several workgroups: a meeting that can run out
    flat = torch.from_numpy(O.glorot_params(cfg.net, rng, np.float32, 0.1)).cuda()
    xs, eps = _dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=tol)
    v0, g0 = cnf.loss_and_grad(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    n0, fb0 = ic.last_stats["launches"], ic.solve_fallbacks()
    g0 = g0.clone()
    ic.set_solve_wait(poll_limit=1)
    v1, g1 = cnf.loss_and_grad(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    assert ic.solve_fallbacks() == fb0 + 1 and ic.last_stats["launches"] > n0, (ic.last_stats, n0)
    ic.set_solve_wait(poll_limit=0x7fffffff)
    v2, g2 = cnf.loss_and_grad(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    assert ic.last_stats["launches"] == n0 and v2 == v0 and torch.equal(g2, g0)
    # (the streamed solve takes the same accepted steps up to the rounding of a different kernel's error norm)
    assert abs(v1 - v0) <= 2e-4 * max(1.0, abs(v0)), (v1, v0)
    _assert_grad(g1.cpu().numpy(), g0.cpu().numpy().astype(np.float64), f"config {i}: streamed fallback vs one-launch recording", rtol=5e-3)
    ic.close()


def test_streamed_weights_run_the_fused_step_kernel_on_16_sample_tiles():
    """Config 5's network (weights streamed from L2) at ragged batches that leave the last 16-sample workgroup
    partly empty: kernel = auto and kernel = mfma are the same fused step kernel (one launch per step, 16-sample
    tiles, fragment stream across sweeps), TrainMode/VJP and TestMode, RHS and a fixed-dt inference against the oracle."""
    cfg, _, _ = O.baseline_cfg(5)
    for B, seed in ((120, 900), (16, 901), (33, 902)):
        rng = np.random.default_rng(seed)
        flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
        xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
        eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
        for train in (True, False):
            mode = cnf.TrainMode() if train else cnf.TestMode()
            u = rng.standard_normal((cfg.D(train), B)).astype(np.float32)
            ref = cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), train)(u.astype(np.float64))
            _, ref_lp, _, _ = O.inference(cfg, flat.astype(np.float64), xs.astype(np.float64), eps.astype(np.float64),
                                          train, dt=1 / 8, adaptive=False)
            launches = {}
            for kernel in ("auto", "mfma"):
                ic = make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=dict(adaptive=False, dt=1 / 8))
                du = cnf.augmented_f(_dev(u), flat, 0.0, ic, mode, ic.nn, {}, _dev(eps)).cpu().numpy()
                assert_parity(du, ref, f"cfg5 RHS B={B} train={train} kernel={kernel}", trace_row=cfg.n_in)
                logpx, _ = cnf.inference(ic, mode, _dev(xs), flat, {}, eps=_dev(eps))
                assert_parity(logpx.cpu().numpy(), ref_lp, f"cfg5 logpx B={B} train={train} kernel={kernel}")
                assert ic.last_stats["kernel_used"] == _lib.KERNEL_MFMA
                launches[kernel] = ic.last_stats["launches"]
            assert launches["auto"] == launches["mfma"] <= 8 + 4, launches      # 8 steps: one launch each (+ k1, copy)


def test_loss_grad_more_steps_than_the_trajectory_store():
    """100 fixed steps: more than the 64 slots the trajectory store starts with -- the recorded forward notices,
    grows the store and solves again; the gradient still matches the oracle."""
    cfg, _, _ = O.baseline_cfg(2)
    cfg.tspan = (0.0, 1.0)
    val, grad, rval, rgrad, st, _ = _grad_case(cfg, 24, 950, "mfma", dict(adaptive=False, dt=1 / 100),
                                               dict(adaptive=False, dt=1 / 100))
    assert st["naccept"] == 100
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad, rgrad, "100 steps")


def test_one_layer_network_of_the_reference_benchmark_suite():
    """benchmark/benchmarks.jl:24-59: RNODE nvars = naugs = 8, nn = Chain(Dense(16 => 16, tanh)) -- ONE layer --, 64 samples,
    tspan (0, 13), lambda3 = 1e-2; its four benchmarks are loss(TrainMode), loss(TestMode) and their gradients.  The wave
    kernels take the network through an appended identity layer (W_2 = I, b_2 = 0: exact): inference in Train and TestMode in
    ONE launch and loss_and_grad(TrainMode) in two, against the float64 oracle of the one-layer network; other one-layer
    widths; the kernel = generic route beside it."""
    f64 = lambda a: a.astype(np.float64)
    one = _one_launch_expected() and os.environ.get("CNF_WAVE") != "0"
    for (n_in, nvars, naugs, B) in ((16, 8, 8, 64), (5, 3, 2, 33), (16, 16, 0, 1)):
        net = O.Net((n_in, n_in), (O.ACT_TANH,))
        rng = np.random.default_rng(70 + n_in + B)
        flat = O.glorot_params(net, rng, np.float32, 0.6)
        flat[-n_in:] = 0.1 * rng.standard_normal(n_in).astype(np.float32)
        xs = rng.random((nvars, B)).astype(np.float32)                     # rand(rng, Float32, nvars, n)  benchmarks.jl:48
        eps = rng.standard_normal((n_in, B)).astype(np.float32)
        lam3 = 1e-2 if naugs else 0.0
        cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, lam3, tspan=(0.0, 13.0))
        outs = {}
        for kernel in ("auto", "generic"):
            ic = make_icnf(cnf, cfg, kernel=kernel, tag=cnf.RNODE, sol_kwargs=dict(adaptive=False, dt=13 / 32))
            lp, (E, n, A) = cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
            if kernel == "auto":
                assert (ic.last_stats["launches"] == 1) == one, ic.last_stats
            _, ref_lp, ref_regs, _ = O.inference(cfg, f64(flat), f64(xs), f64(eps), True, dt=13 / 32, adaptive=False)
            assert_parity(lp.cpu().numpy(), ref_lp, f"one-layer {n_in} TrainMode logpx {kernel}")
            assert_parity(torch.stack([E, n, A]).cpu().numpy(), np.stack(ref_regs), f"one-layer {n_in} TrainMode regs {kernel}")
            lpt, _ = cnf.inference(ic, cnf.TestMode(), _dev(xs), flat, {})
            if kernel == "auto":
                assert (ic.last_stats["launches"] == 1) == one, ic.last_stats
            _, ref_lpt, _, _ = O.inference(cfg, f64(flat), f64(xs), None, False, dt=13 / 32, adaptive=False)
            assert_parity(lpt.cpu().numpy(), ref_lpt, f"one-layer {n_in} TestMode logpx {kernel}")
            outs[kernel] = lp.cpu().numpy()
            ic.close()
        assert_parity(outs["auto"], outs["generic"].astype(np.float64), f"one-layer {n_in}: wave kernel vs generic")
        # the gradient of the TrainMode loss, adaptive at the package's default tolerances, the oracle on the same steps
        val, grad, rval, rgrad, st, ost = _grad_case(cfg, B, 80 + n_in, "auto", dict(), "replay", scale=0.6)
        assert grad.size == net.n_params
        if one:
            assert st["launches"] <= 2, st
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
        _assert_grad(grad, rgrad, f"one-layer {n_in} gradient")


def test_testmode_loss_gradient_small_networks():
    """The other derivative the reference takes: loss(icnf, TestMode(), xs, ps, st) = -mean(logpx) through the EXACT-TRACE solve,
    w.r.t. ps and xs (test/call_tests.jl `diff_loss` / `diff2_loss` with omode = TestMode(); benchmark/benchmarks.jl:60-99
    "AD-1-order" / "test").  cnf_loss_grad_test (k_solve_wave<TEST, GRAD>: the closed-form trace of two-layer networks and its
    pullback in the launch of the solve) against the float64 oracle on the same accepted steps (itself pinned by torch
    autograd, tests/test_grad_oracle.py): the benchmark's one-layer network, the regression / README networks, ragged
    batches, host arrays; networks it does not take raise NotImplementedError."""
    from oracle import cnf_grad_oracle as G
    T2 = (O.ACT_TANH,) * 2
    cases = [
        (O.Cfg(O.Net((16, 16), (O.ACT_TANH,)), 8, 8, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 64, dict()),        # benchmarks.jl:24-59
        (O.Cfg(O.Net((16, 48, 16), T2), 8, 8, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 32, dict()),               # regression_tests.jl
        (O.Cfg(O.Net((2, 6, 2), T2), 1, 1, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 77, dict()),                  # README.md:47
        (O.Cfg(O.Net((12, 20, 12), T2), 12, 0, 0.0, 0.0, 0.0), 5, dict(adaptive=False, dt=1 / 6)),
        (O.Cfg(O.Net((16, 64, 16), T2), 10, 6, 1e-2, 1e-2, 1e-2, tspan=(1.0, 0.0)), 300, dict(adaptive=False, dt=1 / 5)),
        (O.Cfg(O.Net((7, 7), (O.ACT_TANH,)), 4, 3, 1e-2, 1e-2, 1e-2), 1, dict(adaptive=False, dt=1 / 4)),
        (O.Cfg(O.Net((6, 1, 6), (O.ACT_TANH, O.ACT_IDENTITY)), 6, 0, 0.0, 0.0, 0.0), 40, dict(adaptive=False, dt=1 / 5)),   # a planar flow's MLP form
    ]
    for ci, (cfg, B, sol_kw) in enumerate(cases):
        rng = np.random.default_rng(600 + ci)
        flat = O.glorot_params(cfg.net, rng, np.float32, 0.5)
        flat[-cfg.n_in:] = 0.1 * rng.standard_normal(cfg.n_in).astype(np.float32)
        xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
        ic = make_icnf(cnf, cfg, kernel="auto", tag=cnf.RNODE, sol_kwargs=dict(sol_kw))
        host = ci == 3
        val, grad, gx = cnf.loss_and_grad(ic, cnf.TestMode(), xs if host else _dev(xs), flat, {}, with_x=True)
        if not host:
            grad, gx = grad.cpu().numpy(), gx.cpu().numpy()
        st = ic.last_stats
        rval, rgrad, ost = G.loss_and_grad_test(cfg, flat.astype(np.float64), xs.astype(np.float64), dts=[float(d) for d in ic.last_steps])
        what = f"TestMode gradient {cfg.net.dims} B={B}"
        assert st["launches"] <= 2 and st["naccept"] == ost.naccept, (what, st)
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval)), (what, val, rval)
        _assert_grad(grad, rgrad, what)
        _assert_grad(gx, ost.grad_x, what + " d/dxs", rtol=2e-4)
        # the loss value is the one loss() returns (its launch evaluates tanh in the exp2 / rcp form, this one with the
        # polynomial near 0: rounding apart)
        assert abs(val - cnf.loss(ic, cnf.TestMode(), _dev(xs), flat, {})) <= 1e-5 * max(1.0, abs(val))
        ic.close()
    # a conditional model (CondRNODE 4 + 2 with 3 conditioning inputs)
    net = O.Net((9, 18, 6), T2)
    cfg = O.Cfg(O.Net((6, 18, 6), T2), 4, 2, 1e-2, 1e-2, 1e-2)
    rng = np.random.default_rng(77)
    flat = O.glorot_params(net, rng, np.float32, 0.5)
    xs = rng.standard_normal((4, 21)).astype(np.float32)
    ys = rng.standard_normal((3, 21)).astype(np.float32)
    layers = [cnf.Dense(9, 18, "tanh"), cnf.Dense(18, 6, "tanh")]
    ic = cnf.construct(cnf.CondRNODE, cnf.Chain(*layers), 4, 2, compute_mode=cnf.HIPVecJacMatrixMode("auto"), tspan=(0.0, 1.0),
                       lambda3=1e-2, sol_kwargs=dict(adaptive=False, dt=1 / 6))
    val, grad, gx = cnf.loss_and_grad(ic, cnf.TestMode(), _dev(xs), _dev(ys), flat, {}, with_x=True)
    c64 = O.Cfg(net, 4, 2, 1e-2, 1e-2, 1e-2)
    rval, rgrad, ost = G.loss_and_grad_test(c64, flat.astype(np.float64), xs.astype(np.float64), ys.astype(np.float64), adaptive=False, dt=1 / 6)
    assert ic.last_stats["launches"] <= 2
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad.cpu().numpy(), rgrad, "TestMode gradient, conditional")
    _assert_grad(gx.cpu().numpy(), ost.grad_x, "TestMode gradient, conditional, d/dxs", rtol=2e-4)
    ic.close()


def test_testmode_loss_gradient_beyond_the_wave_kernels():
    """VERDICT round 4, item 7 / "missing" 5 (test/call_tests.jl:239-252 differentiates the TestMode loss for EVERY model;
    benchmark/benchmarks.jl:60-99): cnf_loss_grad_test for networks k_solve_wave<TEST, GRAD> does not take -- a recorded exact-trace
    solve and ONE launch of the generic adjoint kernel k_adj_test over all accepted steps (cnf_gradt.hip).  The headline network
    32-128-128-32 at B = 256 (adaptive, README tolerances, and fixed dt), a four-layer mixed-activation chain, config 5's
    128-384-128 (the P / Q matrices in the global scratch), a one-layer network wider than the wave kernels take and a
    conditional three-layer model, against the float64 oracle on the same accepted steps (itself pinned by torch autograd,
    tests/test_grad_oracle.py); the gradient w.r.t. the data from the same sweep; bit-reproducible."""
    from oracle import cnf_grad_oracle as G
    T = O.ACT_TANH
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    cases = [
        (O.baseline_cfg(3)[0], 256, tol, "mfma"),
        (O.baseline_cfg(3)[0], 37, dict(adaptive=False, dt=1 / 4), "auto"),
        (O.Cfg(O.Net((10, 24, 17, 24, 10), (T, O.ACT_SOFTPLUS, O.ACT_SIGMOID, T)), 7, 3, 0.0, 0.0, 1e-2), 300, dict(adaptive=False, dt=1 / 5), "auto"),   # (more samples than workgroups)
        (O.baseline_cfg(5)[0], 6, dict(adaptive=False, dt=1 / 2), "auto"),
        (O.Cfg(O.Net((40, 40), (T,)), 40, 0, 0.0, 0.0, 0.0, tspan=(1.0, 0.0)), 9, dict(adaptive=False, dt=1 / 5), "auto"),
    ]
    for ci, (cfg, B, sol_kw, kernel) in enumerate(cases):
        rng = np.random.default_rng(640 + ci)
        flat = O.glorot_params(cfg.net, rng, np.float32, 0.3)
        flat[-cfg.n_in:] = 0.1 * rng.standard_normal(cfg.n_in).astype(np.float32)
        xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
        ic = make_icnf(cnf, cfg, kernel=kernel, tag=cnf.RNODE, sol_kwargs=dict(sol_kw))
        val, grad, gx = cnf.loss_and_grad(ic, cnf.TestMode(), _dev(xs), flat, {}, with_x=True)
        grad, gx = grad.cpu().numpy(), gx.cpu().numpy()
        st = ic.last_stats
        rval, rgrad, ost = G.loss_and_grad_test(cfg, flat.astype(np.float64), xs.astype(np.float64), dts=[float(d) for d in ic.last_steps])
        what = f"TestMode gradient (k_adj_test) {cfg.net.dims} B={B}"
        assert st["naccept"] == ost.naccept == len(ic.last_steps), (what, st)
        assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval)), (what, val, rval)
        _assert_grad(grad, rgrad, what)
        _assert_grad(gx, ost.grad_x, what + " d/dxs", rtol=2e-4)
        # equal to the loss an inference call returns (its solve runs the fast exact-trace kernels: the solver tolerance apart)
        lv = cnf.loss(ic, cnf.TestMode(), _dev(xs), flat, {})
        assert abs(val - lv) <= (5e-3 if sol_kw.get("adaptive", True) else 1e-5) * max(1.0, abs(val)), (what, val, lv)
        if ci == 1:                                            # the same call again: the same bits
            val2, grad2 = cnf.loss_and_grad(ic, cnf.TestMode(), _dev(xs), flat, {})
            assert val2 == val and np.array_equal(grad2.cpu().numpy(), grad)
        ic.close()
    # a conditional three-layer model (CondRNODE 5 + 2 with 3 conditioning inputs)
    net = O.Net((10, 20, 20, 7), (T,) * 3)
    rng = np.random.default_rng(78)
    flat = O.glorot_params(net, rng, np.float32, 0.4)
    xs = rng.standard_normal((5, 23)).astype(np.float32)
    ys = rng.standard_normal((3, 23)).astype(np.float32)
    layers = [cnf.Dense(10, 20, "tanh"), cnf.Dense(20, 20, "tanh"), cnf.Dense(20, 7, "tanh")]
    ic = cnf.construct(cnf.CondRNODE, cnf.Chain(*layers), 5, 2, compute_mode=cnf.HIPVecJacMatrixMode("auto"), tspan=(0.0, 1.0),
                       lambda3=1e-2, sol_kwargs=dict(adaptive=False, dt=1 / 6))
    val, grad, gx = cnf.loss_and_grad(ic, cnf.TestMode(), _dev(xs), _dev(ys), flat, {}, with_x=True)
    c64 = O.Cfg(net, 5, 2, 1e-2, 1e-2, 1e-2)
    rval, rgrad, ost = G.loss_and_grad_test(c64, flat.astype(np.float64), xs.astype(np.float64), ys.astype(np.float64), adaptive=False, dt=1 / 6)
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad.cpu().numpy(), rgrad, "TestMode gradient (k_adj_test), conditional")
    _assert_grad(gx.cpu().numpy(), ost.grad_x, "TestMode gradient (k_adj_test), conditional, d/dxs", rtol=2e-4)
    ic.close()


def test_submitted_gradients_equal_the_synchronous_ones():
    """cnf_loss_grad_submit / cnf_loss_grad_collect / cnf_set_params_async: gradients enqueued back to back with a parameter
    update between them and nothing waited for -- loss and gradient (device tensors) equal those of the synchronous calls bit
    for bit; TestMode likewise; a fourth submission is refused; networks without an in-launch gradient raise
    NotImplementedError; and `fit` run pipelined ends with the very parameters of the synchronous loop."""
    cfg = O.Cfg(O.Net((16, 48, 16), (O.ACT_TANH,) * 2), 8, 8, 1e-2, 1e-2, 1e-2, tspan=(0.0, 4.0))
    rng = np.random.default_rng(17)
    flat0 = O.glorot_params(cfg.net, rng, np.float32, 0.5)
    batches = [(_dev(rng.standard_normal((8, B))), _dev(rng.standard_normal((16, B)))) for B in (32, 33, 64)]
    for mode in (cnf.TrainMode(), cnf.TestMode()):
        train = isinstance(mode, cnf.TrainMode)
        # synchronous: three gradients with a (sign-SGD) parameter update after each
        ic = make_icnf(cnf, cfg, kernel="mfma")
        ps = torch.from_numpy(flat0.copy()).cuda()
        want = []
        for xs, eps in batches:
            v, g = cnf.loss_and_grad(ic, mode, xs, ps, {}, **(dict(eps=eps) if train else {}))
            want.append((v, g.clone()))
            ps.sub_(torch.sign(g), alpha=1e-3)
        ps_sync = ps.clone()
        ic.close()
        # submitted: the same three, nothing waited for in between
        ic = make_icnf(cnf, cfg, kernel="mfma")
        ps = torch.from_numpy(flat0.copy()).cuda()
        got = []
        for xs, eps in batches:
            lossd, g = cnf.loss_and_grad_submit(ic, mode, xs, ps, {}, **(dict(eps=eps) if train else {}))
            got.append((lossd, g))
            ps.sub_(torch.sign(g), alpha=1e-3)
        with pytest.raises(cnf.CNFError):
            cnf.loss_and_grad_submit(ic, mode, batches[0][0], ps, {}, **(dict(eps=batches[0][1]) if train else {}))
        for _ in range(3):
            st = cnf.loss_and_grad_collect(ic)
            assert st["launches"] == 1 and st["naccept"] > 0
        torch.cuda.synchronize()
        for (v, g), (lossd, gd) in zip(want, got):
            assert float(lossd[0]) == np.float32(v) and torch.equal(g, gd), mode
        assert torch.equal(ps, ps_sync)
        ic.close()
    cfg3, _, _ = O.baseline_cfg(3)
    ic = make_icnf(cnf, cfg3, kernel="mfma")
    f3 = torch.from_numpy(O.glorot_params(cfg3.net, rng, np.float32, 0.1)).cuda()
    with pytest.raises(NotImplementedError):
        cnf.loss_and_grad_submit(ic, cnf.TrainMode(), _dev(np.zeros((cfg3.nvars, 8), np.float32)), f3, {})
    ic.close()
    # a submitted launch that gives up (every wait of its 128 waves runs out after one poll): nothing downstream of it may see
    # garbage -- zeros and a NaN loss are delivered, the collect reports it; the synchronous call falls back to the streamed path
    ic = make_icnf(cnf, cfg, kernel="mfma")
    ps = torch.from_numpy(flat0.copy()).cuda()
    xs, eps = _dev(rng.standard_normal((8, 2048))), _dev(rng.standard_normal((16, 2048)))
    ic.set_solve_wait(poll_limit=1)
    lossd, g = cnf.loss_and_grad_submit(ic, cnf.TrainMode(), xs, ps, {}, eps=eps)
    with pytest.raises(cnf.CNFError):
        cnf.loss_and_grad_collect(ic)
    torch.cuda.synchronize()
    assert torch.isnan(lossd).all() and float(g.abs().max()) == 0.0
    fb = ic.solve_fallbacks()
    v, g2 = cnf.loss_and_grad(ic, cnf.TrainMode(), xs, ps, {}, eps=eps)          # (gives up as well, then the streamed gradient path)
    assert ic.solve_fallbacks() == fb + 1 and ic.last_stats["launches"] > 2
    ic.set_solve_wait(poll_limit=0x7fffffff)
    v3, g3 = cnf.loss_and_grad(ic, cnf.TrainMode(), xs, ps, {}, eps=eps)
    assert ic.last_stats["launches"] <= 2 and abs(v - v3) <= 1e-5 * max(1.0, abs(v3))
    _assert_grad(g2.cpu().numpy(), g3.cpu().numpy().astype(np.float64), "streamed fallback vs in-launch gradient", rtol=5e-4)
    ic.close()
    # fit: pipelined and synchronous loops end with the same parameters and report the same losses
    data = np.random.default_rng(3).beta(2.0, 4.0, size=(256, 2)).astype(np.float32)
    res = []
    for pipelined in (True, False):
        nn = cnf.Chain(cnf.Dense(4, 12, "tanh"), cnf.Dense(12, 4, "tanh"))
        icf = cnf.construct(cnf.RNODE, nn, 2, 2, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 3.0), steer_rate=0.1, lambda3=1e-2, rng=5)
        model = cnf.ICNFModel(icf, optimizers=(cnf.Adam(eta=1e-3),), n_epochs=2, batch_size=32, pipelined=pipelined)
        (psf, _), _, rep = cnf.fit(model, 0, data)
        assert rep["stats"]["pipelined"] == pipelined
        res.append((psf, rep["losses"]))
        icf.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    # ... and a pipelined fit whose submitted launches GIVE UP (every wait runs out at its first poll) loses no step: the gated
    # update leaves parameters and momentum alone, the batches run again synchronously (on the streamed gradient path, since the
    # synchronous in-launch attempt gives up as well) and no NaN reaches the report (mlj.py)
    nn = cnf.Chain(cnf.Dense(4, 12, "tanh"), cnf.Dense(12, 4, "tanh"))
    icf = cnf.construct(cnf.RNODE, nn, 2, 2, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 3.0), steer_rate=0.1, lambda3=1e-2, rng=5)
    icf.set_solve_wait(poll_limit=1)
    model = cnf.ICNFModel(icf, optimizers=(cnf.Adam(eta=1e-3),), n_epochs=2, batch_size=32, pipelined=True)
    fb0 = icf.solve_fallbacks()
    (psg, _), _, repg = cnf.fit(model, 0, data)
    assert icf.solve_fallbacks() > fb0, "no launch gave up: the test did not exercise the path"
    # (a batch that runs again draws new probes and a new steered end time, so the runs are not comparable number by number:
    # every iteration has a finite loss of the same size, and the parameters moved as far as 16 Adam steps move them)
    assert np.isfinite(repg["losses"]).all() and len(repg["losses"]) == len(res[1][1]) == repg["stats"]["iterations"]
    assert abs(np.mean(repg["losses"]) / np.mean(res[1][1]) - 1.0) <= 0.25, (repg["losses"], res[1][1])
    assert np.isfinite(psg).all() and 0 < np.abs(psg - res[1][0]).max() <= 2 * 16 * 1e-3
    icf.close()


def test_set_params_async_does_not_change_a_submitted_inference():
    """ADVICE round 4 (cnf_abi.hip): a submitted INFERENCE that gives up is run again by its collect call with the parameters the
    handle holds THEN, so cnf_set_params_async must not slip new parameters under it: it settles such submissions first.  A
    submission whose launch gives up for certain (poll_limit = 1), new parameters uploaded asynchronously before the collect: the
    collected result is the one of the parameters it was submitted with."""
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(1850)
    B = 8192
    flat_a = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    flat_b = (flat_a * 0.5).astype(np.float32)
    xs, eps = _dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    pa, pb = torch.from_numpy(flat_a).cuda(), torch.from_numpy(flat_b).cuda()
    want_a, _ = cnf.inference(ic, cnf.TrainMode(), xs, pa, {}, eps=eps)
    want_a = want_a.clone()
    want_b, _ = cnf.inference(ic, cnf.TrainMode(), xs, pb, {}, eps=eps)
    want_b = want_b.clone()
    assert float((want_a - want_b).abs().max()) > 1e-2
    if _one_launch_expected():
        ic.set_solve_wait(poll_limit=1)                              # the submitted launch gives up; its collect runs it again
    logpx, _, _ = cnf.inference_submit(ic, cnf.TrainMode(), xs, pa, {}, eps=eps, with_sums=True)
    ic.set_params_async(pb)                                          # settles the submission first (host wait), then uploads
    cnf.inference_collect(ic)
    torch.cuda.synchronize()
    assert torch.allclose(logpx, want_a, rtol=2e-5, atol=2e-5), float((logpx - want_a).abs().max())
    ic.set_solve_wait(poll_limit=0x7fffffff)
    got_b, _ = cnf.inference(ic, cnf.TrainMode(), xs, pb, {}, eps=eps)
    assert torch.allclose(got_b, want_b, rtol=2e-5, atol=2e-5)
    ic.close()


def test_loss_grad_wave_local_hands_over_beyond_its_step_store():
    """k_solve_wave<GRAD> keeps the step sizes of at most WV_GCAP = 1024 accepted steps: a solve with more ends without a
    gradient and the call runs again on the streamed gradient path -- same loss, same gradient as the oracle."""
    cfg = O.Cfg(O.Net((6, 12, 6), (O.ACT_TANH,) * 2), 6, 0, 1e-2, 1e-2, 0.0)
    val, grad, rval, rgrad, st, _ = _grad_case(cfg, 20, 951, "mfma", dict(adaptive=False, dt=1 / 1100),
                                               dict(adaptive=False, dt=1 / 1100))
    assert st["naccept"] == 1100 and st["launches"] > 2, st
    assert abs(val - rval) <= 1e-5 * max(1.0, abs(rval))
    _assert_grad(grad, rgrad, "1100 steps")


def test_config4_full_batch_on_one_gpu():
    """BASELINE config 4 unsharded: FFJORD, 65536 columns on one GPU (2048 tiles over 512 workgroups: several
    tiles per workgroup, 512 error partials).  Sampled columns of a fixed-dt solve against the oracle; the
    adaptive solve of the whole batch finishes with consistent counters."""
    cfg, B, _ = O.baseline_cfg(4)
    rng = np.random.default_rng(44)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    logpx, regs, sums = cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps), with_sums=True)
    assert torch.isfinite(logpx).all() and float(regs[0].abs().max()) == 0.0      # FFJORD: E and n rows are zero
    if _one_launch_expected():        # 2048 tiles: eight per workgroup of ONE launch (k_solve3jb), state in the integrator's buffers
        assert ic.last_stats["launches"] <= 3, ic.last_stats
    assert float(sums[4]) == B and abs(float(sums[0]) - float(logpx.double().sum())) <= 1e-5 * abs(float(logpx.double().sum()))
    idx = rng.choice(B, 48, replace=False)
    _, ref_lp, _, _ = O.inference(cfg, flat.astype(np.float64), xs[:, idx].astype(np.float64),
                                  eps[:, idx].astype(np.float64), True, dt=1 / 8, adaptive=False)
    assert_parity(logpx[torch.from_numpy(idx).cuda()].cpu().numpy(), ref_lp, "cfg4 full batch, sampled columns")
    # the sharded leg of BASELINE config 4 on this one GPU: eight ranks' column blocks (parallel.shard_range), each solved
    # as its own 8192-column one-launch solve (what each rank of an 8-GPU node runs), the 5-float sums added as the
    # all-reduce adds them: the columns equal the unsharded solve's bit for bit (fixed dt: columns do not interact) and the
    # loss of the summed sums is the unsharded loss
    from continuousnf.jl_amd.parallel import shard_range
    xd, ed = _dev(xs), _dev(eps)
    tot = torch.zeros(5, device="cuda", dtype=torch.float64)
    for r in range(8):
        lo, hi = shard_range(B, 8, r)
        assert hi - lo == 8192
        lp_r, _, sums_r = cnf.inference(ic, cnf.TrainMode(), xd[:, lo:hi].contiguous(), flat, {}, eps=ed[:, lo:hi].contiguous(),
                                        with_sums=True)
        if _one_launch_expected():
            assert ic.last_stats["launches"] == 1, ic.last_stats
        assert torch.equal(lp_r, logpx[lo:hi]), (r, float((lp_r - logpx[lo:hi]).abs().max()))
        tot += sums_r.double()
    assert float(tot[4]) == B
    loss_sharded = float(cnf.loss_from_sums(ic, cnf.TrainMode(), tot.float()))
    loss_whole = float(cnf.loss_from_sums(ic, cnf.TrainMode(), sums))
    assert abs(loss_sharded - loss_whole) <= 2e-6 * abs(loss_whole), (loss_sharded, loss_whole)
    kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    ica = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=kw)
    lp_all, _ = cnf.inference(ica, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
    st = ica.last_stats
    assert st["nf"] == 2 + 6 * (st["naccept"] + st["nreject"]) and st["t_final"] == 1.0
    assert torch.isfinite(lp_all).all()
    if _one_launch_expected():
        assert st["launches"] <= 3, st
    # the adaptive solve of the whole batch against the float32 C oracle on a sub-batch: different step sequences (the
    # error norm is over the batch), same bar as test_adaptive_solve_vs_oracles
    sub = slice(1000, 1256)
    u0 = O.inference_u0(cfg, xs[:, sub], True)
    ref64, _ = O.tsit5_solve(cfg.rhs(flat.astype(np.float64), eps[:, sub].astype(np.float64), True), u0.astype(np.float64),
                             *cfg.tspan, reltol=1e-10, abstol=1e-10)
    ref_lp = O.inference_sol(cfg, ref64, True)[0]
    assert_parity(lp_all[sub].cpu().numpy(), ref_lp, "cfg4 full batch adaptive, sub-batch vs float64", rtol=5e-3)


def test_inference_with_sums_and_distributed_loss_single_process():
    """cnf_inference_sums (solve + post-processing + local loss sums in one call) agrees with the separate
    calls, and parallel.distributed_loss without a process group is the plain loss."""
    from continuousnf.jl_amd.parallel import distributed_loss
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(31)
    B = 300
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs, eps = _dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))
    ic = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=1 / 16))
    for mode in (cnf.TrainMode(), cnf.TestMode()):
        logpx, regs, sums = cnf.inference(ic, mode, xs, flat, {}, eps=eps, with_sums=True)
        lp2, regs2 = cnf.inference(ic, mode, xs, flat, {}, eps=eps)
        assert torch.equal(logpx, lp2) and all(torch.equal(a, b) for a, b in zip(regs, regs2))
        ref = cnf.loss_sums(ic, lp2, regs2)
        assert torch.allclose(sums, ref, rtol=1e-6, atol=0) and float(sums[4]) == B
        val = distributed_loss(ic, mode, xs, flat, {}, eps=eps)
        assert abs(val - cnf.loss(ic, mode, xs, flat, {}, eps=eps)) <= 1e-6 * max(1.0, abs(val))
    with pytest.raises(ValueError):
        cnf.inference(ic, cnf.TrainMode(), xs.cpu().numpy(), flat, {}, eps=eps.cpu().numpy(), with_sums=True)


def test_upload_cache_is_keyed_on_the_object_not_its_address():
    """Two temporaries of equal size in a row: the caching allocator may hand the second one the address of the
    first, so an address-keyed upload cache would silently keep the old weights (round-1 advisor finding)."""
    g, cfg = load_golden("cfg2_regression")
    icnf = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=float(g["dt"])))
    xs, eps = _dev(g["xs"]), _dev(g["eps"])

    def run(scale):
        return cnf.inference(icnf, cnf.TrainMode(), xs, torch.from_numpy(g["flat"] * scale).cuda(), {}, eps=eps)[0].clone()

    a, b = run(1.0), run(0.5)
    assert not torch.allclose(a, b)
    assert_parity(a.cpu().numpy(), g["logpx_train_vjp"], "temporary params 1")
    # in-place update of a tensor the caller keeps: the version counter invalidates the cache
    ps = torch.from_numpy(g["flat"].copy()).cuda()
    a2 = cnf.inference(icnf, cnf.TrainMode(), xs, ps, {}, eps=eps)[0].clone()
    ps.mul_(0.5)
    b2 = cnf.inference(icnf, cnf.TrainMode(), xs, ps, {}, eps=eps)[0].clone()
    assert torch.equal(a2, a) and torch.equal(b2, b)
    # sol_kwargs changed in place are seen too
    icnf.sol_kwargs["dt"] = float(g["dt"]) / 2
    c = cnf.inference(icnf, cnf.TrainMode(), xs, ps, {}, eps=eps)[0]
    assert icnf.last_stats["naccept"] == 2 * round(1.0 / float(g["dt"])) and not torch.equal(c, b2)
    icnf.close()


def test_loss_allreduce_through_the_c_abi_on_real_rccl():
    """cnf_comm_* / cnf_loss_allreduce against the real librccl: a one-rank communicator on this GPU (two ranks
    cannot share a device under RCCL; the two-process path is covered on CPU with the test double and by
    bench.py --gpus N on a multi-GPU node)."""
    from continuousnf.jl_amd.parallel import RcclComm
    g, cfg = load_golden("cfg2_regression")
    icnf = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=float(g["dt"])))
    comm = RcclComm(1, 0, RcclComm.unique_id(), 0)
    assert comm.size() == 1 and _lib.lib().cnf_comm_library()
    _, _, sums = cnf.inference(icnf, cnf.TrainMode(), _dev(g["xs"]), g["flat"], {}, eps=_dev(g["eps"]), with_sums=True)
    before = sums.clone()
    out = comm.allreduce_sums(icnf, sums)
    torch.cuda.synchronize()
    assert torch.equal(out, before) and float(out[4]) == g["xs"].shape[1]
    t = torch.arange(1000, dtype=torch.float32, device="cuda")
    assert torch.equal(comm.allreduce(t.clone()), t)
    # lock-step through the communicator: with one rank the adaptive solve is the plain adaptive solve
    kw = dict(reltol=1e-4, abstol=1e-6)
    ic2 = make_icnf(cnf, cfg, sol_kwargs=kw)
    ref = cnf.inference(ic2, cnf.TrainMode(), _dev(g["xs"]), g["flat"], {}, eps=_dev(g["eps"]))[0].clone()
    ref_st = dict(ic2.last_stats)
    comm.lockstep(ic2)
    got = cnf.inference(ic2, cnf.TrainMode(), _dev(g["xs"]), g["flat"], {}, eps=_dev(g["eps"]))[0]
    assert (ic2.last_stats["naccept"], ic2.last_stats["nreject"]) == (ref_st["naccept"], ref_st["nreject"])
    assert_parity(got.cpu().numpy(), ref.cpu().numpy(), "RCCL lock-step, 1 rank", rtol=1e-5)
    comm.lockstep(ic2, enable=False)
    # bench.py's loop on the real RCCL: every step submitted, its all-reduce enqueued behind it on the same stream,
    # collected one step later -- the collective sits between two one-launch solves of the headline shape
    cfg3, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(12)
    flat3 = O.glorot_params(cfg3.net, rng, np.float32, 0.1)
    ic3 = make_icnf(cnf, cfg3, sol_kwargs=dict(reltol=3.45e-4, abstol=1.19e-7))
    xs3, eps3 = _dev(rng.standard_normal((cfg3.nvars, 8192))), _dev(rng.standard_normal((cfg3.n_in, 8192)))
    _, _, want = cnf.inference(ic3, cnf.TrainMode(), xs3, flat3, {}, eps=eps3, with_sums=True)
    want = want.clone()
    pend = []
    for _ in range(6):
        _, _, local = cnf.inference_submit(ic3, cnf.TrainMode(), xs3, flat3, {}, eps=eps3, with_sums=True)
        pend.append(comm.allreduce_sums(ic3, local))
        if len(pend) > 2:
            cnf.inference_collect(ic3)
            assert torch.equal(pend.pop(0), want)
    while pend:
        cnf.inference_collect(ic3)
        assert torch.equal(pend.pop(0), want)
    comm.close(); icnf.close(); ic2.close(); ic3.close()


# ---------------------------------------------------------------------------------------
# the headline kernels in front of the float64 oracle at size
# ---------------------------------------------------------------------------------------
def _one_launch_expected():
    return os.environ.get("CNF_PERSISTENT") != "0" and os.environ.get("CNF_STEP_FP32") != "1" \
        and os.environ.get("CNF_STEP_V1") != "1"


@pytest.mark.parametrize("B,seed", [(1000, 0), (8192, 0), (8224, 0)] + [(8192, s) for s in range(1, 6)])
def test_headline_kernels_strict_vs_float64_at_size(B, seed):
    """Fixed-dt inference of the headline shape (RNODE 32-128-128-32, VJP with the |eps^T J| row) through the kernels the
    bench number comes from: k_solve3b (B = 1000: 32 tiles, ragged; B = 8192: 256 workgroups that meet at every step;
    B = 8224: two tiles per workgroup, the state in the integrator's buffers) and k_step3b (every B in the
    CNF_PERSISTENT=0 child run of test_ab_switches); seeds 1..5 at B = 8192 are SURVEY 8 d-inputs' five draws of
    (xs, eps, weights).  fsol (all rows), logpx and the regularisers of 288 sampled columns -- first and last tile, the
    ragged tail, random ones -- against the float64 oracle at the 1e-4 bar, and the route is asserted (launches)."""
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(1300 + B + 7919 * seed)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    kw = dict(adaptive=False, dt=1 / 8)
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=kw)
    prob = cnf.inference_prob(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
    fsol = cnf.base_sol(ic, prob).view()
    # (B = 8224: 257 tiles over 129 workgroups, the several-tiles instantiation of k_solve3b)
    one = _one_launch_expected()
    assert prob.stats["kernel_used"] == _lib.KERNEL_MFMA and prob.stats["nf"] == 1 + 6 * 8
    assert (prob.stats["launches"] <= 3) == one, prob.stats          # one launch (+ the copy of the final state), or streamed
    logpx, (E, n, A) = cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
    assert (ic.last_stats["launches"] <= 3) == one, ic.last_stats
    idx = np.unique(np.concatenate([np.arange(32), np.arange(B - 40, B), rng.choice(B, 224, replace=False)]))
    f64 = lambda a: a.astype(np.float64)
    u0 = O.inference_u0(cfg, xs[:, idx], True)
    ref, _ = O.tsit5_solve(cfg.rhs(f64(flat), f64(eps[:, idx]), True), f64(u0), *cfg.tspan, dt=1 / 8, adaptive=False)
    ti = torch.from_numpy(idx).cuda()
    assert_parity(fsol[:, ti].cpu().numpy(), ref, f"headline kernel fixed-dt fsol B={B}", trace_row=cfg.n_in)
    _, ref_lp, ref_regs, _ = O.inference(cfg, f64(flat), f64(xs[:, idx]), f64(eps[:, idx]), True, dt=1 / 8, adaptive=False)
    assert_parity(logpx[ti].cpu().numpy(), ref_lp, f"headline kernel logpx B={B}")
    assert_parity(torch.stack([E, n, A])[:, ti].cpu().numpy(), ref_regs, f"headline kernel regs B={B}")
    assert ic.solve_fallbacks() == 0
    ic.close()


def test_split_product_error_bound():
    """The arithmetic of the split kernels, isolated (cnf_selftest_split_product: operand split + six-term bf16 MFMA
    product, as k_step3b / k_solve3b run it) against float64.  An fp32 value is the exact sum of three bf16 pieces taken by
    round-to-nearest, |m| <= 2^-8 |v|, |l| <= 2^-16 |v|, so the three products left out (m l, l m, l l) stay below
    2^-23 |a b| for every input (truncated pieces, round 2's form: 2^-21 with every mantissa bit set -- measured 4e-7);
    the rest is the fp32 accumulation inside and between the MFMAs.  Bounds asserted, relative to S = sum_k |a_k b_k| per
    output entry (measured on MI355X in brackets):
      random N(0,1), K = 32 .. 512 .................. 2^-22 S   [0.65 .. 1.3e-7; v_mfma_f32_16x16x4_f32: 7.6e-8]
      magnitudes over 40 binades, random signs ....... 2^-20 S   [4.1e-7: the MFMA aligns the 32 products of a k-block to
                                                                  the largest one; a property of the accumulate, not of the split]
      all-ones mantissas, one sign ................... 2^-22 S   [1.2e-7]
      exactly cancelling K = 128 sums ................ 2^-22 S absolute (the exact result is 0)   [7.4e-8]
    The measured figures go to parity_report.json."""
    l = _lib.lib()
    rng = np.random.default_rng(2024)

    def run(A, Bt):
        K = A.shape[1]
        A = np.ascontiguousarray(A, dtype=np.float32); Bt = np.ascontiguousarray(Bt, dtype=np.float32)
        Cm = np.empty((16, 16), np.float32)
        _lib.check(l.cnf_selftest_split_product(A.ctypes.data, Bt.ctypes.data, Cm.ctypes.data, K))
        ref = A.astype(np.float64) @ Bt.astype(np.float64).T
        S = np.abs(A).astype(np.float64) @ np.abs(Bt).astype(np.float64).T
        return float(np.max(np.abs(Cm - ref) / S)), float(np.max(np.abs((A @ Bt.T).astype(np.float64) - ref) / S))

    worst = {}
    for K in (32, 64, 128, 512):
        e = max(run(rng.standard_normal((16, K)), rng.standard_normal((16, K)))[0] for _ in range(8))
        worst[f"random K={K}"] = e
        assert e <= 2.0 ** -22, (K, e)
    e = 0.0
    for _ in range(8):
        mag = lambda: np.exp2(rng.uniform(-20, 20, (16, 128))) * rng.choice([-1.0, 1.0], (16, 128))
        e = max(e, run(mag(), mag())[0])
    worst["40 binades"] = e
    assert e <= 2.0 ** -20, e
    ones = np.full((16, 128), np.float32(np.nextafter(np.float32(2.0), np.float32(0.0))))      # 0x3FFFFFFF: every bit set
    e = run(ones, ones)[0]
    worst["all-ones mantissas"] = e
    assert e <= 2.0 ** -22, e
    A = rng.standard_normal((16, 128)).astype(np.float32)
    Bt = rng.standard_normal((16, 128)).astype(np.float32)
    A[:, 64:] = A[:, :64]; Bt[:, 64:] = -Bt[:, :64]            # sum_k a_k b_k = 0 exactly, term by term
    e = run(A, Bt)[0]
    worst["cancelling"] = e
    assert e <= 2.0 ** -22, e
    helpers.REPORT.append({"what": "split product |C - f64| / sum|ab|", "shape": [16, 16], "rtol": 2.0 ** -20,
                           "err_over_bar": max(worst.values()) / 2.0 ** -20, "max_abs_err": max(worst.values()),
                           "max_rel_err": max(worst.values()), "mean_rel_err": float(np.mean(list(worst.values()))),
                           "cases": worst})
    with pytest.raises(_lib.CNFError):
        _lib.check(l.cnf_selftest_split_product(A.ctypes.data, Bt.ctypes.data, A.ctypes.data, 48))


def test_step_trace_follows_the_c_oracle():
    """cnf_set_step_trace: (t, h, EEst, accepted) of every step attempt of a one-launch adaptive solve, against the same
    log of the float32 C oracle (one control law).  The accepted steps tile the span, the attempt counts agree to +-2
    and, wherever EEst is above round-off level, the two estimates agree to a few per cent."""
    if not _one_launch_expected():
        pytest.skip("the trace is filed by the one-launch solve")
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(1400)
    B = 512
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.05)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=kw)
    tr = ic.set_step_trace(64)
    cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
    st = ic.last_stats
    na = st["naccept"] + st["nreject"]
    g = tr.cpu().numpy()[:na]
    ctr = np.zeros((64, 4), np.float32)
    _, cst = CO.solve(cfg, flat, O.inference_u0(cfg, xs, True), eps, True, trace=ctr, **kw)
    nc = cst["naccept"] + cst["nreject"]
    c = ctr[:nc]
    assert abs(na - nc) <= 2 and g[0, 0] == 0.0
    assert abs(float(g[g[:, 3] > 0, 1].sum()) - 1.0) < 1e-5          # accepted steps tile the span
    assert np.all(np.diff(g[g[:, 3] > 0, 0]) > 0)
    # the first attempt starts from the same t and the same automatic dt; its error estimate is a round-off-level quantity
    # on this well-resolved problem (1e-4: the products differ in rounding between the two), so it agrees in magnitude only;
    # where an estimate is well above round-off the two agree to a few per cent
    assert abs(g[0, 1] / c[0, 1] - 1.0) < 1e-3, (g[0], c[0])
    assert 0.2 < g[0, 2] / c[0, 2] < 5.0, (g[0], c[0])
    m = min(na, nc)
    sel = (c[:m, 2] > 1e-2) & (g[:m, 2] > 1e-2) & (np.abs(g[:m, 0] - c[:m, 0]) <= 1e-3 * np.maximum(1e-3, np.abs(c[:m, 0])))
    assert np.all(np.abs(g[:m, 2][sel] / c[:m, 2][sel] - 1.0) < 0.05), (g[:m], c[:m])
    assert np.all((g[:, 2] >= 0) & (g[:, 2] <= 1.0 + 1e-6) | (g[:, 3] == 0))      # accepted attempts have EEst <= 1
    assert ic.set_step_trace(0) is None
    ic.close()


@pytest.mark.parametrize("regime", ["smooth", "stiff"])
@pytest.mark.parametrize("which", ["cfg3", "cfg3-jvp", "cfg2", "cfg5", "cfg5-test"])
def test_adaptive_solves_on_their_own_step_sequence(which, regime):
    """adaptive solves were only compared at the solver tolerance (5e-3), because a float64 controller
    takes other steps than a float32 one.  Here the one-launch solve files (t, h, EEst, accepted) of every attempt
    (cnf_set_step_trace) and the float64 oracle REPLAYS exactly those attempts (oracle.tsit5_step at the device's h).
    * smooth (the bench workload's inputs, SURVEY 8d: Glorot weights, tspan (0, 1), ~20 attempts, estimates at round-off level): the final state,
      logpx and the regularisers agree at the STRICT 1e-4 bar -- the arithmetic of an adaptive solve is as exact as a fixed-dt one.
    * stiff (weights x 5, tspan (0, 2): 50-70 attempts with estimates of 1e-2 .. 1; rounding differences are amplified along the
      flow, so values are compared at the solver tolerance): every accept / reject decision is the one the oracle's own estimate
      gives on the same attempt wherever that estimate is clear of 1 by 10 %, and the two estimates agree to 15 % wherever both
      are above 1e-2.
    Configs 3 (VJP, JVP), 2, 5 (Train and exact trace): k_solve3b, k_solve3jb, k_solve_wave, k_solve_bcast."""
    if not _one_launch_expected():
        pytest.skip("the step trace is filed by the one-launch solves")
    jvp = which.endswith("-jvp")
    train = not which.endswith("-test")
    stiff = regime == "stiff"
    if stiff and which in ("cfg3-jvp", "cfg5-test"):
        pytest.skip("the stiff regime runs on the three kernel families' VJP forms (suite time)")
    ci = int(which[3])
    cfg, _, _ = O.baseline_cfg(ci)
    cfg.use_jvp = jvp
    cfg.tspan = (0.0, 2.0) if stiff else (0.0, 1.0)
    B = {2: 1000, 3: 1024, 5: 200}[ci]
    rng = np.random.default_rng(1450 + ci + (7 if jvp else 0) + (3 if not train else 0))
    flat = (O.glorot_params(cfg.net, rng, np.float32, 0.3) * (5.0 if stiff else 1.0)).astype(np.float32)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32) if train else None
    kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    ic = make_icnf(cnf, cfg, jvp=jvp, kernel="mfma", tag=cnf.RNODE, sol_kwargs=kw)
    tr = ic.set_step_trace(256)
    mode = cnf.TrainMode() if train else cnf.TestMode()
    prob = cnf.inference_prob(ic, mode, _dev(xs), flat, {}, eps=_dev(eps) if train else None)
    fsol = cnf.base_sol(ic, prob).view().cpu().numpy()
    st = prob.stats
    assert st["launches"] <= 3, st
    na = st["naccept"] + st["nreject"]
    g = tr.cpu().numpy()[:na].astype(np.float64)
    assert na <= 256 and int(g[:, 3].sum()) == st["naccept"]
    # replay in float64: the same attempts, the oracle's own error estimate beside the device's
    f64 = lambda a: None if a is None else a.astype(np.float64)
    f = cfg.rhs(f64(flat), f64(eps), train)
    u = O.inference_u0(cfg, f64(xs), train)
    k1 = f(u)
    t = float(cfg.tspan[0])
    n_clear = n_checked = 0
    worst = 0.0
    for (tg, hg, eg, ag) in g:
        assert abs(tg - t) <= 1e-5 * max(1.0, abs(t)), (tg, t)
        un, k7, err = O.tsit5_step(f, u, k1, hg)
        sc = kw["abstol"] + kw["reltol"] * np.maximum(np.abs(u), np.abs(un))
        eo = O._rms(err / sc)
        if eo > 1e-2 and eg > 1e-2:
            n_checked += 1
            worst = max(worst, abs(eg / eo - 1.0))
            assert abs(eg / eo - 1.0) <= 0.15, (which, tg, hg, eg, eo)
        if abs(eo - 1.0) > 0.1:                                # the decision is not a matter of rounding
            n_clear += 1
            assert (ag > 0) == (eo <= 1.0), (which, tg, hg, eg, eo, ag)
        if ag > 0:
            u, k1, t = un, k7, t + hg
    assert abs(t - cfg.tspan[1]) <= 1e-5
    helpers.note(f"adaptive solve replayed in float64 on the device's own {na} attempts ({st['nreject']} rejected), {which} / {regime}: "
                 f"{n_checked} error estimates above 1e-2 (worst disagreement {100 * worst:.1f} %), {n_clear} decisions checked")
    if stiff:
        assert n_clear >= na - 3 and n_checked >= na // 3, (which, na, n_clear, n_checked)
    rt = dict(rtol=5e-3) if stiff else {}
    assert_parity(fsol, u, f"adaptive {which} ({regime}): final state on the device's own steps", trace_row=cfg.n_in, **rt)
    logpx, regs = cnf.inference(ic, mode, _dev(xs), flat, {}, eps=_dev(eps) if train else None)
    ref_lp, ref_regs = O.inference_sol(cfg, u, train)
    assert_parity(logpx.cpu().numpy(), ref_lp, f"adaptive {which} ({regime}): logpx on the device's own steps", **rt)
    if train:
        assert_parity(torch.stack(list(regs)).cpu().numpy(), np.stack(ref_regs), f"adaptive {which} ({regime}): regularisers on the device's own steps", **rt)
    ic.close()


def test_one_launch_solve_falls_back_when_a_workgroup_does_not_arrive():
    """VERDICT round 2, item 3.  (a) A long kernel of another stream holds CUs while cnf_inference runs: the one-launch
    solve either gets all its workgroups placed in time or runs out of a wait -- the call returns CNF_OK with a correct
    result either way.  (b) In the CNF_SOLVE_POLL_LIMIT=1 child run of test_ab_switches every wait runs out at once: the
    abort path is taken for certain, the streamed driver produces the result, the counter says so."""
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(1500)
    B = 8192
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs, eps = _dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))
    forced = os.environ.get("CNF_SOLVE_POLL_LIMIT") == "1"
    ref_ic = make_icnf(cnf, cfg, kernel="generic", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    sub = slice(4000, 4064)
    ref, _ = cnf.inference(ref_ic, cnf.TrainMode(), xs[:, sub].contiguous(), flat, {}, eps=eps[:, sub].contiguous())
    ref = ref.clone()
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)           # parameters resident, buffers sized
    torch.cuda.synchronize()
    base = ic.solve_fallbacks()
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(40):                                    # ~0.5 s of fp32 GEMMs that fill the chip
            a = torch.mm(a, a) * 1e-4
    logpx, _, sums = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True)
    torch.cuda.synchronize()
    fb = ic.solve_fallbacks() - base
    assert torch.isfinite(logpx).all() and float(sums[4]) == B
    assert torch.allclose(logpx[sub], ref, rtol=2e-5, atol=2e-5), float((logpx[sub] - ref).abs().max())
    assert abs(float(sums[0]) - float(logpx.double().sum())) <= 1e-5 * abs(float(sums[0]))
    if not _one_launch_expected():
        assert fb == 0 and ic.last_stats["launches"] > 3, (fb, ic.last_stats)      # streamed anyway: nothing to fall back from
    elif forced:
        assert fb == 1 and ic.last_stats["launches"] > 3, (fb, ic.last_stats)
    else:
        assert (fb == 0) == (ic.last_stats["launches"] <= 3), (fb, ic.last_stats)
    # and the handle goes on working on the one-launch path afterwards
    lp2, _ = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    torch.cuda.synchronize()
    assert torch.allclose(lp2, logpx, rtol=2e-5, atol=2e-5)
    ic.close(); ref_ic.close()


def test_one_launch_solve_gives_up_within_its_time_bound_when_cus_are_held():
    """the stall a co-tenant can cause is bounded in TIME.  128 workgroups of another stream hold a
    whole CU each for 8 ms (cnf_selftest_hold_cus): the 256 workgroups of the headline solve cannot all be placed, the
    launch's waits run out after cnf_set_solve_wait's bound (2 ms by default), the call falls back to the streamed driver
    and returns CNF_OK.  Compared with the same call when every wait gives up at its first poll (poll_limit = 1: no
    waiting at all), the run-out may cost at most 10 ms -- and did cost something (the bound was what ended the wait)."""
    import time
    if not _one_launch_expected():
        pytest.skip("the one-launch solve is switched off in this process")
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(1600)
    B = 8192
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs, eps = _dev(rng.standard_normal((cfg.nvars, B))), _dev(rng.standard_normal((cfg.n_in, B)))
    tol = dict(reltol=3.45e-4, abstol=1.19e-7)
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=tol)
    want, _ = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    assert ic.last_stats["launches"] <= 3
    want = want.clone()
    side = torch.cuda.Stream()
    l = _lib.lib()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count

    def held():
        torch.cuda.synchronize()
        assert l.cnf_selftest_hold_cus(ncu // 2, 8000, C.c_void_p(side.cuda_stream)) == 0
        time.sleep(0.001)                                       # (the holders are placed before the solve is launched)
        t0 = time.perf_counter()
        lp, _ = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
        torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
        return dt, lp, dict(ic.last_stats)
    fb0 = ic.solve_fallbacks()
    ic.set_solve_wait(poll_limit=1)                             # every wait gives up at once: the fallback without the wait
    runs = [held() for _ in range(3)]
    assert ic.solve_fallbacks() - fb0 == 3
    assert all(torch.allclose(lp, want, rtol=5e-3, atol=5e-3) for _, lp, _ in runs)
    t_now = min(r[0] for r in runs)
    ic.set_solve_wait(poll_limit=0x7fffffff)                    # bounded by time alone (2 ms)
    runs = [held() for _ in range(3)]
    assert ic.solve_fallbacks() - fb0 == 6, ("the launch was placed although half of the CUs were held?", [(r[0], r[2]) for r in runs])
    assert all(torch.allclose(lp, want, rtol=5e-3, atol=5e-3) for _, lp, _ in runs)
    assert runs[-1][2]["launches"] > 3
    t_wait = min(r[0] for r in runs)
    helpers.note(f"one-launch solve with half of the CUs held by another stream: fallback after the time bound {1e3 * t_wait:.2f} ms, "
                 f"without waiting {1e3 * t_now:.2f} ms")
    assert t_wait - t_now <= 10e-3, (t_wait, t_now)
    assert t_wait - t_now >= 0.5e-3, (t_wait, t_now)
    # with the CUs free again the handle takes the one-launch path
    torch.cuda.synchronize()
    lp2, _ = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)
    assert ic.last_stats["launches"] <= 3 and torch.equal(lp2, want)
    ic.close()


def test_testmode_one_launch_solve_falls_back_to_the_streamed_driver():
    """ADVICE round 4 (cnf_abi.hip, solve_core): TestMode of the headline network runs k_trace3s<SOLVE> on a state that was
    assembled in the integrator's own buffer (no fused I/O).  When a wait of that launch runs out the call must rebuild u0
    from the caller's data columns and run on the streamed driver -- it used to return CNF_ERR_HIP.  poll_limit = 1 makes
    every wait run out at its first poll, so the abort path is taken for certain."""
    if not _one_launch_expected():
        pytest.skip("the one-launch solve is switched off in this process")
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(1650)
    B = 2048
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs = _dev(rng.standard_normal((cfg.nvars, B)))
    tol = dict(reltol=3.45e-4, abstol=1.19e-7)
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=tol)
    want, _ = cnf.inference(ic, cnf.TestMode(), xs, flat, {})
    one = ic.last_stats["launches"]
    want = want.clone()
    fb0 = ic.solve_fallbacks()
    ic.set_solve_wait(poll_limit=1)
    got, _ = cnf.inference(ic, cnf.TestMode(), xs, flat, {})
    torch.cuda.synchronize()
    if one <= 3:                                                 # the one-launch TestMode solve is what ran the first time
        assert ic.solve_fallbacks() - fb0 == 1 and ic.last_stats["launches"] > 3, ic.last_stats
    assert torch.isfinite(got).all()
    assert torch.allclose(got, want, rtol=5e-3, atol=5e-3), float((got - want).abs().max())
    ic.set_solve_wait(poll_limit=0x7fffffff)
    again, _ = cnf.inference(ic, cnf.TestMode(), xs, flat, {})
    assert ic.last_stats["launches"] == one and torch.equal(again, want)
    ic.close()


@pytest.mark.parametrize("dims,nvars,naugs,acts", [
    ((2, 6, 2), 1, 1, ("tanh", "tanh")),                      # BASELINE config 1 (README.md:47)
    ((16, 48, 16), 8, 8, ("tanh", "tanh")),                   # BASELINE config 2 (test/regression_tests.jl:7)
    ((8, 24, 8), 8, 0, ("tanh", "tanh")),
    ((13, 50, 13), 9, 4, ("softplus", "sigmoid")),            # odd sizes, sigma(0) != 0 in the padded rows
    ((32, 96, 32), 32, 0, ("tanh", "identity")),              # two input tiles, six hidden tiles
])
def test_wave_local_solve_small_networks(dims, nvars, naugs, acts):
    """k_solve_wave (cnf_wave.hip): two-layer networks whose padded widths fit one wave -- the whole solve in ONE launch,
    one wave per 16 samples, registers only.  VJP, JVP and TestMode (closed-form exact trace), fixed dt strictly against the
    float64 oracle and adaptive at the solver tolerance, ragged batches, the route asserted by the launch count, the
    CNF_WAVE=0 route (k_mfma behind the streamed driver) beside it for the same numbers."""
    act_id = {"tanh": O.ACT_TANH, "softplus": O.ACT_SOFTPLUS, "sigmoid": O.ACT_SIGMOID, "identity": O.ACT_IDENTITY}
    net = O.Net(dims, tuple(act_id[a] for a in acts))
    rng = np.random.default_rng(sum(dims) + 17)
    flat = O.glorot_params(net, rng, np.float32, 0.1)
    flat[-dims[2]:] = 0.1 * rng.standard_normal(dims[2]).astype(np.float32)          # non-zero biases
    layers = [cnf.Dense(i, o, a) for i, o, a in zip(dims[:-1], dims[1:], acts)]
    n_in = nvars + naugs
    f64 = lambda a: a.astype(np.float64)
    one = _one_launch_expected() and os.environ.get("CNF_WAVE") != "0"
    lam3 = 1e-2 if naugs else 0.0
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    for B in (1, 17, 1000):
        xs = rng.standard_normal((nvars, B)).astype(np.float32)
        eps = rng.standard_normal((n_in, B)).astype(np.float32)
        for jvp in (False, True):
            cm = cnf.HIPJacVecMatrixMode("mfma") if jvp else cnf.HIPVecJacMatrixMode("mfma")
            cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, lam3, jvp, tspan=(0.0, 2.0))
            ic = cnf.construct(cnf.RNODE, cnf.Chain(*layers), nvars, naugs, compute_mode=cm, tspan=(0.0, 2.0), lambda3=lam3,
                               sol_kwargs=dict(adaptive=False, dt=1 / 8))
            if not _supported(ic, cnf.TrainMode(), B):
                ic.close()
                continue
            # TrainMode, fixed dt: the final state, logpx and the regularisers at the strict bar
            prob = cnf.inference_prob(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
            fsol = cnf.base_sol(ic, prob).view().cpu().numpy()
            assert (prob.stats["launches"] <= 3) == one, (dims, B, jvp, prob.stats)
            assert prob.stats["nf"] == 1 + 6 * 16
            rf, ref_lp, ref_regs, _ = O.inference(cfg, f64(flat), f64(xs), f64(eps), True, dt=1 / 8, adaptive=False)
            assert_parity(fsol, rf, f"wave {dims} fsol jvp={jvp} B={B}", trace_row=n_in)
            logpx, (E, n, A) = cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
            assert (ic.last_stats["launches"] <= 3) == one
            assert_parity(logpx.cpu().numpy(), ref_lp, f"wave {dims} logpx jvp={jvp} B={B}")
            assert_parity(torch.stack([E, n, A]).cpu().numpy(), np.stack(ref_regs), f"wave {dims} regs jvp={jvp} B={B}")
            # loss sums in the same launch
            _, _, sums = cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps), with_sums=True)
            assert float(sums[4]) == B and abs(float(sums[0]) - float(logpx.double().sum())) <= 1e-5 * (abs(float(sums[0])) + 1.0)
            ic.close()
        # TestMode: the closed-form exact trace
        cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, lam3, False, tspan=(0.0, 2.0))
        ic = cnf.construct(cnf.RNODE, cnf.Chain(*layers), nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode("mfma"), tspan=(0.0, 2.0),
                           lambda3=lam3, sol_kwargs=dict(adaptive=False, dt=1 / 8))
        if _supported(ic, cnf.TestMode(), B):
            logpx, (E, n, A) = cnf.inference(ic, cnf.TestMode(), _dev(xs), flat, {})
            assert (ic.last_stats["launches"] <= 3) == one, ic.last_stats
            _, ref_lp, ref_regs, _ = O.inference(cfg, f64(flat), f64(xs), None, False, dt=1 / 8, adaptive=False)
            assert_parity(logpx.cpu().numpy(), ref_lp, f"wave {dims} TestMode logpx B={B}")
            assert float(E.abs().max()) == 0.0 and float(n.abs().max()) == 0.0
            assert_parity(A.cpu().numpy(), ref_regs[2], f"wave {dims} TestMode A B={B}")
        ic.close()
    # adaptive (README tolerances), B = 1000: against a tight float64 solve at the solver tolerance, step counts consistent
    B = 1000
    xs = rng.standard_normal((nvars, B)).astype(np.float32)
    eps = rng.standard_normal((n_in, B)).astype(np.float32)
    cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, lam3, False, tspan=(0.0, 2.0))
    ic = cnf.construct(cnf.RNODE, cnf.Chain(*layers), nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode("mfma"), tspan=(0.0, 2.0),
                       lambda3=lam3, sol_kwargs=tol)
    if _supported(ic, cnf.TrainMode(), B):
        prob = cnf.inference_prob(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
        fsol = cnf.base_sol(ic, prob).view().cpu().numpy()
        st = prob.stats
        assert (st["launches"] <= 3) == one and st["nf"] == 2 + 6 * (st["naccept"] + st["nreject"]) and abs(st["t_final"] - 2.0) < 1e-6
        u0 = O.inference_u0(cfg, xs, True)
        ref64, _ = O.tsit5_solve(cfg.rhs(f64(flat), f64(eps), True), f64(u0), 0.0, 2.0, reltol=1e-10, abstol=1e-10)
        assert_parity(fsol, ref64, f"wave {dims} adaptive vs float64", rtol=5e-3, trace_row=n_in)
        _, cst = CO.solve(cfg, flat, u0, eps, True, **tol)
        # (abstol = eps: the error estimate of the rows that start at 0 is rounding noise, the step count follows it)
        assert abs(st["naccept"] - cst["naccept"]) <= max(3, 0.15 * cst["naccept"]), (st, cst)
    ic.close()


@pytest.mark.parametrize("B", [9000, 32768])
def test_wave_local_solve_beyond_512_tiles(B):
    """VERDICT round 4, item 2 (src/base_icnf.jl:266-286 takes any column count): BASELINE config 2's network beyond 8192 columns
    stays on k_solve_wave -- workgroups of four tiles that meet through LDS inside and the tagged words between them (up to 512
    workgroups = 32768 columns) -- instead of falling to 23-34 step launches.  Fixed dt, sampled columns strictly against the
    float64 oracle (VJP, JVP, TestMode), the loss sums against the whole batch's logpx, the route by the launch count; adaptive:
    the same step counts as the C oracle on a column sample would not be comparable (the norm is over ALL columns), so the
    adaptive solve is checked against a tight float64 solve of sampled columns at the solver tolerance."""
    cfg, _, _ = O.baseline_cfg(2)
    net = cfg.net
    rng = np.random.default_rng(2200 + B)
    flat = O.glorot_params(net, rng, np.float32, 0.1)
    nvars, naugs, n_in = cfg.nvars, cfg.naugs, cfg.n_in
    xs = rng.standard_normal((nvars, B)).astype(np.float32)
    eps = rng.standard_normal((n_in, B)).astype(np.float32)
    one = _one_launch_expected() and os.environ.get("CNF_WAVE") != "0"
    f64 = lambda a: a.astype(np.float64)
    sub = np.r_[0:48, B // 2 - 7:B // 2 + 9, B - 64:B]                 # first / middle / last tiles (the last workgroup may be partial)
    for jvp in (False, True):
        c = O.Cfg(net, nvars, naugs, cfg.lam1, cfg.lam2, cfg.lam3, jvp, tspan=cfg.tspan)
        ic = make_icnf(cnf, c, jvp=jvp, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
        logpx, (E, n, A), sums = cnf.inference(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps), with_sums=True)
        assert (ic.last_stats["launches"] <= 3) == one, ic.last_stats
        assert ic.last_stats["nf"] == 1 + 6 * 8
        _, ref_lp, ref_regs, _ = O.inference(c, f64(flat), f64(xs[:, sub]), f64(eps[:, sub]), True, dt=1 / 8, adaptive=False)
        assert_parity(logpx.cpu().numpy()[sub], ref_lp, f"wave xwg logpx jvp={jvp} B={B}")
        assert_parity(torch.stack([E, n, A]).cpu().numpy()[:, sub], np.stack(ref_regs), f"wave xwg regs jvp={jvp} B={B}")
        assert float(sums[4]) == B and abs(float(sums[0]) - float(logpx.double().sum())) <= 1e-5 * (abs(float(sums[0])) + 1.0)
        assert abs(float(sums[1]) - float(E.double().sum())) <= 1e-5 * (abs(float(sums[1])) + 1.0)
        ic.close()
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    logpx, _ = cnf.inference(ic, cnf.TestMode(), _dev(xs), flat, {})
    assert (ic.last_stats["launches"] <= 3) == one, ic.last_stats
    _, ref_lp, _, _ = O.inference(cfg, f64(flat), f64(xs[:, sub]), None, False, dt=1 / 8, adaptive=False)
    assert_parity(logpx.cpu().numpy()[sub], ref_lp, f"wave xwg TestMode logpx B={B}")
    ic.close()
    # adaptive at the README tolerances: one launch, consistent counters, sampled columns at the solver tolerance
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=tol)
    prob = cnf.inference_prob(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
    fsol = cnf.base_sol(ic, prob).view().cpu().numpy()
    st = prob.stats
    assert (st["launches"] <= 3) == one and st["nf"] == 2 + 6 * (st["naccept"] + st["nreject"]) and abs(st["t_final"] - cfg.tspan[1]) < 1e-6
    u0 = O.inference_u0(cfg, xs[:, sub], True)
    ref64, _ = O.tsit5_solve(cfg.rhs(f64(flat), f64(eps[:, sub]), True), f64(u0), cfg.tspan[0], cfg.tspan[1], reltol=1e-10, abstol=1e-10)
    assert_parity(fsol[:, sub], ref64, f"wave xwg adaptive vs float64 B={B}", rtol=5e-3, trace_row=n_in)
    ic.close()


@pytest.mark.parametrize("B", [2048, 1000, 5, 4096, 4100, 16384])      # (beyond 2048: several tiles per workgroup, rows in global memory)
def test_config5_one_launch_solve_strict_vs_float64(B):
    """k_solve_bcast (cnf_bcast.hip): BASELINE config 5's network (RNODE 64 + 64, 128-384-128 tanh) at eight columns per CU --
    4x4x1 MFMA blocks with the activations broadcast, W1 resident, W2 streamed -- in ONE launch.  TrainMode / VJP and
    TestMode (the closed-form exact trace) at fixed dt: fsol, logpx and the regularisers of sampled columns against the
    float64 oracle at the strict bar; adaptive: step count and values at the solver tolerance; the route is asserted; ragged
    batches and a network narrower than the padding; CNF_BCAST=0 (k_mfma behind the streamed driver) runs the same test
    in test_ab_switches' child."""
    cases = [(O.baseline_cfg(5)[0], "cfg5")]
    if B == 1000:
        cases.append((O.Cfg(O.Net((100, 300, 100), (O.ACT_TANH,) * 2), 70, 30, 1e-2, 1e-2, 1e-2), "100-300-100"))
    one = _one_launch_expected() and os.environ.get("CNF_BCAST") != "0"
    f64 = lambda a: a.astype(np.float64)
    for cfg, name in cases:
        rng = np.random.default_rng(1700 + B)
        flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
        xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
        eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
        idx = np.unique(np.concatenate([np.arange(min(B, 16)), np.arange(max(0, B - 24), B), rng.choice(B, min(B, 96), replace=False)]))
        ti = torch.from_numpy(idx).cuda()
        for train, jvp in ((True, False), (True, True), (False, False)):
            mode = cnf.TrainMode() if train else cnf.TestMode()
            cfg.use_jvp = jvp                                   # (JVP compute mode, src/icnf.jl:384-420: ndot = |J eps|)
            ic = make_icnf(cnf, cfg, jvp=jvp, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
            logpx, (E, n, A) = cnf.inference(ic, mode, _dev(xs), flat, {}, eps=_dev(eps) if train else None)
            assert (ic.last_stats["launches"] <= 3) == one, (name, B, train, ic.last_stats)
            assert ic.last_stats["nf"] == 1 + 6 * 8 and ic.last_stats["kernel_used"] == _lib.KERNEL_MFMA
            _, ref_lp, ref_regs, _ = O.inference(cfg, f64(flat), f64(xs[:, idx]), f64(eps[:, idx]) if train else None, train,
                                                 dt=1 / 8, adaptive=False)
            assert_parity(logpx[ti].cpu().numpy(), ref_lp, f"bcast {name} logpx train={train} jvp={jvp} B={B}")
            if train:
                assert_parity(torch.stack([E, n, A])[:, ti].cpu().numpy(), np.stack(ref_regs), f"bcast {name} regs jvp={jvp} B={B}")
            else:
                assert float(E.abs().max()) == 0.0 and float(n.abs().max()) == 0.0
                assert_parity(A[ti].cpu().numpy(), ref_regs[2], f"bcast {name} TestMode A B={B}")
            prob = cnf.inference_prob(ic, mode, _dev(xs), flat, {}, eps=_dev(eps) if train else None)
            fsol = cnf.base_sol(ic, prob).view()
            assert (prob.stats["launches"] <= 3) == one
            u0 = O.inference_u0(cfg, xs[:, idx], train)
            ref, _ = O.tsit5_solve(cfg.rhs(f64(flat), f64(eps[:, idx]) if train else None, train), f64(u0), *cfg.tspan, dt=1 / 8, adaptive=False)
            assert_parity(fsol[:, ti].cpu().numpy(), ref, f"bcast {name} fsol train={train} jvp={jvp} B={B}", trace_row=cfg.n_in)
            assert ic.solve_fallbacks() == 0
            ic.close()
        cfg.use_jvp = False
        # adaptive, TrainMode
        tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
        ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=tol)
        prob = cnf.inference_prob(ic, cnf.TrainMode(), _dev(xs), flat, {}, eps=_dev(eps))
        fsol = cnf.base_sol(ic, prob).view()
        st = prob.stats
        assert (st["launches"] <= 3) == one and st["nf"] == 2 + 6 * (st["naccept"] + st["nreject"]) and abs(st["t_final"] - cfg.tspan[1]) < 1e-6
        u0 = O.inference_u0(cfg, xs[:, idx], True)
        ref64, _ = O.tsit5_solve(cfg.rhs(f64(flat), f64(eps[:, idx]), True), f64(u0), *cfg.tspan, reltol=1e-10, abstol=1e-10)
        assert_parity(fsol[:, ti].cpu().numpy(), ref64, f"bcast {name} adaptive vs float64 B={B}", rtol=5e-3, trace_row=cfg.n_in)
        ic.close()


def test_config5_shape_conditional_model_on_the_one_launch_solve():
    """VERDICT round 4, item 2 (second half): a CONDITIONAL model of config 5's shape -- CondRNODE, nn(vcat(z, ys)) with
    Dense(128 + n_cond => 384, tanh), Dense(384 => 128, tanh) (src/layers/cond_layer.jl:7-9, src/base_icnf.jl:288-309) -- runs on
    k_solve_bcast: the conditioning columns of W1 enter as a per-sample first-layer bias staged in LDS.  Fixed dt strictly against
    the float64 oracle (TrainMode VJP and TestMode), route asserted by the launch count."""
    n_cond = 5
    nvars, naugs = 64, 64
    n_in = nvars + naugs
    net = O.Net((n_in + n_cond, 384, n_in), (O.ACT_TANH,) * 2)
    cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, 1e-2)
    one = _one_launch_expected() and os.environ.get("CNF_BCAST") != "0"
    f64 = lambda a: a.astype(np.float64)
    for B in (2048, 999):
        rng = np.random.default_rng(1750 + B)
        flat = O.glorot_params(net, rng, np.float32, 0.1)
        xs = rng.standard_normal((nvars, B)).astype(np.float32)
        ys = rng.standard_normal((n_cond, B)).astype(np.float32)
        eps = rng.standard_normal((n_in, B)).astype(np.float32)
        idx = np.unique(np.concatenate([np.arange(min(B, 16)), np.arange(max(0, B - 24), B), rng.choice(B, 64, replace=False)]))
        ti = torch.from_numpy(idx).cuda()
        nn = cnf.Chain(cnf.Dense(n_in + n_cond, 384, "tanh"), cnf.Dense(384, n_in, "tanh"))
        for train in (True, False):
            mode = cnf.TrainMode() if train else cnf.TestMode()
            ic = cnf.construct(cnf.CondRNODE, nn, nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode("mfma"), lambda1=1e-2, lambda2=1e-2,
                               lambda3=1e-2, sol_kwargs=dict(adaptive=False, dt=1 / 8))
            logpx, (E, n, A) = cnf.inference(ic, mode, _dev(xs), _dev(ys), flat, {}, eps=_dev(eps) if train else None)
            assert (ic.last_stats["launches"] <= 3) == one, (B, train, ic.last_stats)
            _, ref_lp, ref_regs, _ = O.inference(cfg, f64(flat), f64(xs[:, idx]), f64(eps[:, idx]) if train else None, train,
                                                 dt=1 / 8, adaptive=False, ys=f64(ys[:, idx]))
            assert_parity(logpx[ti].cpu().numpy(), ref_lp, f"bcast cond logpx train={train} B={B}")
            if train:
                assert_parity(torch.stack([E, n, A])[:, ti].cpu().numpy(), np.stack(ref_regs), f"bcast cond regs B={B}")
            ic.close()


def test_testmode_headline_network_is_three_launches():
    """TestMode (`logpdf` of the README, README.md:98; exact trace, src/icnf.jl:148-164) on the
    32-128-128-32 network at B = 8192: u0 assembly, the WHOLE adaptive solve (k_trace3s<SOLVE>: k1, the automatic initial dt,
    every step attempt with its six stages, norm and controller; the workgroups meet once per norm) and the post-processing
    = three launches (round 3: 15).  Values equal the streamed route's to the solver tolerance (same kernel body, same
    controller; CNF_TRACE_SOLVE=0 runs this test on the streamed route in test_ab_switches' child) and sampled columns
    match the float64 oracle."""
    cfg, _, _ = O.baseline_cfg(3)
    rng = np.random.default_rng(1800)
    B = 8192
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    tol = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    ic = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=tol)
    logpx, (E, n, A) = cnf.inference(ic, cnf.TestMode(), _dev(xs), flat, {})
    st = ic.last_stats
    one = _one_launch_expected() and os.environ.get("CNF_TRACE_SOLVE") != "0" and os.environ.get("CNF_TRACE_FP32") != "1" \
        and os.environ.get("CNF_TRACE_GENERIC") != "1" and os.environ.get("CNF_TRACE_UNFUSED") != "1"
    assert (st["launches"] <= 3) == one, st
    assert st["nf"] == 2 + 6 * (st["naccept"] + st["nreject"]) and abs(st["t_final"] - 1.0) < 1e-6
    idx = rng.choice(B, 48, replace=False)
    f64 = lambda a: a.astype(np.float64)
    _, ref_lp, _, _ = O.inference(cfg, f64(flat), f64(xs[:, idx]), None, False, reltol=1e-10, abstol=1e-10)
    assert_parity(logpx[torch.from_numpy(idx).cuda()].cpu().numpy(), ref_lp, "TestMode headline network, one-launch solve vs float64", rtol=5e-3)
    # fixed dt: strict
    ic2 = make_icnf(cnf, cfg, kernel="mfma", sol_kwargs=dict(adaptive=False, dt=1 / 8))
    lp2, _ = cnf.inference(ic2, cnf.TestMode(), _dev(xs), flat, {})
    assert (ic2.last_stats["launches"] <= 3) == one and ic2.last_stats["nf"] == 1 + 6 * 8
    _, ref2, _, _ = O.inference(cfg, f64(flat), f64(xs[:, idx]), None, False, dt=1 / 8, adaptive=False)
    assert_parity(lp2[torch.from_numpy(idx).cuda()].cpu().numpy(), ref2, "TestMode headline network, one-launch solve, fixed dt")
    ic.close(); ic2.close()


def test_wave_local_solve_ab_route():
    """CNF_WAVE=0 in a child process: the same tests on k_mfma behind the streamed driver (the route the wave kernel
    replaced) -- both routes stay parity-green against the same oracle."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CNF_WAVE="0", CNF_NO_PARITY_REPORT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "test_wave_local_solve_small_networks"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-2000:], r.stderr[-500:])


def test_training_trajectories_of_device_and_oracle_gradients_agree():
    """the README's augmented model (naugs = nvars) from one initial point with the same
    mini-batches, probes and steered end times, 40 Lion steps with the gradient of cnf_loss_grad and 40 with the float64
    oracle's: the gradients agree at equal parameters and the loss trajectories track each other (Lion moves every
    parameter by +-eta: a sign flip of a near-zero gradient entry is the only way the two can part) -- whatever the
    objective does in the long run, both gradients do it (tests/train_trajectory_check.py writes the long comparison)."""
    from tests.train_trajectory_check import run
    res = run(steps=40, eta=1e-3)
    # at equal parameters: against the oracle on the device's own steps (the same discrete map) the 1e-4 bar of the gradient
    # tests; against the oracle's own adaptive float64 solve the two differ by the discretisation (measured <= 2e-2)
    assert max(res["grad_rel_diff_replay"]) < 2e-4, res["grad_rel_diff_replay"]
    assert max(res["grad_rel_diff"]) < 5e-2, res["grad_rel_diff"]
    h, o, pd = np.array(res["hip"]), np.array(res["oracle"]), np.array(res["param_diff"])
    # Lion moves every parameter by +-eta: the two runs hold IDENTICAL parameters until a near-zero gradient entry takes
    # different signs in the two (then they part by 2 eta in that entry and drift).  On the common prefix the losses agree
    # to rounding; the prefix is long (all 300 steps in profiles/round3_train_trajectory_hip_vs_oracle.json)
    same = pd == 0.0
    n_same = int(np.argmin(same)) if not same.all() else len(same)
    assert n_same >= 10, pd
    assert np.all(np.isfinite(h)) and np.max(np.abs(h[:n_same] - o[:n_same])) < 1e-3 * max(1.0, np.abs(o).max()), (h[:n_same], o[:n_same])
    assert np.max(np.abs(h - o)) < 0.1 * np.abs(o).max()                 # and afterwards they stay close
