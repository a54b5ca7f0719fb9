"""Fixtures produced by the reference itself (julia/make_fixtures.jl -> tests/golden/ref_*.bin): the only route from
"parity unpinned" to pinned parity.  Julia is not installed in the build container, so none are committed yet and
these tests skip; the day someone with Julia runs the script and commits its output they check

  * the flat parameter layout (``Vector(ComponentArray(Lux.setup(rng, icnf)[1]))`` = per layer weight, column-major, then bias),
  * ``augmented_f`` in TrainMode and TestMode (src/icnf.jl:318-350, :148-164),
  * a fixed-dt Tsit5 solve step for step, ``inference_sol``'s outputs (src/base_icnf.jl:167-189),
  * OrdinaryDiffEq's adaptive controller: ``sol.stats`` (nf, naccept, nreject) and the final state,

first for the CPU oracle (every run), then for the HIP path through the C ABI (``-m gpu``)."""
import glob
import os
import struct

import numpy as np
import pytest

from oracle import cnf_oracle as O
from tests.helpers import GOLDEN, assert_parity, make_icnf

REF_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_*.bin")))
_DT = {0: np.float32, 1: np.int32, 2: np.float64}


def read_cnfr(path):
    """The container julia/make_fixtures.jl writes: named arrays, column-major."""
    out = {}
    with open(path, "rb") as f:
        assert f.read(4) == b"CNFR"
        version, n = struct.unpack("<II", f.read(8))
        assert version == 1
        for _ in range(n):
            (ln,) = struct.unpack("<I", f.read(4))
            name = f.read(ln).decode()
            code, nd = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<" + "Q" * nd, f.read(8 * nd))
            dt = np.dtype(_DT[code])
            a = np.frombuffer(f.read(int(np.prod(dims)) * dt.itemsize), dtype=dt)
            out[name] = a.reshape(dims, order="F") if nd > 1 else a.copy()
    return out


def write_cnfr(path, arrays):
    """Same container from Python (used by the self-test of the reader below)."""
    inv = {np.dtype(v): k for k, v in _DT.items()}
    with open(path, "wb") as f:
        f.write(b"CNFR" + struct.pack("<II", 1, len(arrays)))
        for name, a in arrays.items():
            a = np.asarray(a)
            f.write(struct.pack("<I", len(name.encode())) + name.encode())
            f.write(struct.pack("<II", inv[a.dtype], a.ndim) + struct.pack("<" + "Q" * a.ndim, *a.shape))
            f.write(np.asfortranarray(a).tobytes(order="F"))


def _cfg(r):
    net = O.Net(tuple(int(d) for d in r["dims"]), tuple(int(a) for a in r["acts"]))
    lam = r["lambdas"]
    return O.Cfg(net, int(r["nvars"][0]), int(r["naugs"][0]), float(lam[0]), float(lam[1]), float(lam[2]), False,
                 (float(r["tspan"][0]), float(r["tspan"][1])))


def test_container_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    arrays = {"a": rng.standard_normal((3, 5)).astype(np.float32), "n": np.array([7, 8], dtype=np.int32),
              "s": np.array([1.5], dtype=np.float32)}
    p = str(tmp_path / "x.bin")
    write_cnfr(p, arrays)
    back = read_cnfr(p)
    assert all(np.array_equal(back[k], v) for k, v in arrays.items())


@pytest.mark.skipif(not REF_FILES, reason="no tests/golden/ref_*.bin: run julia/make_fixtures.jl with Julia + the package")
@pytest.mark.parametrize("path", REF_FILES)
def test_oracle_against_reference_fixtures(path):
    _check_oracle(path)


def _check_oracle(path):
    r = read_cnfr(path)
    cfg = _cfg(r)
    n_layers = len(cfg.net.dims) - 1
    # flat layout: per layer weight (out x in, column-major) then bias
    flat = np.concatenate([np.concatenate([r[f"W_{i}"].reshape(-1, order="F"), r[f"b_{i}"]]) for i in range(1, n_layers + 1)])
    assert np.array_equal(flat, r["ps_flat"]), "ComponentArray flat order differs from cnf_set_params' layout"
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    ps, eps, u0 = f64(r["ps_flat"]), f64(r["eps"]), f64(r["u0"])
    assert_parity(cfg.rhs(ps, eps, True)(u0), r["du_train"], "ref du train", trace_row=cfg.n_in)
    assert_parity(cfg.rhs(ps, None, False)(u0[: cfg.n_in + 1]), r["du_test"], "ref du test", trace_row=cfg.n_in)
    dt = float(r["dt"][0])
    for train, tag in ((True, "train"), (False, "test")):
        fsol, logpx, regs, st = O.inference(cfg, ps, f64(r["xs"]), eps if train else None, train, dt=dt, adaptive=False)
        assert st.nf == int(r[f"nf_fixed_{tag}"][0])
        assert_parity(fsol, r[f"fsol_fixed_{tag}"], f"ref fixed-dt fsol {tag}", trace_row=cfg.n_in)
        assert_parity(logpx, r[f"logpx_{tag}"], f"ref logpx {tag}")
        kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
        u0t = O.inference_u0(cfg, r["xs"], train)
        fa, sa = O.tsit5_solve(cfg.rhs(r["ps_flat"], r["eps"] if train else None, train), u0t, *cfg.tspan, **kw)
        nf, nacc, nrej = (int(x) for x in r[f"stats_adapt_{tag}"])
        # THE check of the controller law restated from memory
        assert (sa.nf, sa.naccept, sa.nreject) == (nf, nacc, nrej), "adaptive step sequence differs from OrdinaryDiffEq's"
        assert_parity(fa, r[f"fsol_adapt_{tag}"], f"ref adaptive fsol {tag}", rtol=5e-3, trace_row=cfg.n_in)


def test_the_checks_themselves_on_a_synthetic_fixture(tmp_path):
    """No Julia here: build a file of the same format from the float64 oracle and run the checks on it, so that the
    consumer of the real fixtures is known to work the day they arrive (this proves nothing about parity)."""
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(5)
    B, dt = 12, 1 / 8
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    Ws, bs = O.unflatten_params(cfg.net, flat)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    arrays = {"dims": np.array(cfg.net.dims, dtype=np.int32), "acts": np.array(cfg.net.acts, dtype=np.int32),
              "nvars": np.array([cfg.nvars], dtype=np.int32), "naugs": np.array([cfg.naugs], dtype=np.int32),
              "lambdas": np.array([cfg.lam1, cfg.lam2, cfg.lam3], dtype=np.float32),
              "tspan": np.array(cfg.tspan, dtype=np.float32), "ps_flat": flat, "xs": xs, "eps": eps,
              "dt": np.array([dt], dtype=np.float32), "u0": O.inference_u0(cfg, xs, True)}
    for i, (W, b) in enumerate(zip(Ws, bs), 1):
        arrays[f"W_{i}"], arrays[f"b_{i}"] = np.asarray(W, dtype=np.float32), np.asarray(b, dtype=np.float32)
    kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    for train, tag in ((True, "train"), (False, "test")):
        u0 = O.inference_u0(cfg, xs, train)
        e = eps if train else None
        arrays[f"du_{tag}"] = cfg.rhs(f64(flat), None if e is None else f64(e), train)(f64(u0)).astype(np.float32)
        fsol, logpx, regs, st = O.inference(cfg, f64(flat), f64(xs), None if e is None else f64(e), train, dt=dt, adaptive=False)
        arrays[f"fsol_fixed_{tag}"], arrays[f"nf_fixed_{tag}"] = fsol.astype(np.float32), np.array([st.nf], dtype=np.int32)
        arrays[f"logpx_{tag}"] = logpx.astype(np.float32)
        arrays[f"regs_{tag}"] = np.stack([np.zeros(B) if x is None else x for x in regs]).astype(np.float32)
        fa, sa = O.tsit5_solve(cfg.rhs(flat, e, train), u0, *cfg.tspan, **kw)
        arrays[f"fsol_adapt_{tag}"] = fa
        arrays[f"stats_adapt_{tag}"] = np.array([sa.nf, sa.naccept, sa.nreject], dtype=np.int32)
    p = str(tmp_path / "ref_synthetic.bin")
    write_cnfr(p, arrays)
    _check_oracle(p)


@pytest.mark.gpu
@pytest.mark.skipif(not REF_FILES, reason="no tests/golden/ref_*.bin: run julia/make_fixtures.jl with Julia + the package")
@pytest.mark.parametrize("path", REF_FILES)
def test_hip_path_against_reference_fixtures(path):
    import continuousnf.jl_amd as cnf
    r = read_cnfr(path)
    cfg = _cfg(r)
    dt = float(r["dt"][0])
    icnf = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=dt))
    du = cnf.augmented_f(r["u0"], r["ps_flat"], 0.0, icnf, cnf.TrainMode(), icnf.nn, {}, r["eps"])
    assert_parity(du, r["du_train"], "HIP vs reference du train", trace_row=cfg.n_in)
    dut = cnf.augmented_f(r["u0"][: cfg.n_in + 1], r["ps_flat"], 0.0, icnf, cnf.TestMode(), icnf.nn, {}, None)
    assert_parity(dut, r["du_test"], "HIP vs reference du test", trace_row=cfg.n_in)
    logpx, regs = cnf.inference(icnf, cnf.TrainMode(), r["xs"], r["ps_flat"], {}, eps=r["eps"])
    assert icnf.last_stats["nf"] == int(r["nf_fixed_train"][0])
    assert_parity(logpx, r["logpx_train"], "HIP vs reference logpx")
    assert_parity(np.stack(regs), r["regs_train"], "HIP vs reference regs")
    kw = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))
    ica = make_icnf(cnf, cfg, sol_kwargs=kw)
    prob = cnf.inference_prob(ica, cnf.TrainMode(), r["xs"], r["ps_flat"], {}, eps=r["eps"])
    fa = cnf.base_sol(ica, prob).view()
    nf, nacc, nrej = (int(x) for x in r["stats_adapt_train"])
    assert (prob.stats["nf"], prob.stats["naccept"], prob.stats["nreject"]) == (nf, nacc, nrej)
    assert_parity(fa, r["fsol_adapt_train"], "HIP vs reference adaptive fsol", rtol=5e-3, trace_row=cfg.n_in)
    icnf.close(); ica.close()
