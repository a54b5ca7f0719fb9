"""Host-side mirror of the reference API: construct defaults, type switches, layouts,
argument checking.  No GPU needed."""
import numpy as np
import pytest

import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import base_icnf as B
from continuousnf.jl_amd.parallel import loss_from_global_sums, shard_range


def _nn(n_in):
    return cnf.Chain(cnf.Dense(n_in, 3 * n_in, "tanh"), cnf.Dense(3 * n_in, n_in, "tanh"))


def test_construct_defaults_follow_reference():
    # src/base_icnf.jl:28-38: RNODE -> lambda1 = lambda2 = 1e-2, lambda3 = 0; FFJORD -> all 0
    r = cnf.construct(cnf.RNODE, _nn(2), 1, 1)
    assert np.float32(r.lambda1) == np.float32(1e-2) and np.float32(r.lambda2) == np.float32(1e-2) and r.lambda3 == 0
    assert r.NORM_Z and r.NORM_J and not r.NORM_Z_AUG and r.AUGMENTED and not r.STEER
    f = cnf.construct(cnf.FFJORD, _nn(2), 2)
    assert (f.lambda1, f.lambda2, f.lambda3) == (0, 0, 0) and not f.AUGMENTED
    assert f.tspan == (0.0, 1.0) and not f.inplace                      # base_icnf.jl:13, 20
    assert isinstance(f.compute_mode, cnf.HIPVecJacMatrixMode)
    j = cnf.construct(cnf.RNODE, _nn(2), 1, 1, **{"λ₁": 0.5, "λ₂": 0.0, "λ₃": 2.0})
    assert (j.lambda1, j.lambda2, j.lambda3) == (0.5, 0.0, 2.0) and not j.NORM_J and j.NORM_Z_AUG


def test_planar_layer_is_a_two_layer_mlp_with_the_lux_parameter_order():
    """PlanarLayer (src/layers/planar_layer.jl): u * act.(w' z + b) = the MLP (nvars + n_cond) -> 1 -> nvars with W1 = w', b1 = b,
    W2 = u, identity output, zero output bias; parameters in Lux order (u, w, b), gradients mapped back."""
    for use_bias, n_cond in ((True, 0), (False, 0), (True, 3)):
        nn = cnf.Chain(cnf.PlanarLayer(4, "tanh", use_bias=use_bias, n_cond=n_cond))
        assert nn.dims == (4 + n_cond, 1, 4) and nn.n_params == 4 + 4 + n_cond + (1 if use_bias else 0)
        ps, st = cnf.setup(0, nn)
        assert ps.shape == (nn.n_params,) and st == {}
        ps = np.arange(1, nn.n_params + 1, dtype=np.float32)
        pi = nn.to_internal(ps)
        nw = 4 + n_cond
        assert pi.shape == (nn.n_params_internal,) == (nw + 1 + 4 + 4,)
        assert np.array_equal(pi[:nw], ps[4:4 + nw]) and np.array_equal(pi[nw + 1:nw + 5], ps[:4]) and not pi[nw + 5:].any()
        assert pi[nw] == (ps[-1] if use_bias else 0.0)
        g = nn.grad_to_external(np.arange(100, 100 + pi.size, dtype=np.float32))
        assert g.shape == ps.shape and np.array_equal(g[:4], 100 + nw + 1 + np.arange(4)) and np.array_equal(g[4:4 + nw], 100 + np.arange(nw))
        # the field itself against the closed form, through the float64 oracle's MLP on the internal layout
        from oracle import cnf_oracle as O
        net = O.Net(nn.dims, (O.ACT_TANH, O.ACT_IDENTITY))
        rg = np.random.default_rng(1)
        z, yc = rg.standard_normal((4, 5)), (rg.standard_normal((n_cond, 5)) if n_cond else None)
        u, w = 0.1 * ps[:4].astype(np.float64), 0.1 * ps[4:4 + nw].astype(np.float64)
        b = 0.1 * float(ps[-1]) if use_bias else 0.0
        zin = z if yc is None else np.vstack([z, yc])
        want = np.outer(u, np.tanh(w @ zin + b))
        out = O.mlp_forward(net, 0.1 * pi.astype(np.float64), z, yc)
        got = out[0] if isinstance(out, tuple) else out
        got = got[-1] if isinstance(got, list) else got
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    p = cnf.construct(cnf.Planar, cnf.Chain(cnf.PlanarLayer(2, "tanh")), 2)
    assert (p.lambda1, p.lambda2) == (0, 0) and not p.cond
    c = cnf.construct(cnf.CondPlanar, cnf.Chain(cnf.PlanarLayer(2, "tanh", n_cond=2)), 2)
    assert c.cond and c.n_cond == 2


def test_construct_rejects_what_is_out_of_scope_or_malformed():
    with pytest.raises(ValueError):
        cnf.construct(cnf.CondPlanar, _nn(2), 2)        # a conditional nn needs n_in + n_cond inputs
    with pytest.raises(ValueError):
        cnf.construct(cnf.CondRNODE, _nn(2), 2)
    with pytest.raises(NotImplementedError):
        cnf.construct(cnf.RNODE, _nn(2), 2, data_type=np.float64)
    with pytest.raises(ValueError):
        cnf.construct(cnf.RNODE, _nn(3), 1, 1)          # nn maps 3->3 but n_in = 2
    with pytest.raises(TypeError):
        cnf.construct(cnf.RNODE, _nn(2), 2, compute_mode="DIVecJacMatrixMode")
    with pytest.raises(TypeError):
        cnf.construct(cnf.RNODE, _nn(2), 2, bogus=1)
    with pytest.raises(NotImplementedError):            # src/base_icnf.jl:16-25: only the default base / eps distributions
        cnf.construct(cnf.RNODE, _nn(2), 2, basedist="Laplace")
    cnf.construct(cnf.RNODE, _nn(2), 2, basedist=None, epsdist=None)
    with pytest.raises(ValueError):
        cnf.Chain(cnf.Dense(2, 3), cnf.Dense(4, 2))
    with pytest.raises(ValueError):
        cnf.Dense(2, 3, "gelu")


def test_n_augment_and_modes():
    r = cnf.construct(cnf.RNODE, _nn(4), 2, 2)
    assert cnf.n_augment(r, cnf.TrainMode()) == 2 and cnf.n_augment(r, cnf.TestMode()) == 0
    assert cnf.n_augment_input(r) == 2
    assert cnf.n_augment_input(cnf.construct(cnf.RNODE, _nn(4), 4)) == 0
    with pytest.raises(TypeError):
        cnf.n_augment(r, "train")


def test_steer_tspan():
    r = cnf.construct(cnf.RNODE, _nn(2), 2, tspan=(0.0, 13.0), steer_rate=0.1, rng=3)
    assert cnf.steer_tspan(r, cnf.TestMode()) == (0.0, 13.0)        # base_icnf.jl:118-120
    ts = [cnf.steer_tspan(r, cnf.TrainMode())[1] for _ in range(200)]
    assert all(13.0 - 1.3 - 1e-4 <= t <= 13.0 + 1.3 + 1e-4 for t in ts) and np.std(ts) > 0.3
    r0 = cnf.construct(cnf.RNODE, _nn(2), 2, tspan=(0.0, 13.0))
    assert cnf.steer_tspan(r0, cnf.TrainMode()) == (0.0, 13.0)


def test_setup_layout_and_size():
    nn = cnf.Chain(cnf.Dense(2, 3, "tanh"), cnf.Dense(3, 2))
    ps, st = cnf.setup(0, nn)
    assert ps.dtype == np.float32 and ps.size == nn.n_params == 2 * 3 + 3 + 3 * 2 + 2 and st == {}
    assert np.all(ps[6:9] == 0) and np.all(ps[-2:] == 0)            # biases
    assert nn.dims == (2, 3, 2) and nn.acts == (1, 0)


def test_column_major_plumbing_roundtrip():
    x = np.arange(12, dtype=np.float64).reshape(3, 4)               # logical D x B
    b = B._as_colmajor(x, 3, "x")
    assert b.arr.dtype == np.float32 and b.rows == 3 and b.B == 4
    assert np.array_equal(b.arr[:3], x[:, 0])                       # column 0 contiguous
    assert np.array_equal(b.view(), x)
    with pytest.raises(ValueError):
        B._as_colmajor(x, 4, "x")
    with pytest.raises(ValueError):
        B._as_colmajor(np.zeros(3), None, "x")


def test_sol_kwargs_mapping():
    r = cnf.construct(cnf.RNODE, _nn(2), 2, sol_kwargs=dict(progress=True, save_everystep=False,
                      reltol=3e-4, abstol=1e-7, maxiters=2**31 - 1))
    o = B._solve_opts(r, (0.0, 1.0))
    assert o.adaptive == 1 and abs(o.reltol - 3e-4) < 1e-9 and o.dt == 0 and o.maxiters == 2**31 - 1
    r2 = cnf.construct(cnf.RNODE, _nn(2), 2, sol_kwargs=dict(adaptive=False, dt=1 / 64))
    o2 = B._solve_opts(r2, (0.0, 1.0))
    assert o2.adaptive == 0 and o2.dt == 1 / 64
    with pytest.raises(ValueError):
        B._solve_opts(cnf.construct(cnf.RNODE, _nn(2), 2, sol_kwargs=dict(adaptive=False)), (0.0, 1.0))
    with pytest.raises(TypeError):
        B._solve_opts(cnf.construct(cnf.RNODE, _nn(2), 2, sol_kwargs=dict(callback=1)), (0.0, 1.0))


def test_shard_range_partitions_columns():
    for Bn in (0, 1, 7, 8192, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_range(Bn, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == Bn
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_loss_from_global_sums_formula():
    s = np.array([-10.0, 2.0, 3.0, 4.0, 5.0])
    assert np.isclose(loss_from_global_sums(s, True, (0.1, 0.2, 0.3)), (10 + 0.2 + 0.6 + 1.2) / 5)
    assert np.isclose(loss_from_global_sums(s, False, (0.1, 0.2, 0.3)), 2.0)
    with pytest.raises(ValueError):
        loss_from_global_sums(np.zeros(5), True, (0, 0, 0))


def test_augmented_f_argument_count_dispatch():
    r = cnf.construct(cnf.RNODE, _nn(2), 2)
    with pytest.raises(TypeError):
        cnf.augmented_f(1, 2, 3)


def test_conditional_construct():
    nn = cnf.Chain(cnf.Dense(7, 9, "tanh"), cnf.Dense(9, 4, "tanh"))       # 4 = nvars + naugs, 3 conditioning rows
    c = cnf.construct(cnf.CondRNODE, nn, 2, 2)
    assert c.cond and c.n_cond == 3 and np.float32(c.lambda1) == np.float32(1e-2)      # base_icnf.jl:14, 28-37
    f = cnf.construct(cnf.CondFFJORD, nn, 4)
    assert f.cond and f.n_cond == 3 and f.lambda1 == 0
    with pytest.raises(TypeError):
        cnf.inference_prob(c, cnf.TrainMode(), np.zeros((2, 3), np.float32), np.zeros(nn.n_params, np.float32))


def test_param_file_roundtrip_and_optimisers(tmp_path):
    """CNFP parameter files (row f4) and the two optimiser update rules of the training front end."""
    import torch
    nn = cnf.Chain(cnf.Dense(5, 7, "tanh"), cnf.Dense(7, 3, "softplus"))
    icnf = cnf.construct(cnf.CondRNODE, nn, 2, 1, compute_mode=cnf.HIPVecJacMatrixMode())
    assert icnf.n_cond == 2
    ps, _ = cnf.setup(3, nn)
    f = tmp_path / "fitted.cnfp"
    cnf.save_params(f, icnf, ps)
    assert np.array_equal(cnf.load_params(f, icnf), ps)
    other = cnf.construct(cnf.RNODE, cnf.Chain(cnf.Dense(5, 7, "tanh"), cnf.Dense(7, 5, "tanh")), 3, 2,
                          compute_mode=cnf.HIPVecJacMatrixMode())
    with pytest.raises(ValueError):
        cnf.load_params(f, other)
    raw = f.read_bytes()
    (tmp_path / "cut.cnfp").write_bytes(raw[:-4])
    with pytest.raises(ValueError):
        cnf.load_params(tmp_path / "cut.cnfp")
    # Lion: the default is the rule of Optimisers.jl (the reference's optimiser, core_icnf.jl:17): state refreshed first with
    # weight b2 on the gradient, then the sign step; "paper" = Chen et al.: sign step with the interpolated momentum, then the
    # state refreshed with weight 1 - b2.  Adam: bias-corrected first step = eta * sign(g).
    g = torch.tensor([0.3, -0.1, 0.0])
    assert cnf.Lion().rule == "optimisers"
    for rule, m_w in (("optimisers", 0.999), ("paper", 0.001)):
        x = torch.tensor([1.0, -2.0, 0.5])
        o = cnf.Lion(eta=0.1, rule=rule)
        stt = o.init(x)
        o.apply(stt, x, g)
        assert torch.allclose(x, torch.tensor([0.9, -1.9, 0.5])) and torch.allclose(stt["m"], m_w * g)
        # second step with a small opposite gradient g' = -g / 500: the paper's rule still follows its momentum
        # (0.9 * 0.001 g outweighs 0.1 g'), Optimisers.jl's follows the new gradient (its state IS the new gradient)
        o.apply(stt, x, -g / 500)
        want = torch.tensor([1.0, -2.0, 0.5]) if rule == "optimisers" else torch.tensor([0.8, -1.8, 0.5])
        assert torch.allclose(x, want), (rule, x)
        # a gated-off step moves neither the parameters nor the state, whatever the gradient holds
        x0, m0 = x.clone(), stt["m"].clone()
        o.apply(stt, x, torch.tensor([float("nan"), 1.0, -1.0]), gate=torch.zeros(1))
        assert torch.equal(x, x0) and torch.equal(stt["m"], m0)
    x = torch.tensor([1.0, -2.0])
    o = cnf.Adam(eta=0.1)
    stt = o.init(x)
    o.apply(stt, x, torch.tensor([0.3, -0.1]))
    assert torch.allclose(x, torch.tensor([0.9, -1.9]), atol=1e-6)
    x0 = x.clone()
    o.apply(stt, x, torch.tensor([float("nan"), 0.0]), gate=torch.zeros(1))
    assert torch.equal(x, x0) and float(stt["t"]) == 1.0
    o.apply(stt, x, torch.tensor([0.3, -0.1]), gate=torch.ones(1))
    assert float(stt["t"]) == 2.0 and torch.allclose(x, torch.tensor([0.8, -1.8]), atol=1e-5)
