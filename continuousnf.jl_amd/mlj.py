"""Training front end: the host-side mirror of the reference's MLJ adapter for this path
(src/exts/mlj_ext/core_icnf.jl: ``ICNFModel`` :1-29, ``fit`` :31-94, ``transform`` :96-123,
``fitted_params`` core.jl:1-4).  ``fit`` runs the reference's loop -- shuffled mini-batches,
``loss(icnf, TrainMode(), xs, ps, st)`` and its gradient per batch, one optimiser after the other
for ``n_epochs`` epochs each -- with the gradient coming from the device (cnf_loss_grad) instead of
Enzyme.  The optimisers are the two the reference names (Optimisers.jl ``Lion`` -- its default --
and ``Adam``), restated from their published update rules; they run on the device through torch
(plumbing: n_params-sized elementwise updates).

No MLJ machinery here (tables, machines): ``X`` is an ``n x nvars`` array, one observation per row,
as ``MLJModelInterface.matrix(X)`` yields (core_icnf.jl:32)."""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Tuple

import numpy as np

from .base_icnf import ICNF, inference, loss_and_grad, loss_and_grad_collect, loss_and_grad_submit
from .layers import setup
from .types import TestMode, TrainMode


@dataclass
class Lion:
    """Optimisers.Lion(eta = 0.001, beta = (0.9, 0.999)), the reference's default optimiser (core_icnf.jl:17).  Optimisers.jl
    is not in /root/reference, so the rule is restated, in two readings (``rule``):

    * ``"optimisers"`` (default: the reference trains with ``Optimisers.Lion``, so its rule is the one a drop-in must run) --
      the update as Optimisers.jl's source states it, from memory and unverifiable here: the state is refreshed FIRST as
      ``b2 * g + (1 - b2) * state`` (with b2 = 0.999 it is the current gradient to a part in a thousand) and the step is
      ``eta * sign((b2 - b1) * g + b1 * state)`` -- in effect sign-SGD;
    * ``"paper"`` (opt-in) -- Chen et al. 2023: x -= eta * sign(b1 m + (1 - b1) g), then m = b2 m + (1 - b2) g.

    The choice decides whether the reference's regression configuration trains at all (DESIGN 7.00,
    profiles/round4_training_ablation.md): the momentum form overshoots on the stiff tspan (0, 13) flow.

    ``apply(..., gate)``: a one-element device tensor (1.0 / 0.0); with 0 neither the parameters nor the state move -- a
    submitted gradient whose launch gave up (zeros, NaN loss) must not be a momentum-only step."""
    eta: float = 1e-3
    beta: Tuple[float, float] = (0.9, 0.999)
    rule: str = "optimisers"

    def init(self, ps):
        return {"m": ps.new_zeros(ps.shape)}

    def apply(self, state, ps, g, gate=None):
        import torch
        b1, b2 = self.beta
        m = state["m"]
        if self.rule == "optimisers":
            new_m = m * (1 - b2) + g * b2
            step = torch.sign(g * (b2 - b1) + new_m * b1)
        elif self.rule == "paper":
            step = torch.sign(m * b1 + g * (1 - b1))
            new_m = m * b2 + g * (1 - b2)
        else:
            raise ValueError("Lion.rule must be 'paper' or 'optimisers'")
        if gate is not None:                                          # (where, not a product: a gated-off gradient may hold NaN)
            on = gate > 0
            step = torch.where(on, step, torch.zeros_like(step))
            new_m = torch.where(on, new_m, m)
        ps.sub_(step, alpha=self.eta)
        m.copy_(new_m)


@dataclass
class Adam:
    """Optimisers.Adam(eta = 0.001, beta = (0.9, 0.999), epsilon = 1e-8)."""
    eta: float = 1e-3
    beta: Tuple[float, float] = (0.9, 0.999)
    epsilon: float = 1e-8

    def init(self, ps):
        # (the step count lives on the device so that a gated-off step -- see Lion.apply -- does not advance it)
        return {"m": ps.new_zeros(ps.shape), "v": ps.new_zeros(ps.shape), "t": ps.new_zeros(1)}

    def apply(self, state, ps, g, gate=None):
        b1, b2 = self.beta
        m, v, t = state["m"], state["v"], state["t"]
        new_t = t + 1
        new_m = m * b1 + g * (1 - b1)
        new_v = v * b2 + g * g * (1 - b2)
        mhat = new_m / (1 - b1 ** new_t)
        vhat = new_v / (1 - b2 ** new_t)
        step = mhat / (vhat.sqrt() + self.epsilon)
        if gate is not None:
            import torch
            on = gate > 0
            step = torch.where(on, step, torch.zeros_like(step))
            new_m, new_v, new_t = torch.where(on, new_m, m), torch.where(on, new_v, v), torch.where(on, new_t, t)
        ps.sub_(step, alpha=self.eta)
        m.copy_(new_m); v.copy_(new_v); t.copy_(new_t)


@dataclass
class ICNFModel:
    """src/exts/mlj_ext/core_icnf.jl:1-29 (same field names and defaults; ``adtype`` has no meaning
    here -- the derivative is the device adjoint)."""
    m: ICNF
    loss: Callable | None = None                            # the reference's second positional argument; None = `loss` (src/icnf.jl:481-490)
    optimizers: tuple = field(default_factory=lambda: (Lion(),))
    n_epochs: int = 300
    adtype: Any = None                                      # accepted, unused: the derivative is the device adjoint
    use_batch: bool = True
    batch_size: int = 32
    sol_kwargs: dict = field(default_factory=dict)
    callback: Callable[[int, float], Any] | None = None     # (iteration, loss) per batch; not in the reference
    init: str = "glorot"                                    # layers.setup: "glorot" or "lux_v1" (not in the reference: Lux.setup decides there)
    pipelined: bool = True                                  # submit the gradients (no host wait per iteration) where the backend can; not in the reference


def _device_matrix(icnf: ICNF, X):
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: the HIP backend has no CPU fallback")
    x = np.ascontiguousarray(np.asarray(X, dtype=np.float32).T)      # nvars x n (core_icnf.jl:32)
    if x.shape[0] != icnf.nvars:
        raise ValueError(f"X has {x.shape[0]} columns, the model has nvars = {icnf.nvars}")
    return torch.from_numpy(x).to(torch.device("cuda", icnf.device))


def fit(model: ICNFModel, verbosity: int, X, ys=None):
    """``MLJModelInterface.fit(model, verbosity, X)`` (core_icnf.jl:31-94).  Returns
    ``(fitresult, cache, report)`` with ``fitresult = (ps, st)``; ``ps`` is the flat Float32 vector as
    a numpy array.  ``ys`` (n x n_cond) for the conditional models (core_cond_icnf.jl)."""
    import torch
    icnf = model.m
    if model.loss is not None:
        from .base_icnf import loss as _builtin_loss
        if model.loss is not _builtin_loss:
            raise NotImplementedError("ICNFModel.loss: the device adjoint differentiates the package's own loss (src/icnf.jl:481-490)")
    x = _device_matrix(icnf, X)
    n = x.shape[1]
    y = None
    if icnf.cond:
        if ys is None:
            raise ValueError("conditional model: ys is required")
        y = torch.from_numpy(np.ascontiguousarray(np.asarray(ys, dtype=np.float32).T)).to(x.device)
    ps_host, st = setup(icnf.rng, icnf.nn, init=model.init)           # core_icnf.jl:37-38
    ps = torch.from_numpy(ps_host).to(x.device)
    bs = model.batch_size if model.use_batch else n                   # core_icnf.jl:47-53
    it = 0
    t0 = time.perf_counter()
    losses = []
    # Without a per-iteration callback nothing on the host needs an iteration's loss before the next one starts: the gradients
    # are SUBMITTED (loss_and_grad_submit: solve + adjoint enqueued, loss and gradient left on the device), the optimiser's
    # update and the next parameter upload are enqueued behind them on the same stream, and the host only ever waits for the
    # launch before the previous one -- the GPU goes from one gradient straight into the next.  Where the gradient does not run
    # in the launch of the solve (larger networks), or a submitted launch gives up, the loop below is the synchronous one.
    pipelined = model.callback is None and model.pipelined and x.is_cuda
    n_iter = len(model.optimizers) * model.n_epochs * ((n + bs - 1) // bs)
    losses_dev = torch.full((max(1, n_iter),), float("nan"), dtype=torch.float32, device=x.device)   # one scalar per iteration: no gradient buffer is kept
    pending = []                                                      # (loss index, batch indices) of the launches still in flight
    redo = []                                                         # batches whose launch gave up: run again synchronously
    was_pipelined = False
    from . import _lib as _L

    def fill_from_device():
        vals = losses_dev[:len(losses)].cpu().numpy()
        for i, v in enumerate(vals):
            if np.isnan(losses[i]) and not np.isnan(v):
                losses[i] = float(v)

    def sync_step(opt, state, idx, slot):
        xb = x[:, idx]
        args = (xb, y[:, idx], ps, st) if y is not None else (xb, ps, st)
        val, g = loss_and_grad(icnf, TrainMode(), *args)
        opt.apply(state, ps, g)
        losses[slot] = val
        return val

    def drain(keep, opt, state):
        nonlocal pipelined
        while len(pending) > keep:
            slot, idx = pending.pop(0)
            try:
                loss_and_grad_collect(icnf)
            except _L.CNFError as e:
                # Only "the launch gave up" (a wait ran out: CNF_ERR_UNSUPPORTED) is survivable: that launch delivered zeros and a
                # NaN loss, its update was gated off on the device (no parameter or momentum moved), the launches queued behind it
                # give up too (the abort word stays set until the host has collected) and are gated off the same way.  All of them
                # run again below, synchronously.  NaN states, MAXITERS and HIP errors are the caller's to see.
                if e.status != _L.ERR_UNSUPPORTED:
                    raise
                pipelined = False
                redo.append((slot, idx))
        if not pipelined:
            while redo and not pending:
                slot, idx = redo.pop(0)
                sync_step(opt, state, idx, slot)
    for opt in model.optimizers:                                      # core_icnf.jl:64-73
        state = opt.init(ps)
        for _epoch in range(model.n_epochs):
            perm = torch.from_numpy(icnf.rng.permutation(n)).to(x.device)     # shuffle = true
            for lo in range(0, n, bs):                                # partial = true
                idx = perm[lo:lo + bs]
                if pipelined:
                    xb = x[:, idx]
                    args = (xb, y[:, idx], ps, st) if y is not None else (xb, ps, st)
                    try:
                        lossd, g = loss_and_grad_submit(icnf, TrainMode(), *args)
                    except NotImplementedError:
                        pipelined = False
                if pipelined:
                    opt.apply(state, ps, g, gate=torch.isfinite(lossd).to(g.dtype))
                    losses_dev[len(losses)].copy_(lossd[0])
                    was_pipelined = True
                    pending.append((len(losses), idx))
                    losses.append(float("nan"))                       # filled in from the device (per epoch when printing, else at the end)
                    it += 1
                    drain(1, opt, state)
                    continue
                drain(0, opt, state)
                losses.append(float("nan"))
                val = sync_step(opt, state, idx, len(losses) - 1)
                it += 1
                if model.callback is not None:
                    model.callback(it, val)
            if verbosity > 0:
                k = max(1, (n + bs - 1) // bs)
                drain(0, opt, state)
                fill_from_device()
                print(f"epoch {_epoch + 1}/{model.n_epochs}: mean loss {np.nanmean(losses[-k:]):.5f}", flush=True)
        drain(0, opt, state)
    torch.cuda.synchronize(x.device)
    fill_from_device()
    report = {"stats": {"time": time.perf_counter() - t0, "iterations": it, "pipelined": was_pipelined}, "losses": np.asarray(losses)}
    return (ps.cpu().numpy(), st), None, report


def transform(model, fitresult, Xnew=None, ys=None):
    """core_icnf.jl:96-123: ``logp_x`` of every row of ``Xnew`` in TestMode (exact trace).  ``transform(mach, Xnew)`` /
    ``transform(mach, (Xnew, Ynew))`` for a fitted ``Machine``."""
    if isinstance(model, Machine):
        data = fitresult
        Xn, Yn = data if isinstance(data, tuple) else (data, None)
        return transform(model.model, model.fitresult, Xn, Yn)
    icnf = model.m
    ps, st = fitresult
    x = _device_matrix(icnf, Xnew)
    if icnf.cond:
        import torch
        y = torch.from_numpy(np.ascontiguousarray(np.asarray(ys, dtype=np.float32).T)).to(x.device)
        logpx, _ = inference(icnf, TestMode(), x, y, ps, st)
    else:
        logpx, _ = inference(icnf, TestMode(), x, ps, st)
    return logpx.cpu().numpy()


def fitted_params(_model, fitresult=None):
    """src/exts/mlj_ext/core.jl:1-4; ``fitted_params(mach)`` for a fitted ``Machine``."""
    if isinstance(_model, Machine):
        _model, fitresult = _model.model, _model.fitresult
    ps, st = fitresult
    return {"learned_parameters": ps, "states": st}


CondICNFModel = ICNFModel            # src/exts/mlj_ext/core_cond_icnf.jl: the same fields; the data are (X, Y)


class Machine:
    """The little of ``MLJBase.machine`` the reference's fit tests use (test/fit_tests.jl:160-200): ``machine(model, X)``
    or ``machine(model, (X, Y))`` for the conditional models, ``fit_(mach)`` (= ``fit!``), ``transform(mach, Xnew)``,
    ``fitted_params(mach)``, and ``ICNFDist(mach, mode)`` / ``CondICNFDist(mach, mode, ys)`` (src/exts/dist_ext/
    core_icnf.jl:8-11, core_cond_icnf.jl:9-16).  MLJ's tables and machine caches are not mirrored."""

    def __init__(self, model: ICNFModel, data):
        self.model = model
        self.X, self.Y = (data if isinstance(data, tuple) else (data, None))
        self.fitresult = None
        self.report = None


def machine(model: ICNFModel, data) -> Machine:
    return Machine(model, data)


def fit_(mach: Machine, verbosity: int = 0) -> Machine:
    mach.fitresult, _cache, mach.report = fit(mach.model, verbosity, mach.X, mach.Y)
    return mach


# ---------------------------------------------------------------------------------------
# Parameter files.  The reference stores ``ps`` with JLD2 (README.md:92-95),
# an HDF5 dialect this build cannot write; the flat vector is all the C ABI needs, so it is stored
# with the shape information that makes it self-checking:
#   bytes 0..3   "CNFP"      | u32 version = 1 | u32 n_layers | u32 dims[n_layers + 1] (dims[0] =
#   columns of the first weight = n_in + n_cond) | u32 acts[n_layers] (cnfhip.h codes) | u32 nvars |
#   u32 naugs | u32 n_cond | u64 n_params | f32 ps[n_params] (Lux order: per layer weight out x in
#   column-major, then bias).  Little endian throughout.
# ---------------------------------------------------------------------------------------
_MAGIC = b"CNFP"


def save_params(path, icnf: ICNF, ps):
    import struct
    if icnf.nn.planar is not None:
        raise NotImplementedError("CNFP files describe Dense chains; keep a PlanarLayer's (u, w, b) vector as it is")
    ps = ps.detach().cpu().numpy() if hasattr(ps, "detach") else np.asarray(ps)
    ps = np.ascontiguousarray(ps, dtype="<f4")
    dims, acts = icnf.nn.dims, icnf.nn.acts
    if ps.size != icnf.nn.n_params:
        raise ValueError("ps does not match the network")
    with open(path, "wb") as f:
        f.write(_MAGIC + struct.pack("<II", 1, len(acts)))
        f.write(struct.pack(f"<{len(dims)}I", *dims) + struct.pack(f"<{len(acts)}I", *acts))
        f.write(struct.pack("<IIIQ", icnf.nvars, icnf.naugmented, icnf.n_cond, ps.size))
        f.write(ps.tobytes())


def load_params(path, icnf: ICNF | None = None):
    """Returns the flat Float32 vector; with ``icnf`` given, checks that the file was written for the
    same network and variable split."""
    import struct
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != _MAGIC:
        raise ValueError("not a CNFP parameter file")
    ver, L = struct.unpack_from("<II", raw, 4)
    if ver != 1:
        raise ValueError(f"unsupported CNFP version {ver}")
    off = 12
    dims = struct.unpack_from(f"<{L + 1}I", raw, off); off += 4 * (L + 1)
    acts = struct.unpack_from(f"<{L}I", raw, off); off += 4 * L
    nvars, naugs, n_cond, n = struct.unpack_from("<IIIQ", raw, off); off += 20
    ps = np.frombuffer(raw, dtype="<f4", count=n, offset=off).astype(np.float32)
    if n != sum(i * o + o for i, o in zip(dims[:-1], dims[1:])) or len(raw) != off + 4 * n:
        raise ValueError("corrupt CNFP file: sizes do not add up")
    if icnf is not None and (tuple(dims) != tuple(icnf.nn.dims) or tuple(acts) != tuple(icnf.nn.acts) or
                             (nvars, naugs, n_cond) != (icnf.nvars, icnf.naugmented, icnf.n_cond)):
        raise ValueError("parameter file was written for a different model")
    return ps
