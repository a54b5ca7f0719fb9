"""The BASELINE.json workloads as host-side models (SURVEY.md section 8 size table), the synthetic
inputs of SURVEY 8(d) and the Glorot initialisation -- what bench.py and the profiling tools build
their problems from.  (The oracle has its own copy of the table; nothing here touches it.)"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .base_icnf import construct
from .layers import Chain, Dense
from .types import FFJORD, RNODE, HIPJacVecMatrixMode, HIPVecJacMatrixMode, TestMode, TrainMode


@dataclass(frozen=True)
class Workload:
    name: str
    tag: type
    dims: tuple          # MLP widths, tanh on every layer (README.md:47)
    nvars: int
    naugs: int
    lambdas: tuple       # (l1, l2, l3)
    tspan: tuple
    batch: int
    train: bool          # TrainMode (Hutchinson) or TestMode (exact trace)

    @property
    def n_in(self):
        return self.nvars + self.naugs

    def mode(self):
        return TrainMode() if self.train else TestMode()


BASELINE = {
    1: Workload("RNODE 1+1, 2-6-2", RNODE, (2, 6, 2), 1, 1, (1e-2, 1e-2, 1e-2), (0.0, 13.0), 1024, True),
    2: Workload("RNODE 8+8, 16-48-16", RNODE, (16, 48, 16), 8, 8, (1e-2, 1e-2, 1e-2), (0.0, 1.0), 4096, True),
    3: Workload("RNODE 32, 32-128-128-32", RNODE, (32, 128, 128, 32), 32, 0, (1e-2, 1e-2, 0.0), (0.0, 1.0), 8192, True),
    4: Workload("FFJORD 32, 32-128-128-32", FFJORD, (32, 128, 128, 32), 32, 0, (0.0, 0.0, 0.0), (0.0, 1.0), 65536, True),
    5: Workload("RNODE 64+64, 128-384-128", RNODE, (128, 384, 128), 64, 64, (1e-2, 1e-2, 1e-2), (0.0, 1.0), 2048, False),
}

README_TOLERANCES = dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps))


def build(w: Workload, *, kernel="auto", jvp=False, sol_kwargs=None, rng=0):
    """The ICNF of a workload (compute mode = the HIP backend)."""
    nn = Chain(*[Dense(a, b, "tanh") for a, b in zip(w.dims[:-1], w.dims[1:])])
    cm = HIPJacVecMatrixMode(kernel) if jvp else HIPVecJacMatrixMode(kernel)
    return construct(w.tag, nn, w.nvars, w.naugs, compute_mode=cm, tspan=w.tspan, lambda1=w.lambdas[0],
                     lambda2=w.lambdas[1], lambda3=w.lambdas[2], sol_kwargs=dict(sol_kwargs or {}), rng=rng)


def glorot_params(dims, seed, bias_scale=0.0):
    """Flat Float32 vector, Lux order: Glorot-uniform weights (out x in, column-major), biases
    ``bias_scale * N(0, 1)`` (zero)."""
    rng = np.random.default_rng(seed) if not isinstance(seed, np.random.Generator) else seed
    parts = []
    for i, o in zip(dims[:-1], dims[1:]):
        lim = math.sqrt(6.0 / (i + o))
        parts.append(rng.uniform(-lim, lim, size=i * o))
        parts.append(bias_scale * rng.standard_normal(o))
    return np.concatenate(parts).astype(np.float32)


def synthetic_inputs(w: Workload, B, seed):
    """xs ~ N(0,1)^{nvars x B}, eps ~ N(0,1)^{n_in x B}, Float32."""
    rng = np.random.default_rng(seed)
    xs = rng.standard_normal((w.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((w.n_in, B)).astype(np.float32)
    return xs, eps
