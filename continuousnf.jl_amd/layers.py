"""Minimal stand-ins for the Lux layer descriptors the reference passes to ``construct``
(``Chain(Dense(n_in => 3n_in, tanh), Dense(3n_in => n_in, tanh))``, README.md:47).  Only the
shape, the activation and the flat parameter layout matter to the hot path; the arithmetic
happens in libcnfhip."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass(frozen=True)
class Dense:
    in_dims: int
    out_dims: int
    activation: str = "identity"

    def __post_init__(self):
        if self.activation not in _lib.ACT:
            raise ValueError(f"unsupported activation {self.activation!r}; have {sorted(_lib.ACT)}")
        if self.in_dims < 1 or self.out_dims < 1:
            raise ValueError("Dense dims must be positive")


class Chain:
    def __init__(self, *layers: Dense):
        if not layers:
            raise ValueError("empty Chain")
        for a, b in zip(layers[:-1], layers[1:]):
            if a.out_dims != b.in_dims:
                raise ValueError(f"layer size mismatch: {a.out_dims} -> {b.in_dims}")
        self.layers = tuple(layers)

    @property
    def dims(self):
        return (self.layers[0].in_dims,) + tuple(l.out_dims for l in self.layers)

    @property
    def acts(self):
        return tuple(_lib.ACT[l.activation] for l in self.layers)

    @property
    def n_params(self):
        return sum(l.in_dims * l.out_dims + l.out_dims for l in self.layers)

    def __repr__(self):
        return "Chain(" + ", ".join(f"Dense({l.in_dims} => {l.out_dims}, {l.activation})" for l in self.layers) + ")"


def setup(rng, nn: Chain):
    """``ps, st = Lux.setup(rng, nn); ps = ComponentArray(ps)`` (mlj_ext/core_icnf.jl:37-38):
    returns the flat Float32 vector -- per layer ``weight`` (out x in, column-major) then
    ``bias`` -- and an empty state.  Glorot-uniform weights, zero biases."""
    if isinstance(rng, (int, np.integer)):
        rng = np.random.default_rng(int(rng))
    parts = []
    for l in nn.layers:
        lim = math.sqrt(6.0 / (l.in_dims + l.out_dims))
        parts.append(rng.uniform(-lim, lim, size=l.in_dims * l.out_dims))
        parts.append(np.zeros(l.out_dims))
    return np.concatenate(parts).astype(np.float32), {}


@dataclass
class CondLayer:
    """``CondLayer(nn, ys)`` (src/layers/cond_layer.jl:1-9): ``nn(vcat(z, ys))``.  Built per call by
    ``inference_prob`` for the Cond* models (src/base_icnf.jl:302)."""
    nn: Chain
    ys: object
