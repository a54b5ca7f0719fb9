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


@dataclass(frozen=True)
class PlanarLayer:
    """``PlanarLayer(nvars, activation; use_bias, n_cond)`` (src/layers/planar_layer.jl): ``u * act.(w' * z .+ b)`` -- a
    rank-one field.  To the kernels it is a two-layer MLP ``(nvars + n_cond) -> 1 -> nvars`` with ``W1 = w'``, ``b1 = b``,
    ``W2 = u``, identity output and a zero output bias that is not a parameter; ``Chain`` maps the Lux parameter order
    ``(u, w, b)`` (planar_layer.jl:43-57) to that layout and gradients back."""
    nvars: int
    activation: str = "identity"
    use_bias: bool = True
    n_cond: int = 0

    def __post_init__(self):
        if self.activation not in _lib.ACT:
            raise ValueError(f"unsupported activation {self.activation!r}; have {sorted(_lib.ACT)}")
        if self.nvars < 1 or self.n_cond < 0:
            raise ValueError("PlanarLayer sizes must be positive")


class Chain:
    def __init__(self, *layers):
        if not layers:
            raise ValueError("empty Chain")
        self.planar = layers[0] if isinstance(layers[0], PlanarLayer) else None
        if self.planar is not None:
            if len(layers) != 1:
                raise ValueError("a PlanarLayer stands alone in its Chain (as in the reference's tests)")
            p = self.planar
            layers = (Dense(p.nvars + p.n_cond, 1, p.activation), Dense(1, p.nvars, "identity"))
        for a, b in zip(layers[:-1], layers[1:]):
            if a.out_dims != b.in_dims:
                raise ValueError(f"layer size mismatch: {a.out_dims} -> {b.in_dims}")
        self.layers = tuple(layers)

    # ---- parameter layouts: `external` = what the caller holds (Lux order), `internal` = the MLP layout of the C ABI ----
    @property
    def n_params_internal(self):
        return sum(l.in_dims * l.out_dims + l.out_dims for l in self.layers)

    def to_internal(self, ps):
        """external -> internal (identity for Dense chains).  numpy arrays and torch tensors."""
        if self.planar is None:
            return ps
        p = self.planar
        nw = p.nvars + p.n_cond
        flat = ps.reshape(-1)
        if flat.shape[0] != self.n_params:
            raise ValueError(f"PlanarLayer has {self.n_params} parameters, got {flat.shape[0]}")
        u, w = flat[:p.nvars], flat[p.nvars:p.nvars + nw]
        if hasattr(flat, "new_zeros"):                       # torch
            import torch
            b = flat[p.nvars + nw:] if p.use_bias else flat.new_zeros(1)
            return torch.cat([w, b, u, flat.new_zeros(p.nvars)])
        b = flat[p.nvars + nw:] if p.use_bias else np.zeros(1, dtype=flat.dtype)
        return np.concatenate([w, b, u, np.zeros(p.nvars, dtype=flat.dtype)])

    def grad_to_external(self, g):
        """gradient in the internal layout -> the caller's layout (the zero output bias has no entry)."""
        if self.planar is None:
            return g
        p = self.planar
        nw = p.nvars + p.n_cond
        gw, gb, gu = g[:nw], g[nw:nw + 1], g[nw + 1:nw + 1 + p.nvars]
        parts = [gu, gw] + ([gb] if p.use_bias else [])
        if hasattr(g, "new_zeros"):
            import torch
            return torch.cat(parts)
        return np.concatenate(parts)

    @property
    def dims(self):
        return (self.layers[0].in_dims,) + tuple(l.out_dims for l in self.layers)

    @property
    def acts(self):
        return tuple(_lib.ACT[l.activation] for l in self.layers)

    @property
    def n_params(self):
        if self.planar is not None:                           # planar_layer.jl:59-61
            p = self.planar
            return p.nvars + (p.nvars + p.n_cond) + (1 if p.use_bias else 0)
        return sum(l.in_dims * l.out_dims + l.out_dims for l in self.layers)

    def __repr__(self):
        if self.planar is not None:
            p = self.planar
            return f"Chain(PlanarLayer({p.nvars}, {p.activation}; n_cond = {p.n_cond}))"
        return "Chain(" + ", ".join(f"Dense({l.in_dims} => {l.out_dims}, {l.activation})" for l in self.layers) + ")"


def setup(rng, nn: Chain, init: str = "glorot"):
    """``ps, st = Lux.setup(rng, nn); ps = ComponentArray(ps)`` (mlj_ext/core_icnf.jl:37-38):
    returns the flat Float32 vector -- per layer ``weight`` (out x in, column-major) then
    ``bias`` -- and an empty state.  Lux is not in /root/reference; two initialisations are offered:
    ``"glorot"`` (default): Glorot-uniform weights, zero biases (Lux 0.5's Dense default);
    ``"lux_v1"``: Lux >= 1.0's Dense default as remembered (unverifiable here): kaiming-uniform weights
    U(+-gain sqrt(3 / in)) with gain 5/3 for tanh and 1 otherwise, biases U(+-1 / sqrt(in)).
    The parameters are an INPUT of the hot path, so neither choice touches parity; it does decide how the
    reference's regression configuration trains (profiles/round4_training_ablation.md)."""
    if isinstance(rng, (int, np.integer)):
        rng = np.random.default_rng(int(rng))
    if nn.planar is not None:                                 # planar_layer.jl:43-57: (u, w, b), Glorot-uniform vectors, zero bias
        p = nn.planar
        nw = p.nvars + p.n_cond
        lim_u, lim_w = math.sqrt(6.0 / (2 * p.nvars)), math.sqrt(6.0 / (2 * nw))
        parts = [rng.uniform(-lim_u, lim_u, size=p.nvars), rng.uniform(-lim_w, lim_w, size=nw)]
        if p.use_bias:
            parts.append(np.zeros(1))
        return np.concatenate(parts).astype(np.float32), {}
    if init not in ("glorot", "lux_v1"):
        raise ValueError("init must be 'glorot' or 'lux_v1'")
    parts = []
    for l in nn.layers:
        if init == "glorot":
            lim = math.sqrt(6.0 / (l.in_dims + l.out_dims))
            parts.append(rng.uniform(-lim, lim, size=l.in_dims * l.out_dims))
            parts.append(np.zeros(l.out_dims))
        else:
            gain = 5.0 / 3.0 if l.activation == "tanh" else 1.0
            lim = gain * math.sqrt(3.0 / l.in_dims)
            parts.append(rng.uniform(-lim, lim, size=l.in_dims * l.out_dims))
            parts.append(rng.uniform(-1.0 / math.sqrt(l.in_dims), 1.0 / math.sqrt(l.in_dims), size=l.out_dims))
    return np.concatenate(parts).astype(np.float32), {}


@dataclass
class CondLayer:
    """``CondLayer(nn, ys)`` (src/layers/cond_layer.jl:1-9): ``nn(vcat(z, ys))``.  Built per call by
    ``inference_prob`` for the Cond* models (src/base_icnf.jl:302)."""
    nn: Chain
    ys: object
