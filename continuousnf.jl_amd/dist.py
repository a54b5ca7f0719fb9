"""``ICNFDist`` (src/exts/dist_ext/core_icnf.jl:1-31): the Distributions-style front end of
log-density evaluation.  ``logpdf(d, A)`` is ``first(inference(d.m, d.mode, A, d.ps, d.st))``
(core_icnf.jl:27)."""
from __future__ import annotations

from typing import Any

import numpy as np

from .base_icnf import ICNF, _is_torch, generate, inference


def _from_machine(m):
    """``(icnf, ps, st)`` of a fitted machine (core_icnf.jl:8-11: ``fitted_params(mach)``, ``mach.model.m``)."""
    if m.fitresult is None:
        raise ValueError("the machine has not been fitted")
    ps, st = m.fitresult
    return m.model.m, ps, st


class ICNFDistribution:
    """src/exts/dist_ext/core.jl:3-16 (``ICNFDistribution <: ContinuousMultivariateDistribution``): ``length(d)`` is the
    number of variables, ``eltype(d)`` the model's float type."""

    def __len__(self):
        return self.m.nvars

    @property
    def eltype(self):
        return np.float32                                 # (the HIP compute modes are Float32: types.py)


class ICNFDist(ICNFDistribution):
    """``ICNFDist(icnf, mode, ps, st)`` or, from a fitted machine, ``ICNFDist(mach, mode)`` (core_icnf.jl:1-11)."""

    def __init__(self, m, mode, ps=None, st=None):
        if not isinstance(m, ICNF) and hasattr(m, "fitresult"):
            m, ps, st = _from_machine(m)
        self.m, self.mode, self.ps, self.st = m, mode, ps, st


class CondICNFDist(ICNFDistribution):
    """src/exts/dist_ext/core_cond_icnf.jl:1-16: ``CondICNFDist(icnf, mode, ys, ps, st)`` / ``CondICNFDist(mach, mode, ys)``."""

    def __init__(self, m, mode, ys, ps=None, st=None):
        if not isinstance(m, ICNF) and hasattr(m, "fitresult"):
            m, ps, st = _from_machine(m)
        self.m, self.mode, self.ys, self.ps, self.st = m, mode, ys, ps, st


def logpdf(d, A, *, eps=None):
    if not isinstance(d.m, ICNF):
        raise NotImplementedError("Not Implemented")      # core_icnf.jl:19
    vec = (A.dim() if _is_torch(A) else np.ndim(A)) == 1
    if vec:                                               # core_icnf.jl:13-21: hcat(x)
        A = A.reshape(-1, 1)
    if isinstance(d, CondICNFDist):                        # core_cond_icnf.jl:27-35: ys[:, 1:size(A, 2)]
        lp = inference(d.m, d.mode, A, d.ys[:, : A.shape[1]], d.ps, d.st, eps=eps)[0]
    else:
        lp = inference(d.m, d.mode, A, d.ps, d.st, eps=eps)[0]
    return lp[0] if vec else lp


def pdf(d: ICNFDist, A, *, eps=None):
    lp = logpdf(d, A, eps=eps)
    return lp.exp() if _is_torch(lp) else np.exp(lp)


def rand(d: ICNFDist, n: int = None, *, z0=None, eps=None):
    """``rand(d, n)`` (src/exts/dist_ext/core_icnf.jl:46-58): ``generate(d.m, d.mode, d.ps, d.st, n)``; ``rand(d)``: one
    draw, as a vector of nvars entries."""
    if not isinstance(d.m, ICNF):
        raise NotImplementedError("Not Implemented")
    if n is None:
        return rand(d, 1, z0=z0, eps=eps)[:, 0]
    if isinstance(d, CondICNFDist):
        return generate(d.m, d.mode, d.ps, d.st, n, ys=d.ys[:, :n], z0=z0, eps=eps)
    return generate(d.m, d.mode, d.ps, d.st, n, z0=z0, eps=eps)


def rand_(d: ICNFDist, A, *, z0=None, eps=None):
    """``rand!(d, A)`` (src/exts/dist_ext/core_icnf.jl:34-58, ``Distributions._rand!``): fills the vector (one draw) or the
    columns of the matrix ``A`` in place and returns it."""
    vec = (A.dim() if _is_torch(A) else np.ndim(A)) == 1
    draw = rand(d, None if vec else A.shape[1], z0=z0, eps=eps)
    if _is_torch(A):
        A.copy_(draw if _is_torch(draw) else type(A)(draw))
    else:
        A[...] = draw.cpu().numpy() if _is_torch(draw) else draw
    return A
