"""``augmented_f`` for the HIP compute modes: the methods a maintainer would add next to
src/icnf.jl:318-350 (out-of-place) and :352-382 (in-place) for
``ICNF{T, <:HIPMatrixMode}``.  One call = one fused launch over the whole batch."""
from __future__ import annotations

import numpy as np

from . import _lib
from .layers import CondLayer
from .base_icnf import (ICNF, _KERNEL, _as_colmajor, _empty_like, _mode_id, _stream,
                        n_augment, n_augment_input, raise_if_no_gpu)


def augmented_f(*args):
    """Out-of-place ``augmented_f(u, p, t, icnf, mode, nn, st, eps) -> du`` (src/icnf.jl:318-350,
    :384-420, :148-164) or in-place ``augmented_f(du, u, p, t, icnf, mode, nn, st, eps) -> None``
    (src/icnf.jl:352-382, :422-456, :166-184), chosen by the argument count as Julia's
    dispatch does.  ``t``, ``nn`` and ``st`` are accepted for signature parity and unused
    (the field is autonomous; the network lives in ``icnf``)."""
    if len(args) == 8:
        du = None
        u, p, _t, icnf, mode, _nn, _st, eps = args
    elif len(args) == 9:
        du, u, p, _t, icnf, mode, _nn, _st, eps = args
    else:
        raise TypeError("augmented_f takes 8 (out-of-place) or 9 (in-place) arguments")
    if not isinstance(icnf, ICNF):
        raise TypeError("icnf must come from construct(...)")
    m = _mode_id(mode)
    n_in = icnf.nvars + n_augment_input(icnf)
    D = n_in + 1 + n_augment(icnf, mode)
    ub = _as_colmajor(u, D, "u")
    B = ub.B
    icnf.set_params(p)
    # conditional models hand the conditioning input over inside the layer: CondLayer(nn, ys)
    icnf.set_cond(_nn.ys if isinstance(_nn, CondLayer) else None, B)
    eb = None
    if m == _lib.MODE_TRAIN:
        if eps is None:
            raise ValueError("TrainMode needs the probe matrix eps")
        eb = _as_colmajor(eps, n_in, "eps")
        if eb.B != B:
            raise ValueError("eps must have one column per sample")
        if (eb.torch is None) != (ub.torch is None):
            raise ValueError("u and eps must both be host arrays or both be GPU tensors")
    l, h = _lib.lib(), icnf.handle()
    k = _KERNEL[icnf.compute_mode.kernel]
    if du is not None:
        if tuple(du.shape) != (D, B):
            raise ValueError(f"du has shape {tuple(du.shape)}, expected {(D, B)}")
        # the in-place form writes where the caller's du is, when du is laid out as the reference keeps it (column-major
        # D x B: a sample's rows contiguous) -- no temporary, no copy; any other layout goes through one
        if ub.torch is not None and isinstance(du, ub.torch.Tensor) and du.is_cuda and du.dtype == ub.torch.float32 \
                and du.device == ub.arr.device and du.t().is_contiguous() and du.data_ptr() != ub.ptr:
            _lib.check(l.cnf_rhs(h, m, k, ub.ptr, eb.ptr if eb else None, du.data_ptr(), B, _stream(ub)), h)
            return None
        if ub.torch is None and isinstance(du, np.ndarray) and du.dtype == np.float32 and du.flags.f_contiguous \
                and du.flags.writeable and not np.shares_memory(du, ub.arr):
            _lib.check(l.cnf_rhs_host(h, m, k, ub.ptr, eb.ptr if eb else None, du.ctypes.data, B), h)
            return None
    out = _empty_like(ub, D, B)
    if ub.torch is not None:
        _lib.check(l.cnf_rhs(h, m, k, ub.ptr, eb.ptr if eb else None, out.ptr, B, _stream(ub)), h)
    else:
        _lib.check(l.cnf_rhs_host(h, m, k, ub.ptr, eb.ptr if eb else None, out.ptr, B), h)
    res = out.view()
    if du is None:
        return res
    if ub.torch is not None:
        du.copy_(res)
    else:
        du[...] = res
    return None
