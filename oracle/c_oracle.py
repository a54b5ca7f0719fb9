"""ctypes wrapper of oracle/liboracle.so (the float32 C restatement).  Test/bench
infrastructure only -- see the header of cnf_oracle.c."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


class oc_net(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("dims", C.c_int * 9), ("acts", C.c_int * 8),
                ("nvars", C.c_int), ("naugs", C.c_int), ("norm_z", C.c_int),
                ("norm_j", C.c_int), ("norm_z_aug", C.c_int), ("jvp", C.c_int)]


class oc_stats(C.Structure):
    _fields_ = [("nf", C.c_int), ("naccept", C.c_int), ("nreject", C.c_int),
                ("t_final", C.c_float), ("dt_last", C.c_float)]


def build():
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("building liboracle.so failed:\n" + r.stdout + r.stderr)
    return _PATH


_NATIVE = os.path.join(_HERE, "liboracle_native.so")
_flags = "-march=x86-64-v3"


def use_native():
    """bench.py's CPU baseline only: when the host has AVX-512, build the same source with -march=native on THIS host
    (16-sample blocks = one zmm register) and load that build instead of the portable x86-64-v3 one the tests pin.  Must be
    called before the library is first used; returns the -march flag in effect."""
    global _PATH, _flags
    if _lib is not None:
        return _flags
    try:
        has512 = any("avx512f" in line for line in open("/proc/cpuinfo") if line.startswith("flags"))
    except OSError:
        has512 = False
    if not has512:
        return _flags
    r = subprocess.run(["gcc", "-O3", "-march=native", "-ffp-contract=fast", "-fopenmp", "-fPIC", "-shared", "-o", _NATIVE,
                        os.path.join(_HERE, "cnf_oracle.c"), "-lm"], capture_output=True, text=True)
    if r.returncode == 0 and os.path.exists(_NATIVE):
        _PATH, _flags = _NATIVE, "-march=native (AVX-512 host)"
    return _flags


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.oc_threads.restype = C.c_int
        _lib.oc_solve_tsit5.restype = C.c_int
    return _lib


def make_net(cfg):
    """cfg: oracle.cnf_oracle.Cfg"""
    n = oc_net()
    n.n_layers = cfg.net.n_layers
    for i, d in enumerate(cfg.net.dims):
        n.dims[i] = d
    for i, a in enumerate(cfg.net.acts):
        n.acts[i] = a
    n.nvars, n.naugs = cfg.nvars, cfg.naugs
    n.norm_z, n.norm_j, n.norm_z_aug = int(cfg.lam1 != 0), int(cfg.lam2 != 0), int(cfg.lam3 != 0)
    n.jvp = int(cfg.use_jvp)
    return n


def _cm(x):
    """logical (rows, B) -> column-major float32 flat"""
    return np.ascontiguousarray(np.asarray(x).T, dtype=np.float32).reshape(-1)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def rhs(cfg, flat, u, eps, train):
    net = make_net(cfg)
    D, B = u.shape
    uf, pf = _cm(u), np.ascontiguousarray(flat, dtype=np.float32)
    ef = _cm(eps) if eps is not None else np.zeros(1, np.float32)
    du = np.empty(D * B, np.float32)
    lib().oc_rhs(C.byref(net), _p(pf), _p(uf), _p(ef), _p(du), C.c_int(B), C.c_int(int(train)))
    return du.reshape(B, D).T


def solve(cfg, flat, u0, eps, train, *, dt=0.0, adaptive=True, abstol=1e-6, reltol=1e-3,
          maxiters=100000, tspan=None, trace=None):
    """``trace``: a float32 array of shape (cap, 4) that receives (t, signed h, EEst, accepted) per step attempt."""
    net = make_net(cfg)
    if trace is not None:
        assert trace.dtype == np.float32 and trace.flags.c_contiguous and trace.shape[1] == 4
        lib().oc_set_trace(_p(trace), C.c_int(trace.shape[0]))
    D, B = u0.shape
    t0, t1 = tspan if tspan is not None else cfg.tspan
    uf, pf = _cm(u0), np.ascontiguousarray(flat, dtype=np.float32)
    ef = _cm(eps) if eps is not None else np.zeros(1, np.float32)
    out = np.empty(D * B, np.float32)
    st = oc_stats()
    rc = lib().oc_solve_tsit5(C.byref(net), _p(pf), _p(uf), _p(ef), _p(out), C.c_int(B),
                              C.c_int(int(train)), C.c_float(t0), C.c_float(t1), C.c_float(abstol),
                              C.c_float(reltol), C.c_float(dt), C.c_int(int(adaptive)),
                              C.c_int(maxiters), C.byref(st))
    if trace is not None:
        lib().oc_set_trace(None, C.c_int(0))
    if rc:
        raise RuntimeError(f"oc_solve_tsit5 rc={rc}")
    return out.reshape(B, D).T, {"nf": st.nf, "naccept": st.naccept, "nreject": st.nreject,
                                 "t_final": st.t_final, "dt_last": st.dt_last}


def post(cfg, fsol, train):
    net = make_net(cfg)
    D, B = fsol.shape
    ff = _cm(fsol)
    logpx = np.empty(B, np.float32)
    regs = np.empty(3 * B, np.float32)
    lib().oc_post(C.byref(net), _p(ff), _p(logpx), _p(regs), C.c_int(B), C.c_int(int(train)))
    return logpx, regs.reshape(3, B)


def threads():
    return lib().oc_threads()


def set_threads(n):
    """OpenMP threads of the following calls (the runtime may have been sized by another library of the process)."""
    lib().oc_set_threads(int(n))
