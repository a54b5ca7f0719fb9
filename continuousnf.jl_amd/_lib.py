"""ctypes binding of libcnfhip.so (include/cnfhip.h).  The library is the product; this
module only loads it and declares the signatures.  There is no CPU fallback: if the
library is missing, or no gfx950 device is visible when a handle is created, the call
raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CNFHIP_LIB") or os.path.join(_HERE, "libcnfhip.so")   # CNFHIP_LIB: A/B builds
CSRC = os.path.join(_HERE, "csrc")

# enums (include/cnfhip.h)
OK, ERR_BAD_ARG, ERR_BAD_SHAPE, ERR_HIP, ERR_NO_DEVICE, ERR_MAXITERS, ERR_UNSUPPORTED, \
    ERR_NO_PARAMS, ERR_NONFINITE, ERR_RCCL = range(10)
MODE_TEST, MODE_TRAIN = 0, 1
AD_VJP, AD_JVP = 0, 1
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_MFMA = 0, 1, 2
ACT = {"identity": 0, "tanh": 1, "sigmoid": 2, "softplus": 3, "relu": 4, "swish": 5, "elu": 6}


class cnf_config(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dims", C.POINTER(C.c_int32)),
                ("acts", C.POINTER(C.c_int32)), ("nvars", C.c_int32), ("naugs", C.c_int32),
                ("ad", C.c_int32), ("lambda1", C.c_float), ("lambda2", C.c_float),
                ("lambda3", C.c_float), ("device", C.c_int32), ("n_cond", C.c_int32)]


class cnf_solve_opts(C.Structure):
    _fields_ = [("t0", C.c_float), ("t1", C.c_float), ("abstol", C.c_float),
                ("reltol", C.c_float), ("dt", C.c_float), ("adaptive", C.c_int32),
                ("maxiters", C.c_int32), ("kernel", C.c_int32)]


class cnf_solve_stats(C.Structure):
    _fields_ = [("nf", C.c_int32), ("naccept", C.c_int32), ("nreject", C.c_int32),
                ("t_final", C.c_float), ("dt_last", C.c_float), ("kernel_used", C.c_int32),
                ("launches", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_fp = C.c_void_p  # device or host float*; passed as raw addresses
shard_reduce_fn = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p)
_SIGNATURES = {
    "cnf_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(cnf_config)]),
    "cnf_destroy": (C.c_int, [C.c_void_p]),
    "cnf_set_params_host": (C.c_int, [C.c_void_p, _fp, C.c_size_t]),
    "cnf_set_params": (C.c_int, [C.c_void_p, _fp, C.c_size_t, C.c_void_p]),
    "cnf_set_cond": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_void_p]),
    "cnf_set_cond_host": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "cnf_set_shard_reduce": (C.c_int, [C.c_void_p, shard_reduce_fn, C.c_void_p]),
    "cnf_rhs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, C.c_void_p]),
    "cnf_rhs_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int]),
    "cnf_solve_tsit5": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp, C.c_int,
                                  C.POINTER(cnf_solve_opts), C.POINTER(cnf_solve_stats), C.c_void_p]),
    "cnf_solve_tsit5_host": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp, C.c_int,
                                       C.POINTER(cnf_solve_opts), C.POINTER(cnf_solve_stats)]),
    "cnf_build_u0": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, C.c_int, C.c_void_p]),
    "cnf_inference_post": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp, C.c_int, C.c_void_p]),
    "cnf_inference": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_int,
                                C.POINTER(cnf_solve_opts), C.POINTER(cnf_solve_stats), C.c_void_p]),
    "cnf_inference_sums": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_int,
                                     C.POINTER(cnf_solve_opts), C.POINTER(cnf_solve_stats), C.c_void_p]),
    "cnf_inference_host": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_int,
                                     C.POINTER(cnf_solve_opts), C.POINTER(cnf_solve_stats)]),
    "cnf_inference_submit": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_int,
                                       C.POINTER(cnf_solve_opts), C.c_void_p]),
    "cnf_inference_collect": (C.c_int, [C.c_void_p, C.POINTER(cnf_solve_stats)]),
    "cnf_inference_pending": (C.c_int, [C.c_void_p]),
    "cnf_loss_sums": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, _fp, C.c_void_p]),
    "cnf_loss_from_sums": (C.c_int, [C.c_void_p, C.c_int, _fp, C.POINTER(C.c_float)]),
    "cnf_status_string": (C.c_char_p, [C.c_int]),
    "cnf_last_error": (C.c_char_p, [C.c_void_p]),
    "cnf_loss_grad": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.POINTER(cnf_solve_opts),
                                C.POINTER(C.c_float), _fp, C.POINTER(cnf_solve_stats), C.c_void_p]),
    "cnf_loss_grad_host": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.POINTER(cnf_solve_opts),
                                     C.POINTER(C.c_float), _fp, C.POINTER(cnf_solve_stats)]),
    "cnf_loss_grad_test": (C.c_int, [C.c_void_p, _fp, C.c_int, C.POINTER(cnf_solve_opts), C.POINTER(C.c_float), _fp,
                                     C.POINTER(cnf_solve_stats), C.c_void_p]),
    "cnf_loss_grad_test_host": (C.c_int, [C.c_void_p, _fp, C.c_int, C.POINTER(cnf_solve_opts), C.POINTER(C.c_float), _fp,
                                          C.POINTER(cnf_solve_stats)]),
    "cnf_loss_grad_submit": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, C.c_int, C.POINTER(cnf_solve_opts), _fp, _fp, C.c_void_p]),
    "cnf_loss_grad_collect": (C.c_int, [C.c_void_p, C.POINTER(cnf_solve_stats)]),
    "cnf_set_params_async": (C.c_int, [C.c_void_p, _fp, C.c_size_t, C.c_void_p]),
    "cnf_grad_steps": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "cnf_solve_kernel_time": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "cnf_abi_version": (C.c_int, []),
    "cnf_state_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "cnf_kernel_for": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cnf_rhs_work": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cnf_comm_unique_id": (C.c_int, [C.c_char_p]),
    "cnf_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "cnf_comm_destroy": (C.c_int, [C.c_void_p]),
    "cnf_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "cnf_comm_allreduce": (C.c_int, [C.c_void_p, _fp, C.c_size_t, C.c_void_p]),
    "cnf_comm_device_key": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "cnf_comm_last_error": (C.c_char_p, []),
    "cnf_comm_library": (C.c_char_p, []),
    "cnf_loss_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, _fp, C.c_void_p]),
    "cnf_set_shard_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cnf_set_step_trace": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "cnf_grad_x": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_void_p]),
    "cnf_solve_fallbacks": (C.c_int, [C.c_void_p]),
    "cnf_set_solve_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cnf_set_grad_split": (C.c_int, [C.c_int]),
    "cnf_selftest_hold_cus": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "cnf_selftest_split_product": (C.c_int, [_fp, _fp, _fp, C.c_int]),
}
COMM_ID_BYTES = 128
EXPORTS = tuple(_SIGNATURES)

_lib = None


class CNFError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"libcnfhip status {status}: {detail}")


def build(verbose=False):
    """Compile libcnfhip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout + r.stderr)
    if r.returncode:
        raise RuntimeError("building libcnfhip.so failed")
    return LIB_PATH


def lib():
    """Load libcnfhip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` or "
                f"`make -C {CSRC}`.  There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; the dynamic
        # loader shares one HIP runtime per soname, so torch's copy has to be the one that
        # gets loaded (a system runtime loaded first, paired with torch's HSA, sees no
        # device).  Without torch the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


def check(status, handle=None):
    if status != OK:
        l = lib()
        detail = l.cnf_status_string(status).decode()
        if handle:
            extra = l.cnf_last_error(handle).decode()
            if extra:
                detail += ": " + extra
        raise CNFError(status, detail)
