"""The C-ABI library loads and exports every symbol include/cnfhip.h declares.  No compute
calls here (this suite runs without a GPU)."""
import ctypes
import os
import re

import pytest

import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "cnfhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cnf_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'`"
    assert os.path.dirname(_lib.LIB_PATH).startswith(ROOT)


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 18
    l = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(l, n), f"{n} declared in cnfhip.h but not exported"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)


def test_abi_version_and_status_strings():
    l = _lib.lib()
    assert l.cnf_abi_version() == 1
    assert l.cnf_status_string(0) == b"ok"
    assert b"shape" in l.cnf_status_string(_lib.ERR_BAD_SHAPE)


def test_grad_split_switch_is_a_plain_process_setting():
    """cnf_set_grad_split touches no device: it returns the mode that was in force and clamps what it is given."""
    l = _lib.lib()
    was = l.cnf_set_grad_split(1)
    try:
        assert was in (-1, 0, 1)
        assert l.cnf_set_grad_split(0) == 1
        assert l.cnf_set_grad_split(-7) == 0
        assert l.cnf_set_grad_split(5) == -1
        assert l.cnf_set_grad_split(-1) == 1
    finally:
        l.cnf_set_grad_split(was)


def test_create_validates_before_touching_the_device():
    l = _lib.lib()
    h = ctypes.c_void_p()
    dims = (ctypes.c_int32 * 3)(2, 6, 3)          # n_out != n_in
    acts = (ctypes.c_int32 * 2)(1, 1)
    cfg = _lib.cnf_config(2, dims, acts, 1, 1, 0, 0.0, 0.0, 0.0, 0)
    assert l.cnf_create(ctypes.byref(h), ctypes.byref(cfg)) == _lib.ERR_BAD_SHAPE
    assert l.cnf_create(None, ctypes.byref(cfg)) == _lib.ERR_BAD_ARG
    assert l.cnf_destroy(None) == _lib.ERR_BAD_ARG


def test_no_cpu_fallback_when_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    nn = cnf.Chain(cnf.Dense(2, 6, "tanh"), cnf.Dense(6, 2, "tanh"))
    icnf = cnf.construct(cnf.RNODE, nn, 1, 1)
    with pytest.raises(cnf.CNFError) as e:
        icnf.handle()
    assert e.value.status == _lib.ERR_NO_DEVICE
