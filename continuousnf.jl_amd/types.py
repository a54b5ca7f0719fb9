"""Dispatch tags, mirroring src/types.jl:1-35 and the model tag structs of src/icnf.jl:1-56.

The reference picks the AD product through the ``compute_mode`` type
(``DIVecJacMatrixMode`` / ``DIJacVecMatrixMode``, src/types.jl:17-23).  The HIP backend
plugs in at the same place with two new ``MatrixMode`` subtypes; nothing else of the
lattice is needed by the batched hot path (vector modes are out of scope)."""
from __future__ import annotations


class Mode:                      # src/types.jl:1
    pass


class TestMode(Mode):            # src/types.jl:2  -> exact trace (src/icnf.jl:148-164)
    __test__ = False             # not a pytest class
    cnf = 0


class TrainMode(Mode):           # src/types.jl:3  -> Hutchinson + regulariser rows (src/icnf.jl:318-350)
    cnf = 1


class ComputeMode:               # src/types.jl:5
    pass


class MatrixMode(ComputeMode):   # src/types.jl:7
    pass


class HIPMatrixMode(MatrixMode):
    """Fused gfx950 backend; ``kernel`` is 'auto' | 'generic' | 'mfma'."""
    ad = 0

    def __init__(self, kernel: str = "auto"):
        if kernel not in ("auto", "generic", "mfma"):
            raise ValueError(f"unknown kernel {kernel!r}")
        self.kernel = kernel

    def __repr__(self):
        return f"{type(self).__name__}({self.kernel!r})"


class HIPVecJacMatrixMode(HIPMatrixMode):
    """Counterpart of DIVecJacMatrixMode (src/types.jl:18-20): eps^T J by a reverse sweep;
    n-row = ||J^T eps|| (src/icnf.jl:342-343)."""
    ad = 0


class HIPJacVecMatrixMode(HIPMatrixMode):
    """Counterpart of DIJacVecMatrixMode (src/types.jl:21-23): J eps by a forward sweep;
    n-row = ||J eps|| (src/icnf.jl:412-413)."""
    ad = 1


# model tags: empty structs in the reference (src/icnf.jl:1-56), used by `construct` only to
# choose the lambda defaults (src/base_icnf.jl:28-37).
class AbstractICNF:
    pass


class RNODE(AbstractICNF):
    pass


class FFJORD(AbstractICNF):
    pass


class _OutOfScope(AbstractICNF):
    pass


class Planar(AbstractICNF):       # any field without the RNODE defaults; used with PlanarLayer (layers.py maps it onto the MLP kernels)
    pass


class CondRNODE(AbstractICNF):    # conditional models: nn(vcat(z, ys))
    pass


class CondFFJORD(AbstractICNF):
    pass


class CondPlanar(AbstractICNF):
    pass
