"""MI355X-native backend for the batched augmented-ODE right-hand side of
ContinuousNormalizingFlows.jl (log-density evaluation): hand-written HIP kernels behind a C
ABI (``libcnfhip.so``, ``include/cnfhip.h``) plus this thin host mirror of the reference's
``construct`` / ``inference`` / ``loss`` / ``augmented_f`` / ``ICNFDist`` surface."""
from . import _lib
from ._lib import CNFError, build
from .base_icnf import (ICNF, ODEProblem, base_sol, construct, generate, generate_prob, generate_sol,
                        inference, inference_prob,
                        inference_sol, loss, loss_and_grad, loss_from_sums, loss_sums, n_augment,
                        n_augment_input, steer_tspan)
from .dist import CondICNFDist, ICNFDist, logpdf, pdf, rand
from .icnf import augmented_f
from .layers import Chain, CondLayer, Dense, setup
from .types import (FFJORD, RNODE, CondFFJORD, CondPlanar, CondRNODE, HIPJacVecMatrixMode,
                    HIPMatrixMode, HIPVecJacMatrixMode, Planar, TestMode, TrainMode)
from . import mlj, parallel
from .mlj import Adam, ICNFModel, Lion, fit, fitted_params, load_params, save_params, transform

__all__ = [n for n in dir() if not n.startswith("_")]
