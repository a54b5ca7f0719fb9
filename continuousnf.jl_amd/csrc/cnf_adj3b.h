// k_adj3b (cnf_adj3b.hip): the pullback of one Runge-Kutta step of the headline shape on split-bf16 products (k_adj3's work at
// the matrix rate of the forward kernels); same arguments and outputs as launch_adj_mfma_step for whole steps.
#pragma once
#include "cnf_grad.h"

bool adj3b_supported(const NetDesc& nd);          // 32-128-128-32 (padded), tanh, VJP handle, no conditioning; CNF_ADJ3B=0 switches it off
// d_img3b: the split-fragment image of k_step3b (MfmaPlan::d_img3b)
hipError_t launch_adj3b(const NetDesc& nd, const GradLayout& g, const void* d_img3b, const AdjStepArgs& S, hipStream_t s);
