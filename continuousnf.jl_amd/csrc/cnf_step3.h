// k_step3: the Tsit5 step kernel of the headline shape 32 -> 128 -> 128 -> 32, tanh (cnf_step3.hip).
#pragma once
#include "cnf_mfma_dev.h"

size_t step3_img_floats();
// register-fragment weight image from the flat parameter vector
void step3_pack(const NetDesc& nd, const float* d_params, float* d_img3, hipStream_t s);
// one step attempt (MfmaArgs as for k_mfma mode 2; no cond, no dump, TrainMode)
void step3_launch(const MfmaArgs& a, const float* d_img3, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s);
// the same for the JVP compute mode (k_step3j); also exact for VJP handles without the |eps^T J| row (norm_j == 0):
// ldot = -eps.(J eps) = -(eps^T J).eps and zdot do not depend on the mode
void step3j_launch(const MfmaArgs& a, const float* d_img3, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s);
