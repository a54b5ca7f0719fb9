// k_step3: the Tsit5 step kernel of the headline shape 32 -> 128 -> 128 -> 32, tanh (cnf_step3.hip).
#pragma once
#include "cnf_mfma_dev.h"

size_t step3_img_floats();
// register-fragment weight image from the flat parameter vector
void step3_pack(const NetDesc& nd, const float* d_params, float* d_img3, hipStream_t s);
// one step attempt (MfmaArgs as for k_mfma mode 2; no cond, no dump, TrainMode)
// single: 0 = a step attempt; 1 / 2 = ONE evaluation, the first / second launch of the automatic initial dt
// (f(u) -> a.du with the norms of phase 0; f(u + h k1) -> a.Ks0 with the norm of phase 1; a.st, a.st_out, a.partials,
// a.ticket, a.n_total as for k_mfma's modes 0 / 1 with init_phase 0 / 1)
void step3_launch(const MfmaArgs& a, const float* d_img3, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single = 0);
// the same for the JVP compute mode (k_step3j); also exact for VJP handles without the |eps^T J| row (norm_j == 0):
// ldot = -eps.(J eps) = -(eps^T J).eps and zdot do not depend on the mode
void step3j_launch(const MfmaArgs& a, const float* d_img3, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single = 0);
// k_step3jb: k_step3j with every product formed from six bf16 MFMA terms on exactly split operands
size_t step3b_img_bytes();
void step3b_pack(const NetDesc& nd, const float* d_params, void* d_imgb, hipStream_t s);
void step3jb_launch(const MfmaArgs& a, const void* d_imgb, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single = 0);
// k_step3b: the VJP step kernel (k_step3) on six-term bf16 products
void step3b_launch(const MfmaArgs& a, const void* d_imgb, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single = 0);
// the whole solve of one shard in ONE launch (k_solve3b): grid = tiles of 32 columns, all resident
// (Solve3Args: cnf_mfma.h)
// (jvp: k_solve3jb, the JVP compute mode)
cnf_status step3b_solve_launch(const MfmaArgs& a, const void* d_imgb, int n_in, int norm_z, int norm_j, int grid, hipStream_t s,
                               const Solve3Args& sv, bool jvp, int device);
// workgroups of the one-launch solve kernels the device can hold at once (occupancy x CUs; 0: unknown device)
int step3b_solve_resident(bool jvp, bool record, int device);
// C = A Bt^T (16 x K each, K a multiple of 32) on the split-bf16 six-term product of the kernels above: arithmetic self-test
hipError_t split_product_test_launch(const float* dA, const float* dBt, float* dC, int K, hipStream_t s);
