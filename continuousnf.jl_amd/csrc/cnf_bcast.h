// k_solve_bcast (cnf_bcast.hip): the whole Tsit5 solve of BASELINE config 5's network (two tanh layers, 64 < n_in <= 128,
// 256 < hidden <= 384) at eight samples per CU in one launch: v_mfma_f32_4x4x1_16B_f32 with the activations broadcast as the
// A operand, W1 resident in registers in both orientations, W2 streamed.  TrainMode / VJP and TestMode (closed-form trace).
#pragma once
#include "cnf_mfma.h"

size_t bcast_img_floats();
// the packed weight images (resident fragments, the W2 streams, C = W1 .* W2^T) from the flat parameter vector
void bcast_pack(const NetDesc& nd, const float* d_params, float* d_img, hipStream_t s);
// this network, compute mode and batch: any batch (beyond 8 columns per CU of `device` every workgroup carries several tiles and
// their Runge-Kutta rows live in a store in global memory: bcast_store_floats); conditional models with one tile per workgroup
bool bcast_solve_supported(const NetDesc& nd, bool train, int B, int device);
// floats of the tile store such a launch needs (0: none)
size_t bcast_store_floats(int B, int device);
// sv as for mfma_solve_persistent; CNF_ERR_UNSUPPORTED: not this network / batch.  store: bcast_store_floats(B) floats or null;
// cond / cbs: the per-sample first-layer bias of a conditional model ([B][cbs]) or null
// rec (gradient path; TrainMode, one tile per workgroup -- CNF_ERR_UNSUPPORTED otherwise): every attempt files u_n and its
// stage states U_2..U_6 ([B][n_in + 3], rows of z) in the trajectory slot of step `naccept` and its signed step size in hs_out
struct BcastRecord {
    float* dump;           // the U_2 array of slot 0; u_n sits `n` floats in front of it, U_3.. behind
    size_t n, slot;        // floats between two arrays of a slot, between two slots
    int cap;               // slots
    float* hs_out;
};
cnf_status bcast_solve_launch(const NetDesc& nd, bool train, const float* d_params, const float* d_img, StepState* st_out, float* U0,
                              const float* eps, int B, hipStream_t s, void* mirror, unsigned seq, const Solve3Args& sv, int device,
                              float* store = nullptr, const float* cond = nullptr, int cbs = 0, const BcastRecord* rec = nullptr);
