// k_adj_test (cnf_gradt.hip): the adjoint of the exact-trace (TestMode) solve for any Dense chain -- the route of
// cnf_loss_grad_test for the networks k_solve_wave<TEST, GRAD> does not take.
#pragma once
#include "cnf_dev.h"

struct AdjTestArgs {
    const float* P;            // flat parameters (Lux layout)
    const float* traj;         // u_n of accepted step n at traj + n * slot_stride: [B][n_in + 1]; slot `nsteps` = the final state
    size_t slot_stride;        // floats between two slots
    const float* hs;           // DEVICE: the signed sizes of the accepted steps
    int nsteps;
    const float* ys;           // [B][n_cond] or null
    float lam_l;               // cotangent of the dlogp row: 1 / B
    float* lam_out;            // [B][n_in]: d loss / d z(t0)   (cnf_grad_x)
    float* gpart;              // [adj_test_workgroups(B)][n_params]: one partial of the flat gradient per workgroup
    float* scratch;            // [adj_test_workgroups(B)][scratch_per_wg]
    size_t scratch_per_wg;     // adj_test_scratch_floats(nd)
    int B, n_params;
};

size_t adj_test_scratch_floats(const NetDesc& nd);
int adj_test_workgroups(int B);
hipError_t launch_adj_test(const NetDesc& nd, const AdjTestArgs& a, hipStream_t s);
