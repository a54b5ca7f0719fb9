// k_adj3b (cnf_adj3b.hip): the pullback of a run of recorded Runge-Kutta steps of the headline shape on split-bf16 products
// (k_adj3's work at the matrix rate of the forward kernels), last step first, in ONE launch: per step the six stage pullbacks
// and lambda <- lambda + sum of their zbar; the factor rows of stage i of the j-th step of the run (j = 0: the last one) are
// filed in slot 6 j + i of the four factor arrays, as launch_wgrad contracts them.
#pragma once
#include "cnf_grad.h"

constexpr int ADJ3B_MAX_STEPS = 32;               // steps per launch (their sizes travel in the kernel arguments)
constexpr int ADJ3B_PARK_FLOATS = 8 * 2048 + 3 * 1024;   // two-launch form: parked state per stage and 32-sample workgroup

struct Adj3bSteps {
    const float* traj;              // the trajectory store: slot s holds the six stage states of step s, [6][B][n_in + 3]
    size_t slot_stride;             // floats between two slots
    size_t n;                       // floats between two stage states of a slot: B (n_in + 3)
    int step_hi, step_lo;           // the run: steps step_hi, step_hi - 1, ..., step_lo
    float hs[ADJ3B_MAX_STEPS];      // hs[j]: signed size of step step_hi - j
    const float* eps;               // [B][n_in]
    const float* lam;               // [B][n_in]  cotangent of the z rows behind step step_hi
    float* lam_out;                 // [B][n_in]  ... in front of step step_lo (may be lam)
    float* HS; float* TS;           // [6 steps][B][sum_in]
    float* AB; float* PB;           // [6 steps][B][sum_out]
    float lam_l, lam_E, lam_n;      // cotangents of the three scalar rows (constant along the solve)
    float bw[6];                    // b_i
    float kc[6][5];                 // kc[m][d] = a_{m, m-1-d} (0 past stage 0): what zbar_m adds to the sum of the d-th stage after it
    int B;
    float* park;                    // two-launch form: adj3b_park_floats(B, steps of the run) floats, or null (one launch)
};

bool adj3b_supported(const NetDesc& nd);          // 32-128-128-32 (padded), tanh, VJP handle, no conditioning; CNF_ADJ3B=0 switches it off
// d_img3b: the split-fragment image of k_step3b (MfmaPlan::d_img3b)
// Whether a run of `steps` steps at batch B is better served by two launches (the stage-parallel sweeps 1-3, then the hbar chains
// in turn) on this device -- then M.park must hold adj3b_park_floats(B, steps); CNF_ADJ_SPLIT=0: never
bool adj3b_split(int B, int steps);
size_t adj3b_park_floats(int B, int steps);
hipError_t launch_adj3b(const NetDesc& nd, const GradLayout& g, const void* d_img3b, const Adj3bSteps& M, hipStream_t s);
