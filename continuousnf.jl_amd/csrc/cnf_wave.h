// k_solve_wave (cnf_wave.hip): the whole Tsit5 solve of a small two-layer network in one launch, one wave per 16-sample
// tile, registers only (BASELINE configs 1 and 2; the README / regression networks n_in -> 3 n_in -> n_in).
#pragma once
#include "cnf_mfma.h"

// The gradient path of small batches in the same launch (k_solve_wave<.., GRAD>): after the forward solve the waves run the
// discrete adjoint of the accepted steps and leave one partial of the flat gradient each (k_grad_reduce adds them).
#define WV_GCAP 1024                   // accepted steps a launch can differentiate (step sizes in LDS)
struct WaveGradArgs {
    float* traj = nullptr;             // [traj_cap][6 stages][waves][64 lanes][n_in tiles] x 4 floats: z rows of U_1 = u_n, U_2..U_6 as the lanes hold them
    int traj_cap = 0;
    float* hs_out = nullptr;           // [traj_cap] signed step sizes (the host's copy: cnf_grad_steps)
    float* gpart = nullptr;            // [waves][n_params]
    float* lam_out = nullptr;          // [B][n_in]  d loss / d z(t0)   (cnf_grad_x)
    float* rich = nullptr;             // TrainMode / VJP, small batches: [traj_cap][6][waves][64][3 (n_in + hidden tiles)] x 4 floats -- the
                                       // forward evaluations' intermediates, so that the backward pass does not form them again
    const float* ys = nullptr;         // conditional models: [B][n_cond] (the columns of W_1 behind z get their gradient from them)
    int n_params = 0;
    float lam1 = 0.f, lam2 = 0.f, lam3 = 0.f;
};
// floats of trajectory store per accepted step, waves of a launch
size_t wave_grad_traj_floats(const NetDesc& nd, int B);
int wave_grad_waves(int B);
size_t wave_grad_rich_floats(const NetDesc& nd, int B, bool train);   // per accepted step; 0: no such form
// two tanh layers (or one + the appended identity), n_in (+ n_cond of a conditional model) <= 16, at most 512 waves
// (B <= 8192); TrainMode:
// both compute modes; TestMode: the adjoint of the exact-trace solve (closed form of two-layer networks)
bool wave_grad_supported(const NetDesc& nd, int B, bool train = true);

// a two-layer network whose 16-row tile counts have an instantiation, B within the meeting buffer's reach
bool wave_solve_supported(const NetDesc& nd, bool train, int B);
// sv as for mfma_solve_persistent (Solve3Args: the initial state by value, the meeting buffer and its index base, the wait
// bounds; either sv.xs + the post-processing outputs, or sv.u0 / sv.u_out of a bare solve -- u_out null: the final columns
// go to U0).  CNF_ERR_UNSUPPORTED: not this network / batch.
cnf_status wave_solve_launch(const NetDesc& nd, bool train, const float* d_params, const float* cond, int cbs, StepState* st_out,
                             float* U0, const float* eps, int B, hipStream_t s, void* mirror, unsigned seq, const Solve3Args& sv,
                             const WaveGradArgs* grad = nullptr);
