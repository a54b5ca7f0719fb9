// k_solve_wave (cnf_wave.hip): the whole Tsit5 solve of a small two-layer network in one launch, one wave per 16-sample
// tile, registers only (BASELINE configs 1 and 2; the README / regression networks n_in -> 3 n_in -> n_in).
#pragma once
#include "cnf_mfma.h"

// a two-layer network whose 16-row tile counts have an instantiation, B within the meeting buffer's reach
bool wave_solve_supported(const NetDesc& nd, bool train, int B);
// sv as for mfma_solve_persistent (Solve3Args: the initial state by value, the meeting buffer and its index base, the wait
// bounds; either sv.xs + the post-processing outputs, or sv.u0 / sv.u_out of a bare solve -- u_out null: the final columns
// go to U0).  CNF_ERR_UNSUPPORTED: not this network / batch.
cnf_status wave_solve_launch(const NetDesc& nd, bool train, const float* d_params, const float* cond, int cbs, StepState* st_out,
                             float* U0, const float* eps, int B, hipStream_t s, void* mirror, unsigned seq, const Solve3Args& sv);
