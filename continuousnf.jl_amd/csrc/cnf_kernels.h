// Kernel argument blocks and launch wrappers shared between the kernel translation units
// (cnf_generic.hip, cnf_mfma.hip) and the C ABI (cnf_abi.hip).
#pragma once
#include "cnf_dev.h"

// One RHS evaluation, optionally preceded by a Tsit5 stage combination.
struct RhsArgs {
    const StepState* st;    // null: plain RHS on (u -> du); else buffers are picked by st->cur
    int B;                  // samples
    size_t S;               // workspace sample stride (>= B)
    int train;              // Mode: 1 TrainMode, 0 TestMode
    float* ws;              // generic-path workspace
    const float* eps;       // n_in x B
    const float* u;         // st == null: input state
    float* du;              // output (unless du_is_k7)
    float* U[2];            // st != null: ping-pong state buffers
    float* K1[2];           // st != null: ping-pong k1/k7 (FSAL)
    float* Ks[5];           // k2..k6
    int nk;                 // number of k's in the stage combination (0..6)
    float coef[6];          // a_{s,1..nk}
    float* ustage;          // optional: write the stage state here
    int ustage_is_unew;     // write the stage state to U[1-cur] (stage 7: it is u_new)
    int du_is_k7;           // write du to K1[1-cur]
    const float* cond;      // conditional models: per-sample first-layer bias [B][cbs], else null
    int cbs;
};

struct NormArgs {
    const StepState* st;
    int kind;               // 0: init A, 1: init B, 2: step error
    size_t n;               // D*B
    float* U[2];
    float* K1[2];
    float* Ks[5];
    float* partials;        // 2 floats per block
    // fused controller (optional): the last block to finish reduces the partials and runs this phase
    unsigned* ticket;       // device counter, zero between launches; null: partials only
    StepState* st_mut;
    int ctrl_phase;
    float n_total;
    void* mirror;           // pinned host mirror of the state (streamed solve) or null; written after the controller
    unsigned seq;           //   with this tag
};

void launch_rhs_generic(const NetDesc& nd, const float* P, const RhsArgs& a, hipStream_t s);
void launch_norm_partials(const NormArgs& a, int nblocks, hipStream_t s);
void launch_controller(StepState* st, const float* partials, int phase, float n_total,
                       hipStream_t s);
void launch_reduce_partials(const StepState* st, const float* partials, float* out3, float n_local,
                            hipStream_t s);
void launch_controller_sums(StepState* st, const float* sums3, int phase, hipStream_t s);
void launch_build_u0(const float* xs, float* u0, int nvars, int D, int B, hipStream_t s, StepState* st_dst = nullptr,
                     const StepState* st_val = nullptr);
void launch_copy_final(const StepState* st, const float* U0, const float* U1, float* out,
                       size_t n, hipStream_t s);
void launch_post(const NetDesc& nd, int train, const float* fsol, float* logpx, float* regs,
                 int B, hipStream_t s);
// same, reading the final state straight from the integrator's buffer U[st->cur]
// (need_done: do nothing unless st->done; sums5: also the five loss sums, through `part` (4 floats per 64 samples) and `ticket`)
void launch_post_state(const NetDesc& nd, int train, const StepState* st, const float* U0, const float* U1,
                       float* logpx, float* regs, int B, hipStream_t s, bool need_done = false, float* sums5 = nullptr,
                       float* part = nullptr, unsigned* ticket = nullptr);
void launch_set_state(StepState* dst, const StepState& v, hipStream_t s);
void launch_cond_bias(const NetDesc& nd, const float* P, const float* ys, float* cond, int cbs, int B,
                      hipStream_t s);
void launch_loss_sums(const float* logpx, const float* regs, int B, float* sums5,
                      hipStream_t s);
