// Argument blocks and launch wrappers of the gradient path (cnf_grad.hip), used by the C ABI.
#pragma once
#include "cnf_dev.h"

#define GRAD_MAX_KSPLIT 128

// Row layouts of the [sample][feature] arrays exchanged between k_adj and k_wgrad.
struct GradLayout {
    int in0;                        // n_in + n_cond: input rows of the first layer
    int in_off[CNF_MAX_LAYERS];     // offset of layer l's INPUT (h_{l-1}, t_{l-1}) in an HS/TS row
    int out_off[CNF_MAX_LAYERS];    // offset of layer l's OUTPUT side (abar_l, pbar_l) in an AB/PB row
    int sum_in, sum_out;            // row lengths
    int out_last;                   // dims[L]
    int max_dim;
};

struct AdjArgs {
    const float* P;                 // flat parameters (Lux layout)
    const float* PT;                // per-layer transposed weights, same offsets
    const float* ustage;            // [B][n_in + 3] stage state (rows of z are read)
    const float* eps;               // [B][n_in]
    const float* ys;                // [B][n_cond] or null
    const float* lam;               // [B][n_in]  cotangent of the z rows of u_{n+1}
    const float* w[5];              // zbar of the later stages
    float wc[5];                    // a_{m,i}
    int nw;
    float cb;                       // b_i
    float hstep;                    // signed step size
    float c_l, c_E, c_n;            // h * b_i * (cotangents of the three scalar rows)
    float* w_out;                   // [B][n_in]
    float* HS; float* TS;           // [B][sum_in]
    float* AB; float* PB;           // [B][sum_out]
    int B;
};

// the six stage pullbacks of one Runge-Kutta step in one launch (MFMA pullback): stages first .. last, then
// lambda <- lambda + sum of their zbar
struct AdjStepArgs {
    AdjArgs st[6];
    int first, last;
    int B;
    int lam_update;
    float* lam_out;
    float kc[6][5];                 // kc[m][d] = a_{m, m-1-d} (0 past stage 0): what zbar_m adds to the sum of the d-th stage after it
};

struct StageK {
    const float* k[6];
    float coef[6];
    int nk;
};

GradLayout grad_layout(const NetDesc& nd);
bool grad_supported(const NetDesc& nd, const GradLayout& g);
void grad_ksplit(const NetDesc& nd, const GradLayout& g, int B, int* ksplit, int* chunk);
hipError_t launch_adj(const NetDesc& nd, const GradLayout& g, const AdjArgs& a, hipStream_t s);
hipError_t launch_wgrad(const NetDesc& nd, const GradLayout& g, const float* AB, const float* PB, const float* HS,
                        const float* TS, float* gpart, int n_params, int B, int ksplit, int chunk, hipStream_t s);
hipError_t launch_grad_reduce(const float* gpart, float* grad, int n_params, int ksplit, hipStream_t s);
// the same sum guarded by the final state of the launch that wrote the partials (zeros if it gave up) + the loss, in device memory
hipError_t launch_grad_finish(const float* gpart, float* grad, int n_params, int ksplit, const StepState* state, const float* sums5,
                              float l1, float l2, float l3, int train, float* loss_dev, hipStream_t s);
hipError_t launch_transpose_params(const NetDesc& nd, const float* P, float* PT, hipStream_t s);
hipError_t launch_stage_combine(const float* u, const StageK& ks, float h, float* out, size_t n, hipStream_t s);
hipError_t launch_lambda_update(float* lam, const StageK& ws, size_t n, hipStream_t s);
hipError_t launch_final_cotangent(const NetDesc& nd, float lambda3, const float* fsol, float* lam, int B,
                                  hipStream_t s);

// ---- MFMA pullback kernel (k_adj_mfma): padded weight images + LDS carve-up -----------------------
struct AdjMfmaLayout {
    int L;
    int dp[CNF_MAX_LAYERS + 1];     // dims padded to 16; dp[0] = pad16(n_in + n_cond)
    int f_off[CNF_MAX_LAYERS];      // forward image  F_l [dp[l+1]][dp[l]]   (row-major, k contiguous)
    int r_off[CNF_MAX_LAYERS];      // reverse image  R_l [dp[l]][dp[l+1]]
    // the same two images in FRAGMENT order for the am_* stream (cnf_am.h): 1 KB per (16-row tile, 16-column k-block), lane L's
    // four values at 16 L -- one fully coalesced b128 per lane instead of 16 rows x 64 bytes per wave load
    int ff_off[CNF_MAX_LAYERS], fr_off[CNF_MAX_LAYERS];
    int b_off[CNF_MAX_LAYERS];      // padded bias
    int o_off[CNF_MAX_LAYERS];      // offset of layer l's output side in the D1/D2/TB rows
    int sum_o;                      // padded row length of D1/D2/TB
    int maxd;                       // max dp
    int nin_p;                      // pad16(n_in)
    int img_floats;
    // per-sample LDS offsets (floats) and stride
    int D1, D2, TB, S0, S1, E, AH, PS;
    int vec4, vec4o;                // HS/TS resp. AB/PB rows are 16-byte aligned: vector stores in the epilogues
    int SR;                         // floats per sample and stage of the scratch rows of the two-launch form: sigma', q, tbar, zdot
};
AdjMfmaLayout adj_mfma_layout(const NetDesc& nd, const GradLayout& g);
bool adj_mfma_supported(const NetDesc& nd, const AdjMfmaLayout& m);
hipError_t launch_pack_adj_images(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* P,
                                  float* img, hipStream_t s);
// one launch per step (the six stage pullbacks and the lambda update)
hipError_t launch_adj_mfma_step(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                                const AdjStepArgs& S, hipStream_t s);
// ... or a RUN of whole steps (first = 5, last = 0, lam_update) as two launches where the batch leaves CUs idle: sweeps 1-3 of
// all stages of all steps side by side, then the hbar chains in turn (k_adj_mfma_run<PHASE 1 / 2>).  d_steps: the steps' arguments
// in DEVICE memory, last step first; scratch: adj_mfma_scratch_floats(m, B, nsteps) floats.
bool adj_mfma_run_split(const NetDesc& nd, const AdjMfmaLayout& m, int B, int nsteps);
// (h_steps: the same arguments in host memory -- a run of one step passes them by value)
hipError_t launch_adj_mfma_run(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                               const AdjStepArgs* d_steps, const AdjStepArgs* h_steps, int nsteps, int B, float* scratch, hipStream_t s);
size_t adj_mfma_scratch_floats(const AdjMfmaLayout& m, size_t B, int nsteps);
// the two-launch forms of the pullback kernels (k_adj_mfma, k_adj3b): -1 where it pays (default; CNF_ADJ_SPLIT=0|1 overrides at
// start-up), 0 never, 1 wherever the parked state fits -- process-wide, for A/B runs and the parity tests (cnf_debug_adj_split)
int adj_split_mode();
void set_adj_split_mode(int mode);
