// Exact trace (TestMode, src/icnf.jl:148-184 with jacobian_batched src/utils.jl:1-36) on MFMA for
// networks with three or more layers -- launch wrappers of cnf_trace.hip.
#pragma once
#include "cnf_grad.h"

struct TraceArgs {
    const float* u;         // [B][n_in + 1] state (rows of z are read)
    const float* ys;        // [B][n_cond] or null
    float* du;              // [B][n_in + 1]
    const StepState* st;    // solver: skip when done; null for a plain evaluation
    float* K1[2];           // solver, stage 7: du goes to K1[1 - st->cur]
    int du_is_k7;
    int B;
    // Runge-Kutta stage state formed by the kernel itself (nk > 0; `u` is ignored then):
    //   u = U[cur] + h * sum_j coef[j] k_{j+1},   k_1 = K1[cur], k_2.. = Ks[0..],   h and cur from the device state;
    // also_unew: the stage state is the new solution (stage 6) and is stored to U[1 - cur], all D rows
    float* U[2];
    const float* Ks[5];
    int nk;
    float coef[6];
    int also_unew;
    // k_trace3s only (trace_fused_supported): norms and controller in the same launch, as k_norm_partials does them
    int norm_kind;          // -1: none; 0 / 1: the two norms of the automatic initial dt over this evaluation; 2: the error norm
    int fused_step;         // 1: the six stage evaluations of an attempt in this launch (then norm_kind = 2)
    float* partials;        // 2 floats per workgroup (agent-scope atomics)
    unsigned* ticket;       // zero between launches; the workgroup that draws the last one runs the controller phase
    StepState* st_mut;      // the state it acts on (tolerances are read from it)
    float n_total;          // D * B
    void* mirror;           // pinned host mirror, written after the controller
    unsigned seq;
};

struct TraceLayout {
    int nct;                // column tiles of 16 tangent columns per group
    int gs;                 // samples per group = 16 nct / nin_p
    int PD;                 // per-sample stride of the sigma' rows
    int PT;                 // per-column stride of the tangent buffers
    int off_T0, off_T1, off_red;   // float offsets after the 16 sigma' rows
    int total_floats;
};

bool trace_mfma_supported(const NetDesc& nd, const AdjMfmaLayout& m);
TraceLayout trace_layout(const NetDesc& nd, const AdjMfmaLayout& m);
hipError_t launch_trace_mfma(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                             const TraceArgs& a, hipStream_t s);
bool trace_fused_supported(const NetDesc& nd, const AdjMfmaLayout& m, int B);
int trace_fused_grid(int B);          // workgroups (= error partials) of a fused launch
// the whole TestMode solve in one launch (k_trace3s<SOLVE>): needs Solve3Args (cnf_mfma.h)
struct Solve3Args;
bool trace_solve_supported(const NetDesc& nd, const AdjMfmaLayout& m, int B, int device);
hipError_t launch_trace_solve(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                              const TraceArgs& a, const Solve3Args& sv, hipStream_t s);
bool jvp_mfma_supported(const NetDesc& nd, const AdjMfmaLayout& m);
hipError_t launch_jvp_mfma(const NetDesc& nd, const GradLayout& g, const AdjMfmaLayout& m, const float* img,
                           const TraceArgs& a, const float* eps, hipStream_t s);
