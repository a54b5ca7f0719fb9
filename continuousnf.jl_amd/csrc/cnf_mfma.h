// MFMA (v_mfma_f32_16x16x4_f32) path of libcnfhip: fused RHS and fused Tsit5 step kernels.
#pragma once
#include "../../include/cnfhip.h"
#include "cnf_dev.h"

#define MF_NB 32          // samples per workgroup tile (two 16-sample MFMA column tiles)
#define MF_THREADS 512    // 8 waves: 2 per SIMD
#define MF_LDS_BYTES 163840

// LDS plan, computed on the host and passed by value.
struct MfmaLayout {
    int L;
    int P[CNF_MAX_LAYERS + 1];      // layer sizes padded to a multiple of 16
    int dims[CNF_MAX_LAYERS + 1];   // true sizes
    int acts[CNF_MAX_LAYERS];
    int SW[CNF_MAX_LAYERS];         // row stride of layer l's weight image (floats)
    int w_off[CNF_MAX_LAYERS];      // offsets inside the weight+bias image
    int b_off[CNF_MAX_LAYERS];
    int img_floats;                 // size of the whole HBM image (multiple of 4)
    int core_img;                   // weights + biases (+ transposes): the part the kernel layouts know;
                                    // with wlds this many floats are copied to LDS
    int wlds;                       // 1: the image is copied to LDS; 0: weights stay in HBM/L2 (big nets)
    int SWT[CNF_MAX_LAYERS];        // wlds == 0: row stride of the transposed weight image of layer l
    int wt_off[CNF_MAX_LAYERS];     // wlds == 0: its offset in the image
    int jvp;                        // forward-mode (J eps) sweep: a tangent image per layer, no reverse sweep
    int tx_off[CNF_MAX_LAYERS + 1]; // LDS offsets of the tangent images tau_1..tau_{L-1} (tau_0 is the eps image)
    int c_off, SWC;                 // 2-layer nets: image of C = W_1 .* W_2^T (P1 x n_in) for the exact trace, -1 if none
    int SX[CNF_MAX_LAYERS + 1];     // row stride of activation region l ([sample][feature])
    int x_off[CNF_MAX_LAYERS + 1];  // LDS offsets (floats) of the activation regions
    int eps_off, du_off, red_off;   // EPS [NB][SX0], DU [NB][SX0], RED [3][P0/16][NB]
    int sc_off;                     // scalar-row Runge-Kutta state [NB][8][3] (u, k1..k7 of dlogp, E, n)
    int total_floats;
    int n_in, norm_z, norm_j;
    int ept;                        // state elements per thread: ceil(NB*(n_in+3)/MF_THREADS)
};

struct MfmaPlan {
    int variant = 0;                // 0: shape not supported by the MFMA path
    MfmaLayout ly{};
    float* d_img = nullptr;         // weight+bias image in HBM (LDS order), refreshed by pack
    float* d_img3 = nullptr;        // headline shape: register-fragment image of k_step3 / k_step3j, refreshed by pack
    bool shape3 = false;            // the network pads to 32-128-128-32, all tanh (either compute mode)
    void* d_img3b = nullptr;        // headline shape: split-bf16 fragment image of k_step3jb
    StepState* d_idle = nullptr;    // a zeroed integrator state for plain evaluations on the step kernels (they read one)
    const float* cond = nullptr;    // conditional models: per-sample first-layer bias [B][cbs] (owned by the handle)
    int cbs = 0;
};

void mfma_plan_init(MfmaPlan& p, const NetDesc& nd);
void mfma_plan_free(MfmaPlan& p);
cnf_status mfma_plan_pack(MfmaPlan& p, const NetDesc& nd, const float* d_params, hipStream_t s);
bool mfma_supported(const MfmaPlan& p, const NetDesc& nd, bool train, int B);
cnf_status mfma_rhs(const MfmaPlan& p, const NetDesc& nd, bool train, const float* u,
                    const float* eps, float* du, int B, hipStream_t s);
// f(u + h*k1) -> Ks[0]  (initial-dt probe; h and the buffer set come from *st)
cnf_status mfma_rhs_init0(const MfmaPlan& p, const NetDesc& nd, bool train, StepState* st, const float* u,
                          const float* eps, float* du, float* partials, unsigned* ticket, int B, hipStream_t s);
cnf_status mfma_rhs_stage(const MfmaPlan& p, const NetDesc& nd, bool train, const StepState* st,
                          float* const U[2], float* const K1[2], float* const Ks[5],
                          const float* eps, int nk, int B, hipStream_t s, StepState* st_init = nullptr,
                          float* partials = nullptr, unsigned* ticket = nullptr);
// one full Tsit5 step attempt: 6 RHS evaluations + error partials in ONE launch.
//  apply_ctrl: the launch first applies the step controller to (st_in, partials_in) -- the
//              outcome of the previous attempt -- redundantly in every workgroup, block 0
//              stores the new state to st_out; otherwise the launch runs from st_in as is.
//  finalize  : follow the launch by the one-block controller kernel (in place on the state
//              slot the launch ran from), so that the host can poll an up-to-date state.
cnf_status mfma_step(const MfmaPlan& p, const NetDesc& nd, bool train, const StepState* st_in,
                     StepState* st_out, float* const U[2], float* const K1[2], float* const Ks[5],
                     const float* eps, const float* partials_in, float* partials_out, bool apply_ctrl,
                     bool finalize, int B, hipStream_t s, float* dump = nullptr,
                     size_t dump_stride = 0, void* mirror = nullptr,
                     unsigned seq = 0, size_t dump_step_stride = 0, int dump_cap = 0, float* hs_out = nullptr);
// arguments of the one-launch solve (k_solve3b, cnf_step3.hip) beyond those of a step launch
struct Solve3Args {
    float* part;          // error partials: two buffers (meeting index parity) of 2 x 512 words {meeting index, float}
    int spin_limit;       // polls a wait makes before it gives up (abort word; the caller then streams step launches)
    unsigned wait_ticks;  // ... and how long it waits at most, in ticks of the 100 MHz real-time clock (whichever runs out first)
    float* trace;         // null, or 4 floats per step attempt: (t, signed h, EEst, accepted) -- cnf_set_step_trace
    int trace_cap;        //   attempts the buffer holds
    unsigned* base_dev;   // device word: meetings held by earlier launches on the buffer `part` (the indices go on from there;
                          //   the launch advances it when it ends)
    int* abort_flag;      // set when a wait ran out
    unsigned long long* t_out;   // null, or {entry stamp, sum of durations, launches}: workgroup 0's 100 MHz real-time clock
    int maxiters;
    int hairer;           // automatic initial dt: the norms of f(u0), a second evaluation and its norm first
    StepState init;       // the integrator's initial state (by value: no launch, no copy in front of this one)
    // inference in the same launch (all null / 0: a bare solve from the columns in a.U[0]):
    const float* xs;      // the data columns [B][nvars]: u0 = (xs, 0) is assembled here       (src/base_icnf.jl:266-286)
    float* logpx;         // post-processing of the final state                               (src/base_icnf.jl:173-187)
    float* regs;
    float* sums5;         // and the five loss sums                                           (src/icnf.jl:489)
    int nvars, naugs, norm_z_aug;
    // a bare solve called with caller-owned columns: u0 (read where it is when every workgroup owns ONE tile, copied into
    // U[0] by the launcher otherwise) and where the final columns go (null on return from the launcher = not written by
    // the kernel: they are in the integrator's buffers, as st_out says)
    const float* u0;
    float* u_out;
};
// the whole solve in one launch (headline shape, B <= 32 x the workgroups the device can hold at once); CNF_ERR_UNSUPPORTED
// otherwise.  `device`: the handle's device (CU count, occupancy and function attributes are kept per device).  sv (cnf_step3.h): the initial state by value, the meeting buffer and its index base, optionally the data
// columns to assemble u0 from and the outputs of the post-processing; st_out: the final state (cur = 0: the final columns
// are in U[0])
cnf_status mfma_solve_persistent(const MfmaPlan& p, const NetDesc& nd, bool train, StepState* st_out, float* const U[2],
                                 const float* eps, int B, hipStream_t s, void* mirror, unsigned seq, Solve3Args& sv, int device,
                                 float* dump = nullptr, size_t dump_stride = 0, size_t dump_step_stride = 0, int dump_cap = 0,
                                 float* hs_out = nullptr,      // dump ...: the trajectory store of the gradient path (as mfma_step)
                                 float* const* K1 = nullptr);  // the k1 / k7 buffer sets: batches of several tiles per workgroup
// workgroups (= error partials) of a step launch; `recording`: the solve files its stage states (gradient path)
int mfma_grid_for(const MfmaPlan& p, int B, bool recording = false, bool train = true);
