/* cnfhip.h -- C ABI of libcnfhip.so: the MI355X (gfx950) backend for the batched
 * augmented-ODE right-hand side of ContinuousNormalizingFlows.jl.
 *
 * The reference is 100 % Julia and has no FFI of its own, so every entry point below
 * names the reference function (file:line under /root/reference) whose work it takes
 * over.  The Julia-side binding a maintainer would add is shown in INTEGRATION.md and
 * shipped (untested: no Julia here) as julia/ContinuousNormalizingFlowsHIPExt.jl.
 *
 * Conventions
 *  - every function returns a cnf_status (0 = OK); nothing throws or aborts across the ABI;
 *  - the caller owns every buffer; a handle owns only the weights and its scratch space;
 *  - pointers are DEVICE pointers unless the function name ends in _host;
 *  - matrices use the reference's layout: Julia column-major `D x B`, i.e. the D floats
 *    of one sample (column) are contiguous and column b starts at float b*D;
 *  - state rows (src/base_icnf.jl:275-282, src/icnf.jl:349):
 *      0..nvars-1 data | nvars..n_in-1 augmented dims | n_in: dlogp | n_in+1: E | n_in+2: n
 *    (the last two rows exist in TrainMode only; n_in = nvars + naugs);
 *  - eps (the Hutchinson probe, `n_in x B`) is always an INPUT: the reference draws it
 *    once per inference call outside the RHS (src/base_icnf.jl:277-278);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *    asynchronous on it unless stated otherwise; a handle may be used from one stream
 *    at a time; there is no global state.
 */
#ifndef CNFHIP_H
#define CNFHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cnf_ctx* cnf_handle;

typedef enum {
    CNF_OK = 0,
    CNF_ERR_BAD_ARG = 1,      /* null pointer, negative size, unknown enum value     */
    CNF_ERR_BAD_SHAPE = 2,    /* dims / parameter count / batch inconsistent          */
    CNF_ERR_HIP = 3,          /* a HIP runtime call failed (see cnf_last_error)       */
    CNF_ERR_NO_DEVICE = 4,    /* no gfx950 device visible                             */
    CNF_ERR_MAXITERS = 5,     /* solver hit maxiters before reaching t1               */
    CNF_ERR_UNSUPPORTED = 6,  /* valid request this build has no kernel for           */
    CNF_ERR_NO_PARAMS = 7,    /* cnf_set_params has not been called                   */
    CNF_ERR_NONFINITE = 8,    /* solver state became NaN/Inf                          */
    CNF_ERR_RCCL = 9          /* librccl missing or an RCCL call failed (see cnf_comm_last_error) */
} cnf_status;

/* Mode (src/types.jl:1-3). */
enum { CNF_MODE_TEST = 0, CNF_MODE_TRAIN = 1 };

/* Dense activations (Lux `Dense(in => out, act)`). */
enum {
    CNF_ACT_IDENTITY = 0, CNF_ACT_TANH = 1, CNF_ACT_SIGMOID = 2, CNF_ACT_SOFTPLUS = 3,
    CNF_ACT_RELU = 4, CNF_ACT_SWISH = 5, CNF_ACT_ELU = 6
};

/* Compute-mode flag: which AD product drives the trace estimate and the n-row.
 * VJP <-> DIVecJacMatrixMode (src/icnf.jl:318-382), JVP <-> DIJacVecMatrixMode
 * (src/icnf.jl:384-456); type lattice at src/types.jl:17-23. */
enum { CNF_AD_VJP = 0, CNF_AD_JVP = 1 };

/* Kernel selection (for tests and benchmarks; AUTO is what a caller wants). */
enum { CNF_KERNEL_AUTO = 0, CNF_KERNEL_GENERIC = 1, CNF_KERNEL_MFMA = 2 };

typedef struct {
    int32_t n_layers;          /* number of Dense layers (1..8)                        */
    const int32_t* dims;       /* n_layers+1 entries: n_in, h1, ..., n_out (= n_in)    */
    const int32_t* acts;       /* n_layers entries: CNF_ACT_*                          */
    int32_t nvars;             /* src/icnf.jl:89                                       */
    int32_t naugs;             /* naugmented, src/icnf.jl:90                           */
    int32_t ad;                /* CNF_AD_VJP | CNF_AD_JVP                              */
    float lambda1, lambda2, lambda3;  /* src/icnf.jl:100-102; NORM_Z / NORM_J /
                                  NORM_Z_AUG are derived as `!iszero(lambda)` exactly as
                                  construct does (src/base_icnf.jl:42-51)              */
    int32_t device;            /* HIP device ordinal                                   */
    int32_t n_cond;            /* conditional models (CondRNODE/CondFFJORD, src/layers/cond_layer.jl):
                                  rows of `ys`; the first Dense layer then takes n_in + n_cond
                                  inputs, dims[0] = n_in + n_cond.  0 for unconditional models */
} cnf_config;

typedef struct {
    float t0, t1;              /* tspan (src/icnf.jl:83); t1 < t0 integrates backwards */
    float abstol, reltol;      /* sol_kwargs (README.md:64-65)                         */
    float dt;                  /* adaptive: initial dt, 0 = automatic; fixed: the step */
    int32_t adaptive;          /* 1 = PI-controlled Tsit5, 0 = fixed steps of `dt`     */
    int32_t maxiters;          /* bound on step attempts (accepted + rejected)         */
    int32_t kernel;            /* CNF_KERNEL_*                                         */
} cnf_solve_opts;

typedef struct {
    int32_t nf;                /* RHS evaluations performed (sol.stats.nf)             */
    int32_t naccept, nreject;
    float t_final, dt_last;
    int32_t kernel_used;       /* CNF_KERNEL_GENERIC | CNF_KERNEL_MFMA                 */
    int32_t launches;          /* kernel launches enqueued for this solve              */
} cnf_solve_stats;

/* ---- lifecycle ------------------------------------------------------------------- */

/* Takes the role of `construct` (src/base_icnf.jl:1-77) for the device side: records
 * the network shape and the static switches. Synchronous. */
cnf_status cnf_create(cnf_handle* out, const cnf_config* cfg);
cnf_status cnf_destroy(cnf_handle h);

/* Upload the flat parameter vector `ps` = ComponentArray(Lux.setup(rng, nn)[1]): per
 * layer `weight` (out x in, column-major) then `bias` (out), layers in order -- the `p`
 * argument of augmented_f (src/icnf.jl:320, used at :329).  Synchronous. */
cnf_status cnf_set_params_host(cnf_handle h, const float* flat, size_t n);
cnf_status cnf_set_params(cnf_handle h, const float* flat_dev, size_t n, void* stream);

/* Conditional models: nn = CondLayer(icnf.nn, ys), nn(z) = icnf.nn(vcat(z, ys))
 * (src/layers/cond_layer.jl:7-9, built per call in inference_prob src/base_icnf.jl:288-309).
 * ys: n_cond x B.  The call folds the conditioning columns of the first layer into a
 * per-sample bias W1[:, n_in:] * ys + b1 held by the handle; every later RHS / solve call with
 * the same B uses it.  Must be called again when ys, B or the parameters change. */
cnf_status cnf_set_cond(cnf_handle h, const float* ys, int B, void* stream);
cnf_status cnf_set_cond_host(cnf_handle h, const float* ys, int B);

/* Lock-step sharded solves.  The reference solves the whole D x B_total batch as
 * ONE ODE system (inference_prob src/base_icnf.jl:266-286), so the adaptive controller sees one
 * error norm over every column.  When the columns are split over GPUs, register a callback that
 * sums `n` floats IN PLACE over all shards (MPI_Allreduce / RCCL / torch.distributed) and returns
 * 0: every adaptive solve on this handle then takes its accept/reject decisions and step sizes
 * from the global sums -- all shards take bit-identical decisions, namely those of the unsharded
 * solve up to the association order of the float sum (1 ulp of the error norm).  The
 * callback runs on the calling thread, three floats at a time, once per attempted step (plus twice
 * for the automatic initial dt); every shard must call the solve collectively.  fn = NULL
 * (default): independent per-shard solves.  Fixed-dt solves never call it. */
typedef int (*cnf_shard_reduce_fn)(float* sums, int n, void* user);
cnf_status cnf_set_shard_reduce(cnf_handle h, cnf_shard_reduce_fn fn, void* user);

/* ---- the hot path ---------------------------------------------------------------- */

/* One evaluation of augmented_f over the whole batch:
 *   TrainMode/VJP  src/icnf.jl:318-350 (out-of-place), :352-382 (in-place)
 *   TrainMode/JVP  src/icnf.jl:384-420, :422-456
 *   TestMode       src/icnf.jl:148-164, :166-184 with jacobian_batched src/utils.jl:1-36
 * u: D x B, eps: n_in x B (ignored in TestMode, may be NULL), du: D x B; u and du must
 * not alias.  The ODE is autonomous, so `t` is not an argument (icnf.jl:321 ignores it). */
cnf_status cnf_rhs(cnf_handle h, int mode, int kernel, const float* u, const float* eps,
                   float* du, int B, void* stream);
cnf_status cnf_rhs_host(cnf_handle h, int mode, int kernel, const float* u,
                        const float* eps, float* du, int B);

/* base_sol (src/base_icnf.jl:137-143) with Tsit5 fixed as the algorithm: integrates
 * u' = augmented_f(u) from opts->t0 to opts->t1 on the device and writes the final
 * `D x B` state -- what `get_fsol(sol)` returns (src/base_icnf.jl:213-215).  Blocks until
 * the solve has finished (the step count of an adaptive solve is data dependent). */
cnf_status cnf_solve_tsit5(cnf_handle h, int mode, const float* u0, const float* eps,
                           float* u_out, int B, const cnf_solve_opts* opts,
                           cnf_solve_stats* stats, void* stream);

/* Same with HOST matrices (copied to the device and back explicitly): the form a CPU-array
 * caller such as the Julia shim binds when it has no device arrays of its own. */
cnf_status cnf_solve_tsit5_host(cnf_handle h, int mode, const float* u0, const float* eps,
                                float* u_out, int B, const cnf_solve_opts* opts,
                                cnf_solve_stats* stats);

/* inference_prob's state assembly (src/base_icnf.jl:275-276, 282):
 * u0 = vcat(xs, zeros(naugs + n_aug + 1, B)); xs: nvars x B. */
cnf_status cnf_build_u0(cnf_handle h, int mode, const float* xs, float* u0, int B,
                        void* stream);

/* inference_sol (src/base_icnf.jl:167-189): from the final state computes
 * logpx[b] = logpdf(MvNormal(0, I), z_b) - dlogp_b and the rows (E, n, A) into
 * regs (3 x B, row-major: regs[0*B+b]=E, [1*B+b]=n, [2*B+b]=A; E and n are 0 in
 * TestMode, where the reference has no such rows). */
cnf_status cnf_inference_post(cnf_handle h, int mode, const float* u_final, float* logpx,
                              float* regs, int B, void* stream);

/* inference (src/base_icnf.jl:407-415) = build_u0 + solve + post, all on device, no allocations.
 * logpx / regs (and u_final) are written stream-ordered: valid for later work on `stream`, or after
 * a stream synchronisation for the host. */
cnf_status cnf_inference(cnf_handle h, int mode, const float* xs, const float* eps,
                         float* logpx, float* regs, float* u_final /* may be NULL */,
                         int B, const cnf_solve_opts* opts, cnf_solve_stats* stats,
                         void* stream);
/* cnf_inference followed by cnf_loss_sums in one call (no host round trip in between): what one rank of a
 * sharded `loss` evaluation needs before its 5-float all-reduce. */
cnf_status cnf_inference_sums(cnf_handle h, int mode, const float* xs, const float* eps, float* logpx,
                              float* regs, float* sums5, int B, const cnf_solve_opts* opts,
                              cnf_solve_stats* stats, void* stream);
cnf_status cnf_inference_host(cnf_handle h, int mode, const float* xs, const float* eps,
                              float* logpx, float* regs, float* u_final, int B,
                              const cnf_solve_opts* opts, cnf_solve_stats* stats);

/* Submitted inferences -- what a loop over data batches (`loss` over the mini-batches of an epoch,
 * src/exts/mlj_ext/core_icnf.jl:59-73; `logpdf` over many column blocks, src/exts/dist_ext/core_icnf.jl:23-31) needs to
 * keep the GPU busy: cnf_inference_submit enqueues cnf_inference_sums (sums5 may be NULL) and returns without waiting for
 * it when the solve is ONE launch (the headline path); cnf_inference_collect completes the OLDEST submitted inference and
 * returns ITS status and statistics (errors of a submitted inference are reported there).  Up to three may be outstanding
 * per handle; they must use the same stream and distinct output buffers.  An inference that does not take the one-launch
 * path is simply completed inside the submit call; a launch that could not place its workgroups is run again on the
 * streamed driver inside collect.  Any other call on the handle collects outstanding submissions first (their
 * statistics are dropped).  cnf_inference_pending: how many are outstanding. */
cnf_status cnf_inference_submit(cnf_handle h, int mode, const float* xs, const float* eps, float* logpx,
                                float* regs, float* sums5 /* may be NULL */, int B, const cnf_solve_opts* opts,
                                void* stream);
cnf_status cnf_inference_collect(cnf_handle h, cnf_solve_stats* stats /* may be NULL */);
int cnf_inference_pending(cnf_handle h);

/* The local part of `loss` (src/icnf.jl:481-490; src/base_icnf.jl:489-497): writes
 * sums[5] = (sum logpx, sum E, sum n, sum A, B) to DEVICE memory.  These five floats are
 * the only cross-shard quantity: the caller all-reduces them (RCCL, ncclSum) and then
 * cnf_loss_from_sums gives mean(-logpx + l1 E + l2 n + l3 A) (Train) or -mean(logpx). */
cnf_status cnf_loss_sums(cnf_handle h, const float* logpx, const float* regs, int B,
                         float* sums5, void* stream);
cnf_status cnf_loss_from_sums(cnf_handle h, int mode, const float* sums5_host, float* loss);

/* ---- multi-GPU: the one collective of the path (b') ------------------------
 *
 * The reference has no collective: it is single-process and `mean` at src/icnf.jl:489 /
 * src/base_icnf.jl:496 runs over all columns.  With the columns sharded over GPUs (one process per
 * GPU) that mean is an RCCL all-reduce (ncclSum, ncclFloat32, over xGMI) of the five floats of
 * cnf_loss_sums.  These entry points let a caller without torch.distributed (the Julia shim) do
 * it: rank 0 draws an id (cnf_comm_unique_id), hands the 128 bytes to the other ranks by any means
 * it has (a file, a socket, MPI), every rank calls cnf_comm_init, then cnf_loss_allreduce after
 * cnf_loss_sums / cnf_inference_sums.  A cnf_comm IS an ncclComm_t: a communicator the caller
 * already owns (NCCL.jl, torch) can be passed to cnf_loss_allreduce as it is.  librccl is loaded on
 * first use (dlopen; CNFHIP_RCCL_LIB overrides the search), so single-GPU use needs no RCCL. */
#define CNF_COMM_ID_BYTES 128
typedef void* cnf_comm;                                  /* ncclComm_t */
cnf_status cnf_comm_unique_id(char* id /* CNF_COMM_ID_BYTES */);          /* ncclGetUniqueId   */
cnf_status cnf_comm_init(cnf_comm* out, int world_size, int rank,
                         const char* id /* CNF_COMM_ID_BYTES */, int device);  /* ncclCommInitRank on `device`
                                                      (< 0: on the device the caller has already selected) */
cnf_status cnf_comm_destroy(cnf_comm comm);
/* "<hostname>|<boot id>/<pci bus id>" of `device` (< 0: the current one), NUL-terminated into out[cap] (128 bytes suffice).
 * The ranks exchange these keys over their own bootstrap BEFORE cnf_comm_init and must not go on when two are equal: RCCL
 * does not accept two ranks on one GPU (ncclInvalidUsage "Duplicate GPU detected" out of ncclCommInitRank, on every rank, after
 * its bootstrap).  No reference counterpart (the reference has no multi-GPU code). */
cnf_status cnf_comm_device_key(int device, char* out, size_t cap);
cnf_status cnf_comm_size(cnf_comm comm, int* world_size);                     /* ncclCommCount */
/* In-place sum over the ranks of n DEVICE floats, stream-ordered (ncclAllReduce). */
cnf_status cnf_comm_allreduce(cnf_comm comm, float* buf_dev, size_t n, void* stream);
const char* cnf_comm_last_error(void);    /* detail for the last CNF_ERR_RCCL on this thread */
const char* cnf_comm_library(void);       /* which librccl was resolved ("" if none)          */

/* `mean` of loss over ALL shards (src/icnf.jl:489): all-reduce sums5 (device, from cnf_loss_sums /
 * cnf_inference_sums) in place on `stream`; afterwards every rank holds the global sums and
 * cnf_loss_from_sums gives the loss of the whole batch.  One 20-byte ncclAllReduce: latency-bound
 * (SURVEY section 5), no data-path traffic. */
cnf_status cnf_loss_allreduce(cnf_handle h, cnf_comm comm, float* sums5, void* stream);
/* Lock-step sharded solves through RCCL instead of a host callback (see cnf_set_shard_reduce): the
 * three controller floats are all-reduced on the solve's stream, no host round trip.  comm = NULL
 * switches it off.  Every rank must call the solve collectively. */
cnf_status cnf_set_shard_comm(cnf_handle h, cnf_comm comm);

/* ---- training ------------------------------------------------ */

/* loss(icnf, TrainMode(), xs, ps, st) (src/icnf.jl:481-490) AND its gradient w.r.t. the flat
 * parameter vector -- what MLJModelInterface.fit (src/exts/mlj_ext/core_icnf.jl:59-73) obtains
 * from Enzyme + SciMLSensitivity through `solve`.  Computed as the discrete adjoint of the Tsit5
 * steps the forward solve took (step sizes are constants of the differentiation): the exact
 * gradient of the returned loss value.  xs: nvars x B, eps: n_in x B (drawn by the caller, as in
 * inference_prob src/base_icnf.jl:277-278); *loss_out is a HOST float; grad: n_params floats in the
 * layout of cnf_set_params (device pointer; host pointer in the _host variant).  Both are means
 * over the B columns given; a caller that shards the batch combines (loss, grad) weighted by B.
 * Conditional models: call cnf_set_cond first, as for any other solve.  TrainMode (the mode the
 * reference trains in: core_icnf.jl:60); the TestMode loss has cnf_loss_grad_test below. */
cnf_status cnf_loss_grad(cnf_handle h, const float* xs, const float* eps, int B,
                         const cnf_solve_opts* opts, float* loss_out, float* grad,
                         cnf_solve_stats* stats, void* stream);
cnf_status cnf_loss_grad_host(cnf_handle h, const float* xs, const float* eps, int B,
                              const cnf_solve_opts* opts, float* loss_out, float* grad,
                              cnf_solve_stats* stats);
/* loss(icnf, TestMode(), xs, ps, st) = -mean(logpx) (src/base_icnf.jl:489-497) and its gradient w.r.t. the flat parameters
 * through the exact-trace solve: the derivative the reference's call tests and benchmark suite take besides the TrainMode
 * one (test/call_tests.jl `diff_loss` with omode = TestMode(); benchmark/benchmarks.jl:60-99, "AD-1-order" / "test").
 * Device pointers, arguments as cnf_loss_grad (no eps: the exact trace draws nothing); cnf_grad_steps / cnf_grad_x apply
 * to it as well.  Where the whole gradient runs in the launch of the solve -- two tanh layers (closed-form trace) or one,
 * n_in <= 16, <= 64 hidden units, n_in + n_cond <= 16, B <= 8192 -- it does (k_solve_wave<TEST, GRAD>); every other Dense chain
 * (deeper, wider, any activation, conditional) takes a recorded exact-trace solve followed by ONE launch of a generic adjoint
 * kernel over all accepted steps (k_adj_test: coverage first, VALU; CNF_ERR_UNSUPPORTED only beyond its LDS budget). */
cnf_status cnf_loss_grad_test(cnf_handle h, const float* xs, int B, const cnf_solve_opts* opts, float* loss_out,
                              float* grad, cnf_solve_stats* stats, void* stream);
cnf_status cnf_loss_grad_test_host(cnf_handle h, const float* xs, int B, const cnf_solve_opts* opts, float* loss_out,
                                   float* grad, cnf_solve_stats* stats);      /* xs and grad in HOST memory */
/* Submitted gradients -- a training loop that never waits for the GPU (the loop of MLJModelInterface.fit,
 * src/exts/mlj_ext/core_icnf.jl:59-73, with the optimiser's update on the device).  cnf_loss_grad_submit enqueues what
 * cnf_loss_grad (mode = CNF_MODE_TRAIN) / cnf_loss_grad_test (CNF_MODE_TEST, eps = NULL) compute and returns: the loss (one float)
 * and the gradient are left in DEVICE memory, stream-ordered, for what the caller enqueues next -- the parameter update, then
 * cnf_set_params_async with the new parameters and the next submission.  cnf_loss_grad_collect completes the oldest submission
 * (the queue is cnf_inference_submit's: at most three in flight, all on one stream) and reports how it ended; a launch that gave
 * up has delivered ZEROS and a NaN loss and is reported as CNF_ERR_UNSUPPORTED (run that batch again with cnf_loss_grad).  Exists
 * where the gradient runs in the launch of the solve (see cnf_loss_grad_test); CNF_ERR_UNSUPPORTED at once otherwise. */
cnf_status cnf_loss_grad_submit(cnf_handle h, int mode, const float* xs, const float* eps, int B, const cnf_solve_opts* opts,
                                float* loss_dev, float* grad, void* stream);
cnf_status cnf_loss_grad_collect(cnf_handle h, cnf_solve_stats* stats);
/* cnf_set_params (device pointer) without its host waits: copy and packing are enqueued on `stream`.  For callers whose every
 * launch on this handle goes to that one stream.  Submitted GRADIENTS stay in flight across it (they are never run again);
 * submitted INFERENCES are settled first (a host wait), because one that gave up is run again with the handle's parameters. */
cnf_status cnf_set_params_async(cnf_handle h, const float* flat_dev, size_t n, void* stream);
/* The signed sizes of the steps the last cnf_loss_grad on this handle accepted (the discrete map
 * it differentiated): writes min(n, cap) floats to hs (may be NULL) and returns n. */
int cnf_grad_steps(cnf_handle h, float* hs, int cap);
/* d loss / d xs of the last cnf_loss_grad call on this handle (the other gradient the reference's own call tests take
 * besides the one w.r.t. ps: test/call_tests.jl, `diff2_loss`): gx is nvars x B, laid out as xs, DEVICE memory, written
 * stream-ordered.  It is the adjoint state at t0 restricted to the data rows (u0 = vcat(xs, zeros), src/base_icnf.jl:
 * 275-276): left behind by the backward sweep, nothing is recomputed.  B must be that call's batch size. */
cnf_status cnf_grad_x(cnf_handle h, float* gx, int B, void* stream);

/* ---- introspection --------------------------------------------------------------- */
const char* cnf_status_string(cnf_status s);
const char* cnf_last_error(cnf_handle h);   /* detail for the last non-OK status       */
int cnf_abi_version(void);
/* Rows of the state: n_in + 1 + (mode == TRAIN ? 2 : 0)  (src/icnf.jl:106-108). */
int cnf_state_rows(cnf_handle h, int mode);
/* Which kernel AUTO resolves to for this handle and batch (CNF_KERNEL_*). */
int cnf_kernel_for(cnf_handle h, int mode, int B);
/* Algorithmic work of ONE RHS evaluation over B samples:
 * bytes = 4*B*(n_in + [train] n_in + D) + 4*P; flops = B*(4M + 6 n_in) (Train);
 * Test: B*3M for 2-layer nets (closed-form trace), else B*(2M + n_in*2M). */
cnf_status cnf_rhs_work(cnf_handle h, int mode, int B, double* flops, double* bytes);
/* Measurement aid (bench.py).  While enabled, the kernel of the one-launch solve reads the 100 MHz real-time clock at
 * its entry and exit (workgroup 0) and adds the interval to a device-side total: no event, no extra synchronisation
 * inside the timed region.  The call returns the mean interval (microseconds) and the number of such launches since the
 * previous call, zeroes both, and sets the switch; pointers may be null.  0 launches: the solves streamed step launches. */
cnf_status cnf_solve_kernel_time(cnf_handle h, int enable, float* mean_us, int* launches);

/* Diagnostic: what `sol.stats` and the integrator's step log would tell a Julia caller if base_sol kept `sol`
 * (src/base_icnf.jl:141-142 returns only the final state).  trace_dev: caller-owned DEVICE buffer of 4 * cap_attempts
 * floats; every later one-launch solve on this handle files (t, signed h, EEst, accepted ? 1 : 0) of step attempt i at
 * floats 4i..4i+3 (attempts beyond the capacity are not filed).  cap_attempts = 0 switches it off. */
cnf_status cnf_set_step_trace(cnf_handle h, float* trace_dev, int cap_attempts);
/* One-launch solves on this handle that ran out of a wait (CUs held by someone else) and were run again, from u0, on the
 * streamed driver; the calls returned CNF_OK with the streamed result. */
int cnf_solve_fallbacks(cnf_handle h);
/* How long a wait inside a one-launch solve of this handle lasts before the launch gives up and the call runs on the
 * streamed driver: wait_us microseconds of the kernel's 100 MHz clock per tile a workgroup carries (default 2000; <= 0:
 * keep), and/or poll_limit polls (default unbounded; 1 makes every wait run out at once -- tests; <= 0: keep).  The
 * reference has no counterpart (its solve is a CPU loop, src/base_icnf.jl:137-143): this bounds the stall a co-tenant of
 * the GPU can cause.  Environment defaults for new handles: CNF_SOLVE_WAIT_US, CNF_SOLVE_POLL_LIMIT. */
cnf_status cnf_set_solve_wait(cnf_handle h, int wait_us, int poll_limit);
/* The gradient's pullback kernels (loss_and_grad, src/exts/mlj_ext/core_icnf.jl:59-73) run a step either as one launch or --
 * where the batch leaves compute units idle -- as two: the three sweeps of every stage that do not depend on the adjoint state
 * side by side, then the remaining chain stage by stage.  mode -1: chosen per call (default; CNF_ADJ_SPLIT=0|1 presets it),
 * 0: always one launch, 1: always two.  Process-wide; both forms compute the same gradient (A/B runs, the parity tests).
 * Returns the mode that was in force. */
int cnf_set_grad_split(int mode);
/* Arithmetic self-test (no handle): C (16 x 16, row-major, HOST) = A Bt^T for HOST matrices A, Bt of 16 x K floats
 * (row-major, K a multiple of 32), computed on one wavefront with the operand split and the six-term bf16 MFMA product
 * the headline kernels use in place of the reference's sgemm (Lux Dense inside src/icnf.jl:331-332).  The parity suite
 * bounds |C - float64| by a multiple of eps32 * sum_k |a b|. */
cnf_status cnf_selftest_split_product(const float* A, const float* Bt, float* C, int K);
/* Test support (no handle): n_workgroups workgroups that each claim a whole CU (all of its LDS) and do nothing for
 * `microseconds` (at most 100000) on `stream` -- a bounded stand-in for another tenant of the GPU, so that the behaviour of a
 * one-launch solve whose workgroups cannot all be placed (cnf_set_solve_wait, cnf_solve_fallbacks) can be tested. */
cnf_status cnf_selftest_hold_cus(int n_workgroups, int microseconds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CNFHIP_H */
