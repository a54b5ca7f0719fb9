// C ABI of libcnfhip.so (see include/cnfhip.h): handle management, parameter upload, the
// RHS entry point, the on-device Tsit5 driver, post-processing and the loss sums.
#include "../../include/cnfhip.h"
#include "cnf_dev.h"
#include "cnf_kernels.h"
#include "cnf_mfma.h"
#include "cnf_grad.h"
#include "cnf_trace.h"
#include "cnf_mirror.h"
#include "cnf_wave.h"
#include "cnf_bcast.h"
#include "cnf_gradt.h"
#include "cnf_adj3b.h"
#include "cnf_step3.h"
#include <immintrin.h>
#include <sched.h>
#include <vector>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <mutex>
#include <deque>
#include <string>
#include <vector>

#define CNF_ABI_VERSION 1

struct cnf_ctx {
    NetDesc nd{};
    float lam[3]{};
    int device = 0;
    size_t n_params = 0;
    float* d_params = nullptr;
    bool have_params = false;
    MfmaPlan mfma{};              // packed weights etc. for the MFMA path (cnf_mfma.hip)

    // scratch, sized for cap_B samples
    size_t cap_B = 0;
    float* arena = nullptr;       // one allocation carved into the buffers below
    float* ws = nullptr;
    float* U[2] = {nullptr, nullptr};
    float* K1[2] = {nullptr, nullptr};
    float* Ks[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* d_cond = nullptr;      // conditional models: per-sample first-layer bias [cond_B][cbs]
    int cond_B = 0, cbs = 0;
    float* tmp_logpx = nullptr;
    float* tmp_regs = nullptr;
    float* post_part = nullptr;   // loss-sum partials of the post-processing kernel: 4 floats per 64 columns
    // inferences submitted and not yet collected (cnf_inference_submit / cnf_inference_collect): the one-launch solve is
    // enqueued and the call returns; the outcome is read later from the launch's own slot of the host mirror
    struct Submitted {
        bool launched = false;    // a one-launch solve is in flight (else: the call completed synchronously; status / stats say how)
        unsigned seq = 0;         // its launch index = the tag it publishes its final state with
        int slot = 0;             // ... into this slot of the host mirror
        bool hairer = false;
        bool grad = false;        // a submitted GRADIENT (cnf_loss_grad_submit): nothing is run again if it gave up
        int mode = 0, B = 0, k = 0;
        const float *xs = nullptr, *eps = nullptr;
        float *logpx = nullptr, *regs = nullptr, *sums5 = nullptr;
        cnf_solve_opts opts{};
        hipStream_t st = nullptr;
        cnf_solve_stats stats{};
        cnf_status status = CNF_OK;
    };
    std::deque<Submitted> submitted;
    unsigned one_launch_count = 0; // one-launch solves so far: they take the mirror slots 1, 2, 3 in turn (slot 0: the streamed solves)
    bool submitting = false;      // inside cnf_inference_submit: a one-launch solve is recorded in `submitted`, not waited for
    bool sub_taken = false;       //   ... and this call was
    bool collecting = false;      // inside submit / collect: check_call does not drain the queue
    bool no_persist = false;      // the fallback of a collected launch: straight to the streamed driver
    bool time_kernel = false;     // cnf_solve_kernel_time: the one-launch solve kernel adds up its own durations
    int fallbacks = 0;            // one-launch solves that ran out of a wait and were run again on the streamed driver
    int wait_us = 0, poll_limit = 0;   // cnf_set_solve_wait (0: the process defaults)
    float* step_trace = nullptr;  // cnf_set_step_trace: caller-owned device buffer, 4 floats per step attempt
    int step_trace_cap = 0;
    float* partials = nullptr;    // 2 * MAX_PARTIALS floats
    StepState* last_state = nullptr; // device slot holding the state at the end of the last solve
    StepState* d_state = nullptr;   // two slots: [0] canonical, [1] ping-pong partner of the fused MFMA path
    StepState* h_state = nullptr; // pinned, two slots for pipelined polling + one init slot
    // streamed solve: the step kernel mirrors the state into pinned, host-coherent memory after every
    // controller run as tagged granules (cnf_mirror.h); the host polls them
    typedef CnfMirrorT<StepState> HostMirror;
    HostMirror* h_mirror = nullptr;        // host address
    HostMirror* d_mirror = nullptr;        // the same memory as the device sees it
    unsigned mirror_base = 1;              // launch indices are monotonic over the handle's life: late launches of
                                           // an earlier solve can never look like news of the current one
    // lock-step sharded solves: host callback summing 3 floats over the shards (null: off)
    cnf_shard_reduce_fn shard_reduce = nullptr;
    void* shard_user = nullptr;
    cnf_comm shard_comm = nullptr;         // ... or an RCCL communicator: reduced on the stream, no host round trip
    // gradient path (cnf_loss_grad): transposed weights, per-step trajectory, adjoint scratch
    float* d_PT = nullptr;
    float* d_adj_img = nullptr;   // padded forward/reverse weight images of the MFMA pullback kernel
    bool pt_valid = false;
    bool img_valid = false;       // d_adj_img holds the images of the current parameters
    float* d_bimg = nullptr;      // k_solve_bcast's images (cnf_bcast.hip), packed on first use after a parameter change
    float* d_gt = nullptr;        // k_adj_test: one partial of the flat gradient and a scratch per workgroup (cnf_gradt.hip)
    size_t gt_floats = 0;
    float* d_bstore = nullptr;    // ... and the store of its tiles' Runge-Kutta rows when a workgroup carries several (bcast_store_floats)
    size_t bstore_floats = 0;
    bool bimg_valid = false;
    bool trace_on = false;        // this call evaluates through an auxiliary MFMA kernel (cnf_trace.hip) behind the generic driver
    bool aux_train = false;       //   false: TestMode exact trace; true: TrainMode JVP
    const float* aux_eps = nullptr;
    float* traj = nullptr;             // trajectory store: traj_cap slots of 6 (n_in + 3) grad_cap_B floats (u_n, U_2..U_6)
    float* traj_hs = nullptr;          // device: signed size of accepted step n (written by the step kernel)
    int traj_cap = 0;
    size_t grad_cap_B = 0;
    int grad_fsteps = 1;          // steps whose factor arrays are kept before one batch contraction
    float* grad_arena = nullptr;
    float* g_US[5] = {};          // stage states 2..6
    float* g_W[6] = {};           // zbar per stage
    float* g_lam = nullptr;
    float *g_HS = nullptr, *g_TS = nullptr, *g_AB = nullptr, *g_PB = nullptr;
    float* d_park = nullptr;      // parked state of the two-launch headline pullback (adj3b_park_floats)
    size_t park_floats = 0;
    float* d_sc = nullptr;        // scratch rows of the two-launch MFMA pullback (adj_mfma_scratch_floats)
    size_t sc_floats = 0;
    AdjStepArgs* d_steps = nullptr;      // ... and the arguments of the steps of its runs: device array and pinned staging
    AdjStepArgs* h_steps = nullptr;
    int steps_cap = 0;
    float* g_part = nullptr;      // GRAD_MAX_KSPLIT x n_params
    NetDesc nd_wave{};            // what the wave kernels see: nd, or a one-layer tanh network with an identity layer appended
    float* wg_traj = nullptr;     // k_solve_wave<GRAD>: z rows of u_n per accepted step, as the lanes hold them; + WV_GCAP step sizes
    size_t wg_traj_floats = 0;
    float* g_grad = nullptr;      // n_params (host-pointer variant)
    std::vector<float> last_hs;   // signed step sizes of the last cnf_loss_grad solve
    int grad_last_B = 0;          // batch of the last cnf_loss_grad call (g_lam holds its d loss / d u(t0))
    float* d_ys = nullptr;        // conditional models: copy of ys (n_cond x cond_B), kept for the weight gradient
    float* stage = nullptr;       // device staging area of the *_host entry points, owned by the handle, grown on demand
    size_t stage_cap = 0;         //   (floats): no allocation per call, nothing to free on an error path
    float* d_sums = nullptr;      // 3 floats
    float* h_sums = nullptr;      // pinned, 3 floats
    std::string err;
};

static const int MAX_PARTIALS = 1024;

#define HIPCHK(h, call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            char buf_[512];                                                              \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call,                  \
                     hipGetErrorString(e_), __FILE__, __LINE__);                         \
            if (h) (h)->err = buf_;                                                      \
            return CNF_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

static cnf_status fail(cnf_handle h, cnf_status s, const char* msg) {
    if (h) h->err = msg;
    return s;
}

static inline int rows_of(const cnf_ctx* h, int mode) {
    return h->nd.n_in + 1 + (mode == CNF_MODE_TRAIN ? 2 : 0);
}

extern "C" const char* cnf_status_string(cnf_status s) {
    switch (s) {
        case CNF_OK: return "ok";
        case CNF_ERR_BAD_ARG: return "bad argument";
        case CNF_ERR_BAD_SHAPE: return "bad shape";
        case CNF_ERR_HIP: return "HIP runtime error";
        case CNF_ERR_NO_DEVICE: return "no gfx950 device";
        case CNF_ERR_MAXITERS: return "maxiters reached";
        case CNF_ERR_UNSUPPORTED: return "unsupported configuration";
        case CNF_ERR_NO_PARAMS: return "parameters not set";
        case CNF_ERR_NONFINITE: return "non-finite solver state";
        case CNF_ERR_RCCL: return "RCCL error";
    }
    return "unknown status";
}

extern "C" const char* cnf_last_error(cnf_handle h) { return h ? h->err.c_str() : "null handle"; }
extern "C" int cnf_abi_version(void) { return CNF_ABI_VERSION; }
extern "C" int cnf_state_rows(cnf_handle h, int mode) { return h ? rows_of(h, mode) : -1; }

// ---------------------------------------------------------------------------------------
extern "C" cnf_status cnf_create(cnf_handle* out, const cnf_config* cfg) {
    if (!out || !cfg || !cfg->dims || !cfg->acts) return CNF_ERR_BAD_ARG;
    *out = nullptr;
    if (cfg->n_layers < 1 || cfg->n_layers > CNF_MAX_LAYERS) return CNF_ERR_BAD_SHAPE;
    if (cfg->nvars < 1 || cfg->naugs < 0) return CNF_ERR_BAD_SHAPE;
    if (cfg->ad != CNF_AD_VJP && cfg->ad != CNF_AD_JVP) return CNF_ERR_BAD_ARG;
    const int n_in = cfg->nvars + cfg->naugs;
    if (cfg->n_cond < 0) return CNF_ERR_BAD_SHAPE;
    if (cfg->dims[0] != n_in + cfg->n_cond || cfg->dims[cfg->n_layers] != n_in) return CNF_ERR_BAD_SHAPE;
    for (int l = 0; l <= cfg->n_layers; ++l)
        if (cfg->dims[l] < 1 || cfg->dims[l] > 4096) return CNF_ERR_BAD_SHAPE;
    for (int l = 0; l < cfg->n_layers; ++l)
        if (cfg->acts[l] < CNF_ACT_IDENTITY || cfg->acts[l] > CNF_ACT_ELU) return CNF_ERR_BAD_ARG;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CNF_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return CNF_ERR_BAD_ARG;

    cnf_ctx* h = new (std::nothrow) cnf_ctx();
    if (!h) return CNF_ERR_BAD_ARG;
    h->device = cfg->device;
    NetDesc& nd = h->nd;
    nd.n_layers = cfg->n_layers;
    int off = 0, mx = 0, sum = 0;
    for (int l = 0; l <= cfg->n_layers; ++l) {
        nd.dims[l] = l == 0 ? n_in : cfg->dims[l];     // the kernels see the z columns only
        mx = nd.dims[l] > mx ? nd.dims[l] : mx;
        sum += nd.dims[l];
    }
    nd.n_cond = cfg->n_cond;
    for (int l = 0; l < cfg->n_layers; ++l) {
        nd.acts[l] = cfg->acts[l];
        nd.w_off[l] = off;
        if (l == 0) nd.wy_off = off + n_in * nd.dims[1];
        off += cfg->dims[l] * nd.dims[l + 1];          // the flat vector holds every column
        nd.b_off[l] = off;
        off += nd.dims[l + 1];
    }
    h->n_params = (size_t)off;
    nd.n_in = n_in;
    nd.nvars = cfg->nvars;
    nd.naugs = cfg->naugs;
    nd.norm_z = cfg->lambda1 != 0.f;       // src/base_icnf.jl:48
    nd.norm_j = cfg->lambda2 != 0.f;       // src/base_icnf.jl:49
    nd.norm_z_aug = cfg->lambda3 != 0.f;   // src/base_icnf.jl:50
    nd.jvp = cfg->ad == CNF_AD_JVP;
    nd.max_dim = mx;
    nd.sum_dims = sum;
    h->lam[0] = cfg->lambda1; h->lam[1] = cfg->lambda2; h->lam[2] = cfg->lambda3;

    // A ONE-layer tanh network (`Dense(n => n, tanh)`: the network of the reference's benchmark suite, benchmark/benchmarks.jl:29)
    // runs on the two-layer wave kernels as (that layer, identity): W_2 = I and b_2 = 0 are kept behind the parameters.
    h->nd_wave = nd;
    size_t id_tail = 0;
    if (cfg->n_layers == 1 && cfg->acts[0] == CNF_ACT_TANH && n_in <= 16) {
        NetDesc& w = h->nd_wave;
        w.n_layers = 2; w.dims[2] = n_in; w.acts[1] = CNF_ACT_IDENTITY;
        w.w_off[1] = off; w.b_off[1] = off + n_in * n_in;
        w.sum_dims = sum + n_in; w.id2 = 1;
        id_tail = (size_t)n_in * n_in + n_in;
    }
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipMalloc(&h->d_params, (h->n_params + id_tail) * sizeof(float));
    if (e == hipSuccess && id_tail) {
        std::vector<float> idm(id_tail, 0.f);
        for (int i = 0; i < n_in; ++i) idm[(size_t)i * n_in + i] = 1.f;
        e = hipMemcpy(h->d_params + h->n_params, idm.data(), id_tail * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&h->d_state, 2 * sizeof(StepState));
    // (the one-launch solve uses it as 8-byte words: 2 x 1024 of the meetings, 2048 of the loss-sum partials)
    if (e == hipSuccess) e = hipMalloc(&h->partials, 8 * MAX_PARTIALS * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->partials, 0, 8 * MAX_PARTIALS * sizeof(float));
    if (e == hipSuccess) e = hipHostMalloc(&h->h_state, 3 * sizeof(StepState), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&h->d_sums, 24 * sizeof(float));      // 8 floats of sums, tickets, the kernel clock words
    if (e == hipSuccess) e = hipMemset(h->d_sums, 0, 24 * sizeof(float));      // words 8.. are device tickets: zero between launches
    if (e == hipSuccess) e = hipHostMalloc(&h->h_sums, 4 * sizeof(float), hipHostMallocDefault);
    // (four slots: the streamed solves use slot 0; the one-launch solves take slots 1, 2, 3 in turn, so that up to three
    // submitted launches can be in flight -- and one of them be run again on the streamed driver -- without overwriting
    // each other's final state)
    if (e == hipSuccess) e = hipHostMalloc(&h->h_mirror, 4 * sizeof(*h->h_mirror), hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->d_mirror, h->h_mirror, 0);
    if (e == hipSuccess) memset(h->h_mirror, 0, 4 * sizeof(*h->h_mirror));     // tag 0 is never a launch index (mirror_base starts at 1)
    if (e != hipSuccess) {
        cnf_destroy(h);
        return CNF_ERR_HIP;
    }
    mfma_plan_init(h->mfma, nd);
    *out = h;
    return CNF_OK;
}

// submitted launches still in flight read the handle's buffers: calls that change or free them wait for those launches
// (the submissions stay collectable: their outcome sits in the host mirror)
static void release_submitted(cnf_handle h);
static void sync_submitted(cnf_handle h) {
    for (const auto& sub : h->submitted)
        if (sub.launched) (void)hipStreamSynchronize(sub.st);
}
// Before the parameters or the conditioning of a handle change, every inference submitted on it is brought to its END --
// outcome read, and a launch that ran out of a wait run again on the streamed driver NOW, with the parameters and the
// conditioning it was submitted with -- and stays in the queue as a completed entry for its collect call.
static cnf_status settle_submitted(cnf_handle h);

extern "C" cnf_status cnf_destroy(cnf_handle h) {
    if (!h) return CNF_ERR_BAD_ARG;
    // called from a finaliser after the HIP runtime has shut down (process exit): nothing left to free on the device
    if (hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); delete h; return CNF_OK; }
    sync_submitted(h);
    release_submitted(h);
    mfma_plan_free(h->mfma);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_cond) (void)hipFree(h->d_cond);
    if (h->d_ys) (void)hipFree(h->d_ys);
    if (h->d_PT) (void)hipFree(h->d_PT);
    if (h->d_adj_img) (void)hipFree(h->d_adj_img);
    if (h->d_park) (void)hipFree(h->d_park);
    if (h->d_sc) (void)hipFree(h->d_sc);
    if (h->d_steps) (void)hipFree(h->d_steps);
    if (h->h_steps) (void)hipHostFree(h->h_steps);
    if (h->d_bimg) (void)hipFree(h->d_bimg);
    if (h->d_bstore) (void)hipFree(h->d_bstore);
    if (h->d_gt) (void)hipFree(h->d_gt);
    if (h->grad_arena) (void)hipFree(h->grad_arena);
    if (h->traj) (void)hipFree(h->traj);
    if (h->traj_hs) (void)hipFree(h->traj_hs);
    if (h->wg_traj) (void)hipFree(h->wg_traj);
    if (h->arena) (void)hipFree(h->arena);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->partials) (void)hipFree(h->partials);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->h_mirror) (void)hipHostFree(h->h_mirror);
    if (h->stage) (void)hipFree(h->stage);
    if (h->d_sums) (void)hipFree(h->d_sums);
    if (h->h_sums) (void)hipHostFree(h->h_sums);
    delete h;
    return CNF_OK;
}

extern "C" cnf_status cnf_set_params(cnf_handle h, const float* flat_dev, size_t n, void* stream) {
    if (!h || !flat_dev) return CNF_ERR_BAD_ARG;
    if (n != h->n_params) return fail(h, CNF_ERR_BAD_SHAPE, "parameter count does not match the layer sizes");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    { const cnf_status ss = settle_submitted(h); if (ss != CNF_OK) return ss; }
    HIPCHK(h, hipMemcpyAsync(h->d_params, flat_dev, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    cnf_status ms = mfma_plan_pack(h->mfma, h->nd, h->d_params, s);
    if (ms != CNF_OK) return fail(h, ms, "MFMA weight packing failed");
    HIPCHK(h, hipStreamSynchronize(s));
    h->have_params = true;
    h->pt_valid = false;
    h->img_valid = false;
    h->bimg_valid = false;
    h->cond_B = 0;       // the conditioning bias depends on W1 and b1
    return CNF_OK;
}

extern "C" cnf_status cnf_set_params_host(cnf_handle h, const float* flat, size_t n) {
    if (!h || !flat) return CNF_ERR_BAD_ARG;
    if (n != h->n_params) return fail(h, CNF_ERR_BAD_SHAPE, "parameter count does not match the layer sizes");
    HIPCHK(h, hipSetDevice(h->device));
    { const cnf_status ss = settle_submitted(h); if (ss != CNF_OK) return ss; }
    HIPCHK(h, hipMemcpy(h->d_params, flat, n * sizeof(float), hipMemcpyHostToDevice));
    cnf_status ms = mfma_plan_pack(h->mfma, h->nd, h->d_params, nullptr);
    if (ms != CNF_OK) return fail(h, ms, "MFMA weight packing failed");
    HIPCHK(h, hipDeviceSynchronize());
    h->have_params = true;
    h->pt_valid = false;
    h->img_valid = false;
    h->bimg_valid = false;
    h->cond_B = 0;
    return CNF_OK;
}

// ---------------------------------------------------------------------------------------
static cnf_status ensure_capacity(cnf_handle h, int B) {
    if ((size_t)B <= h->cap_B) return CNF_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    if (h->arena) { (void)hipFree(h->arena); h->arena = nullptr; h->cap_B = 0; }
    size_t cap = ((size_t)B + 255) & ~(size_t)255;
    const size_t Dmax = (size_t)h->nd.n_in + 3;
    const size_t ws_f = ((size_t)2 * h->nd.sum_dims + (size_t)2 * h->nd.max_dim) * cap;
    const size_t st_f = Dmax * cap;
    const size_t total = ws_f + 9 * st_f + 4 * cap + cap / 16;
    HIPCHK(h, hipMalloc(&h->arena, total * sizeof(float)));
    float* p = h->arena;
    h->ws = p; p += ws_f;
    for (int i = 0; i < 2; ++i) { h->U[i] = p; p += st_f; }
    for (int i = 0; i < 2; ++i) { h->K1[i] = p; p += st_f; }
    for (int i = 0; i < 5; ++i) { h->Ks[i] = p; p += st_f; }
    h->tmp_logpx = p; p += cap;
    h->tmp_regs = p; p += 3 * cap;
    h->post_part = p; p += cap / 16;
    h->cap_B = cap;
    return CNF_OK;
}

// staging area for host-pointer calls: at least `nfloats` floats of device memory owned by the handle
static cnf_status ensure_stage(cnf_handle h, size_t nfloats) {
    if (nfloats <= h->stage_cap) return CNF_OK;
    HIPCHK(h, hipDeviceSynchronize());
    if (h->stage) { (void)hipFree(h->stage); h->stage = nullptr; h->stage_cap = 0; }
    const size_t cap = (nfloats + 4095) & ~(size_t)4095;
    HIPCHK(h, hipMalloc(&h->stage, cap * sizeof(float)));
    h->stage_cap = cap;
    return CNF_OK;
}

static cnf_status collect_one(cnf_handle h, cnf_solve_stats* stats);
static void release_submitted(cnf_handle h);
static cnf_status check_call(cnf_handle h, int mode, int B) {
    if (!h) return CNF_ERR_BAD_ARG;
    if (mode != CNF_MODE_TEST && mode != CNF_MODE_TRAIN) return fail(h, CNF_ERR_BAD_ARG, "unknown mode");
    if (B < 0) return fail(h, CNF_ERR_BAD_SHAPE, "negative batch");
    if (!h->have_params) return fail(h, CNF_ERR_NO_PARAMS, "cnf_set_params has not been called");
    if (h->nd.n_cond > 0 && B > 0 && h->cond_B != B)
        return fail(h, CNF_ERR_NO_PARAMS, "conditional model: call cnf_set_cond with the ys of this batch first");
    h->mfma.cond = h->nd.n_cond > 0 ? h->d_cond : nullptr;
    h->mfma.cbs = h->cbs;
    // any other call on the handle first completes the inferences submitted on it (their statistics are dropped)
    while (!h->collecting && !h->submitted.empty()) {
        const cnf_status s = collect_one(h, nullptr);
        if (s != CNF_OK) return s;
    }
    return CNF_OK;
}

extern "C" cnf_status cnf_set_cond(cnf_handle h, const float* ys, int B, void* stream) {
    if (!h) return CNF_ERR_BAD_ARG;
    if (h->nd.n_cond == 0) return fail(h, CNF_ERR_BAD_ARG, "not a conditional model (n_cond == 0)");
    if (!h->have_params) return fail(h, CNF_ERR_NO_PARAMS, "cnf_set_params has not been called");
    if (!ys || B < 1) return fail(h, CNF_ERR_BAD_ARG, "ys must be n_cond x B with B >= 1");
    HIPCHK(h, hipSetDevice(h->device));
    // (a submitted inference reads d_cond until it ends, and its fallback would read it again: none may be outstanding)
    { const cnf_status ss = settle_submitted(h); if (ss != CNF_OK) return ss; }
    const int cbs = (h->nd.dims[1] + 15) & ~15;
    if (h->cond_B != B || h->cbs != cbs) {
        HIPCHK(h, hipDeviceSynchronize());
        if (h->d_cond) { (void)hipFree(h->d_cond); h->d_cond = nullptr; }
        if (h->d_ys) { (void)hipFree(h->d_ys); h->d_ys = nullptr; }
        h->cond_B = 0;
        HIPCHK(h, hipMalloc(&h->d_cond, (size_t)B * cbs * sizeof(float)));
        HIPCHK(h, hipMalloc(&h->d_ys, (size_t)B * h->nd.n_cond * sizeof(float)));
        h->cbs = cbs;
    }
    // the gradient path needs ys itself (d loss / d W1[:, n_in:] = sum_b abar_1 ys')
    HIPCHK(h, hipMemcpyAsync(h->d_ys, ys, (size_t)B * h->nd.n_cond * sizeof(float), hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    launch_cond_bias(h->nd, h->d_params, ys, h->d_cond, cbs, B, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    h->cond_B = B;
    return CNF_OK;
}

extern "C" cnf_status cnf_set_cond_host(cnf_handle h, const float* ys, int B) {
    if (!h) return CNF_ERR_BAD_ARG;
    if (!ys || B < 1 || h->nd.n_cond == 0) return fail(h, CNF_ERR_BAD_ARG, "bad conditioning input");
    HIPCHK(h, hipSetDevice(h->device));
    cnf_status s = ensure_stage(h, (size_t)B * h->nd.n_cond);
    if (s != CNF_OK) return s;
    float* d = h->stage;
    HIPCHK(h, hipMemcpy(d, ys, (size_t)B * h->nd.n_cond * sizeof(float), hipMemcpyHostToDevice));
    s = cnf_set_cond(h, d, B, nullptr);
    hipError_t e = hipDeviceSynchronize();      // cnf_set_cond keeps its own copy of ys: the staging area is free again
    if (s == CNF_OK && e != hipSuccess) s = fail(h, CNF_ERR_HIP, hipGetErrorString(e));
    return s;
}

static bool trace_ok(cnf_handle h);
static bool jvp_aux_ok(cnf_handle h);
extern "C" int cnf_kernel_for(cnf_handle h, int mode, int B) {
    if (!h) return -1;
    if (mfma_supported(h->mfma, h->nd, mode == CNF_MODE_TRAIN, B)) return CNF_KERNEL_MFMA;
    if (mode == CNF_MODE_TRAIN) return jvp_aux_ok(h) ? CNF_KERNEL_MFMA : CNF_KERNEL_GENERIC;
    return trace_ok(h) ? CNF_KERNEL_MFMA : CNF_KERNEL_GENERIC;
}

// padded forward/reverse weight images shared by the pullback and the exact-trace kernels
static cnf_status ensure_adj_images(cnf_handle h, hipStream_t st) {
    const GradLayout g = grad_layout(h->nd);
    const AdjMfmaLayout m = adj_mfma_layout(h->nd, g);
    if (!h->d_adj_img) HIPCHK(h, hipMalloc(&h->d_adj_img, (size_t)m.img_floats * sizeof(float)));
    if (!h->img_valid) {
        HIPCHK(h, launch_pack_adj_images(h->nd, g, m, h->d_params, h->d_adj_img, st));
        h->img_valid = true;
    }
    return CNF_OK;
}

static bool trace_ok(cnf_handle h) {
    const GradLayout g = grad_layout(h->nd);
    return trace_mfma_supported(h->nd, adj_mfma_layout(h->nd, g));
}
// TrainMode with the JVP compute mode on a network the fused step kernel cannot hold
static bool jvp_aux_ok(cnf_handle h) {
    const GradLayout g = grad_layout(h->nd);
    return jvp_mfma_supported(h->nd, adj_mfma_layout(h->nd, g));
}

// one evaluation with an auxiliary MFMA kernel (TestMode: exact trace; TrainMode: JVP): u -> du (or k7)
// nk > 0 (inside a solve): the kernel forms the Runge-Kutta stage state from nk stage derivatives itself (and, for
// stage 6, stores it as the new solution); otherwise it evaluates at `u`
struct TraceFuse {          // norms / controller / the whole attempt inside the trace launch (k_trace3s; see TraceArgs)
    int norm_kind = -1;     // 0 / 1: norms of the automatic initial dt; 2: the error norm of an attempt
    bool step = false;      // the six stage evaluations of an attempt in one launch
    void* mirror = nullptr;
    unsigned seq = 0;
};
static bool trace_fused_ok(cnf_handle h, int B) {
    const GradLayout g = grad_layout(h->nd);
    return !h->aux_train && trace_fused_supported(h->nd, adj_mfma_layout(h->nd, g), B);
}
static cnf_status launch_trace(cnf_handle h, const float* u, float* du, bool in_solve, bool du_is_k7, int B, hipStream_t st,
                               int nk = 0, const float* coef = nullptr, bool also_unew = false, const TraceFuse* fuse = nullptr) {
    const GradLayout g = grad_layout(h->nd);
    const AdjMfmaLayout m = adj_mfma_layout(h->nd, g);
    TraceArgs a{};
    a.u = u; a.du = du; a.ys = h->nd.n_cond > 0 ? h->d_ys : nullptr;
    a.st = in_solve ? h->d_state : nullptr;
    a.K1[0] = h->K1[0]; a.K1[1] = h->K1[1];
    a.du_is_k7 = du_is_k7 ? 1 : 0;
    a.B = B;
    a.nk = in_solve ? nk : 0;
    for (int i = 0; i < 2; ++i) a.U[i] = h->U[i];
    for (int i = 0; i < 5; ++i) a.Ks[i] = h->Ks[i];
    for (int i = 0; i < a.nk && i < 6; ++i) a.coef[i] = coef[i];
    a.also_unew = also_unew ? 1 : 0;
    a.norm_kind = -1;
    if (fuse) {
        a.norm_kind = fuse->norm_kind; a.fused_step = fuse->step ? 1 : 0;
        a.partials = h->partials; a.ticket = reinterpret_cast<unsigned*>(h->d_sums + 8); a.st_mut = h->d_state;
        a.n_total = (float)((size_t)rows_of(h, CNF_MODE_TEST) * B);
        a.mirror = fuse->mirror; a.seq = fuse->seq;
    }
    if (h->aux_train && h->nd.jvp) HIPCHK(h, launch_jvp_mfma(h->nd, g, m, h->d_adj_img, a, h->aux_eps, st));
    else HIPCHK(h, launch_trace_mfma(h->nd, g, m, h->d_adj_img, a, st));
    return CNF_OK;
}

static cnf_status resolve_kernel(cnf_handle h, int mode, int B, int requested, int* out) {
    const bool ok = mfma_supported(h->mfma, h->nd, mode == CNF_MODE_TRAIN, B) ||
                    (mode == CNF_MODE_TEST && trace_ok(h)) || (mode == CNF_MODE_TRAIN && jvp_aux_ok(h));
    if (requested == CNF_KERNEL_AUTO) { *out = ok ? CNF_KERNEL_MFMA : CNF_KERNEL_GENERIC; return CNF_OK; }
    if (requested == CNF_KERNEL_GENERIC) { *out = CNF_KERNEL_GENERIC; return CNF_OK; }
    if (requested == CNF_KERNEL_MFMA) {
        if (!ok) return fail(h, CNF_ERR_UNSUPPORTED, "the MFMA path has no kernel for this network/mode");
        *out = CNF_KERNEL_MFMA;
        return CNF_OK;
    }
    return fail(h, CNF_ERR_BAD_ARG, "unknown kernel selector");
}

extern "C" cnf_status cnf_solve_kernel_time(cnf_handle h, int enable, float* mean_us, int* launches) {
    if (!h) return CNF_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    unsigned long long w[3] = {0, 0, 0};
    HIPCHK(h, hipMemcpy(w, h->d_sums + 12, sizeof w, hipMemcpyDeviceToHost));
    if (mean_us) *mean_us = w[2] ? (float)((double)w[1] * 0.01 / (double)w[2]) : 0.f;       // s_memrealtime: 100 MHz
    if (launches) *launches = (int)w[2];
    HIPCHK(h, hipMemset(h->d_sums + 12, 0, sizeof w));
    h->time_kernel = enable != 0;
    return CNF_OK;
}

extern "C" cnf_status cnf_set_step_trace(cnf_handle h, float* trace_dev, int cap_attempts) {
    if (!h || cap_attempts < 0 || (cap_attempts > 0 && !trace_dev)) return CNF_ERR_BAD_ARG;
    h->step_trace = cap_attempts > 0 ? trace_dev : nullptr;
    h->step_trace_cap = cap_attempts;
    return CNF_OK;
}

extern "C" cnf_status cnf_selftest_split_product(const float* A, const float* Bt, float* C, int K) {
    if (!A || !Bt || !C || K < 32 || K % 32 != 0 || K > 4096) return CNF_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CNF_ERR_NO_DEVICE;
    float* d = nullptr;
    const size_t na = (size_t)16 * K;
    if (hipMalloc(&d, (2 * na + 256) * sizeof(float)) != hipSuccess) return CNF_ERR_HIP;
    hipError_t e = hipMemcpy(d, A, na * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + na, Bt, na * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = split_product_test_launch(d, d + na, d + 2 * na, K, nullptr);
    if (e == hipSuccess) e = hipMemcpy(C, d + 2 * na, 256 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return e == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

// test support: workgroups that hold a CU each (all of its LDS) for a bounded time
__global__ void __launch_bounds__(64) k_hold_cu(unsigned long long ticks) {
    extern __shared__ char hold_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) hold_lds[0] = 1;
}
extern "C" cnf_status cnf_selftest_hold_cus(int n_workgroups, int microseconds, void* stream) {
    if (n_workgroups < 1 || n_workgroups > 4096 || microseconds < 1 || microseconds > 100000) return CNF_ERR_BAD_ARG;
    constexpr int LDS = 160 * 1024;
    if (hipFuncSetAttribute((const void*)k_hold_cu, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return CNF_ERR_HIP;
    hipLaunchKernelGGL(k_hold_cu, dim3(n_workgroups), dim3(64), LDS, (hipStream_t)stream, 100ull * (unsigned long long)microseconds);
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}
extern "C" int cnf_solve_fallbacks(cnf_handle h) { return h ? h->fallbacks : -1; }
extern "C" int cnf_set_grad_split(int mode) {
    const int was = adj_split_mode();
    set_adj_split_mode(mode);
    return was;
}

extern "C" cnf_status cnf_set_solve_wait(cnf_handle h, int wait_us, int poll_limit) {
    if (!h) return CNF_ERR_BAD_ARG;
    if (wait_us > 0) h->wait_us = wait_us;
    if (poll_limit > 0) h->poll_limit = poll_limit;
    return CNF_OK;
}

extern "C" cnf_status cnf_rhs_work(cnf_handle h, int mode, int B, double* flops, double* bytes) {
    if (!h || !flops || !bytes) return CNF_ERR_BAD_ARG;
    const NetDesc& nd = h->nd;
    double M = 0, P = 0;
    for (int l = 0; l < nd.n_layers; ++l) {
        M += (double)nd.dims[l] * nd.dims[l + 1];
        P += (double)nd.dims[l] * nd.dims[l + 1] + nd.dims[l + 1];
    }
    const double n_in = nd.n_in, D = rows_of(h, mode);
    if (mode == CNF_MODE_TRAIN) {
        *flops = B * (4.0 * M + 6.0 * n_in);
        *bytes = 4.0 * B * (n_in + n_in + D) + 4.0 * P;
    } else {
        // exact trace: minimal formulation for 2-layer nets, tr J = d1^T (W1 .* W2^T) d2 = forward
        // + one h x n_in product (B*3M); otherwise forward + n_in tangent sweeps
        if (nd.n_layers == 2) *flops = B * 3.0 * M;
        else *flops = B * (2.0 * M + n_in * 2.0 * M);
        *bytes = 4.0 * B * (n_in + D) + 4.0 * P;
    }
    return CNF_OK;
}

// ---------------------------------------------------------------------------------------
// RHS (a1/a2/a3)
// ---------------------------------------------------------------------------------------
extern "C" cnf_status cnf_rhs(cnf_handle h, int mode, int kernel, const float* u,
                              const float* eps, float* du, int B, void* stream) {
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!u || !du) return fail(h, CNF_ERR_BAD_ARG, "null state pointer");
    if (u == du) return fail(h, CNF_ERR_BAD_ARG, "u and du must not alias");
    if (mode == CNF_MODE_TRAIN && !eps) return fail(h, CNF_ERR_BAD_ARG, "eps is required in TrainMode");
    if (B == 0) return CNF_OK;
    int k;
    if ((s = resolve_kernel(h, mode, B, kernel, &k)) != CNF_OK) return s;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (k == CNF_KERNEL_MFMA && !mfma_supported(h->mfma, h->nd, mode == CNF_MODE_TRAIN, B)) {
        // TestMode, three or more layers: exact trace on MFMA; TrainMode/JVP beyond LDS (cnf_trace.hip)
        if ((s = ensure_adj_images(h, st)) != CNF_OK) return s;
        h->aux_train = mode == CNF_MODE_TRAIN; h->aux_eps = eps;
        if ((s = launch_trace(h, u, du, false, false, B, st)) != CNF_OK) return s;
    } else if (k == CNF_KERNEL_MFMA) {
        s = mfma_rhs(h->mfma, h->nd, mode == CNF_MODE_TRAIN, u, eps, du, B, st);
        if (s != CNF_OK) return fail(h, s, "MFMA RHS launch failed");
    } else {
        if ((s = ensure_capacity(h, B)) != CNF_OK) return s;
        RhsArgs a{};
        a.st = nullptr; a.B = B; a.S = h->cap_B; a.train = mode == CNF_MODE_TRAIN;
        a.ws = h->ws; a.eps = eps; a.u = u; a.du = du; a.nk = 0;
        a.cond = h->mfma.cond; a.cbs = h->cbs;
        launch_rhs_generic(h->nd, h->d_params, a, st);
    }
    HIPCHK(h, hipGetLastError());
    return CNF_OK;
}

extern "C" cnf_status cnf_rhs_host(cnf_handle h, int mode, int kernel, const float* u,
                                   const float* eps, float* du, int B) {
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!u || !du) return fail(h, CNF_ERR_BAD_ARG, "null state pointer");
    if (B == 0) return CNF_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t D = rows_of(h, mode), n_in = h->nd.n_in;
    if ((s = ensure_stage(h, (2 * D + n_in) * B)) != CNF_OK) return s;
    float* u_d = h->stage;
    float* du_d = u_d + D * B;
    float* e_d = eps ? du_d + D * B : nullptr;
    HIPCHK(h, hipMemcpy(u_d, u, D * B * sizeof(float), hipMemcpyHostToDevice));
    if (eps) HIPCHK(h, hipMemcpy(e_d, eps, n_in * B * sizeof(float), hipMemcpyHostToDevice));
    s = cnf_rhs(h, mode, kernel, u_d, e_d, du_d, B, nullptr);
    if (s == CNF_OK) {
        hipError_t e = hipMemcpy(du, du_d, D * B * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) s = fail(h, CNF_ERR_HIP, hipGetErrorString(e));
    }
    return s;
}

// ---------------------------------------------------------------------------------------
// Tsit5 driver (a7: base_sol, src/base_icnf.jl:137-143)
// ---------------------------------------------------------------------------------------
static int enqueue_attempt_generic(cnf_handle h, int train, const float* eps, int B,
                                    int nblk, hipStream_t s, bool with_controller = true,
                                    float* dump = nullptr, size_t dump_stride = 0, void* mirror = nullptr,
                                    unsigned seq = 0) {
    RhsArgs a{};
    a.st = h->d_state; a.B = B; a.S = h->cap_B; a.train = train; a.ws = h->ws; a.eps = eps;
    a.cond = h->mfma.cond; a.cbs = h->cbs;
    for (int i = 0; i < 2; ++i) { a.U[i] = h->U[i]; a.K1[i] = h->K1[i]; }
    for (int i = 0; i < 5; ++i) a.Ks[i] = h->Ks[i];
    if (h->trace_on && with_controller && !dump && trace_fused_ok(h, B)) {
        // TestMode on the 32-128-128-32 shape: the six stage evaluations, the error norm and the controller in ONE launch
        TraceFuse f; f.norm_kind = 2; f.step = true; f.mirror = mirror; f.seq = seq;
        (void)launch_trace(h, nullptr, nullptr, true, false, B, s, 0, nullptr, false, &f);
        return 1;
    }
    for (int stage = 1; stage <= 6 && h->trace_on; ++stage) {     // the auxiliary kernel forms its stage state itself
        float coef[6];
        tsit5_row(stage, coef);
        (void)launch_trace(h, nullptr, stage < 6 ? h->Ks[stage - 1] : nullptr, true, stage == 6, B, s, stage, coef, stage == 6);
    }
    for (int stage = 1; stage <= 6 && !h->trace_on; ++stage) {      // computes k_{stage+1}
        a.nk = stage;
        tsit5_row(stage, a.coef);
        a.ustage = (dump && stage < 6) ? dump + (size_t)(stage - 1) * dump_stride : nullptr;   // stage states 2..6
        a.ustage_is_unew = stage == 6;
        a.du_is_k7 = stage == 6;
        a.du = stage < 6 ? h->Ks[stage - 1] : nullptr;
        launch_rhs_generic(h->nd, h->d_params, a, s);
    }
    NormArgs n{};
    n.st = h->d_state; n.kind = 2; n.n = (size_t)rows_of(h, train) * B;
    for (int i = 0; i < 2; ++i) { n.U[i] = h->U[i]; n.K1[i] = h->K1[i]; }
    for (int i = 0; i < 5; ++i) n.Ks[i] = h->Ks[i];
    n.partials = h->partials;
    if (with_controller) {      // error norm + controller in one launch
        n.ticket = reinterpret_cast<unsigned*>(h->d_sums + 8); n.st_mut = h->d_state; n.ctrl_phase = 2; n.n_total = (float)n.n;
        n.mirror = mirror; n.seq = seq;
    }
    launch_norm_partials(n, nblk, s);
    return 7;                            // six stage evaluations + the error norm (with the controller)
}

extern "C" cnf_status cnf_set_shard_reduce(cnf_handle h, cnf_shard_reduce_fn fn, void* user) {
    if (!h) return CNF_ERR_BAD_ARG;
    h->shard_reduce = fn;
    h->shard_user = user;
    return CNF_OK;
}

// Lock-step controller: local partial sums -> (p0, p1, n_local) -> host -> sum over the shards ->
// back to the device -> controller on the global sums.  One stream synchronisation per call.
static cnf_status lockstep_controller(cnf_handle h, StepState* state, const float* partials, int phase,
                                      float n_local, hipStream_t st) {
    launch_reduce_partials(state, partials, h->d_sums, n_local, st);
    if (h->shard_comm) {                 // RCCL: the three floats never leave the device
        if (cnf_comm_allreduce(h->shard_comm, h->d_sums, 3, st) != CNF_OK) return fail(h, CNF_ERR_RCCL, cnf_comm_last_error());
        launch_controller_sums(state, h->d_sums, phase, st);
        return CNF_OK;
    }
    HIPCHK(h, hipMemcpyAsync(h->h_sums, h->d_sums, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (h->shard_reduce(h->h_sums, 3, h->shard_user) != 0)
        return fail(h, CNF_ERR_BAD_ARG, "shard_reduce callback reported failure");
    HIPCHK(h, hipMemcpyAsync(h->d_sums, h->h_sums, 3 * sizeof(float), hipMemcpyHostToDevice, st));
    launch_controller_sums(state, h->d_sums, phase, st);
    return CNF_OK;
}

// Trajectory store of the gradient path: state after every accepted step + the step sizes.
struct Recorder {
    std::vector<float> hs;        // signed step of accepted step n (u_n -> u_{n+1})
    int n = 0;                    // accepted steps recorded; slot n holds u_n and the stage states of step n
    bool overflow = false;        // the solve took more steps than the store holds: grow it and solve again
    // the whole gradient in the launch of the solve (k_solve_wave<GRAD>): where it keeps its trajectory and leaves its results;
    // wg_done: it ran (hs holds the step sizes); wg_failed: it could not (not this network, a wait ran out, too many steps)
    const WaveGradArgs* wg = nullptr;
    bool wg_done = false, wg_failed = false, wg_submitted = false;     // wg_submitted: launched and left in flight (cnf_loss_grad_submit)
};
static size_t traj_slot_floats(cnf_handle h) { return 6 * ((size_t)h->nd.n_in + 3) * h->grad_cap_B; }
// make room for `steps` slots (contiguous: the step kernel indexes it by the accepted-step counter)
static cnf_status traj_reserve(cnf_handle h, int steps) {
    if (steps <= h->traj_cap) return CNF_OK;
    int cap = h->traj_cap ? h->traj_cap : 32;
    while (cap < steps) cap *= 2;
    const size_t slot = traj_slot_floats(h);
    float *nt = nullptr, *nh = nullptr;
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMalloc(&nt, slot * cap * sizeof(float)));
    if (hipMalloc(&nh, (size_t)cap * sizeof(float)) != hipSuccess) {
        (void)hipFree(nt);
        return fail(h, CNF_ERR_HIP, "hipMalloc of the trajectory step-size array failed");
    }
    if (h->traj) {
        hipError_t ce = hipMemcpy(nt, h->traj, slot * h->traj_cap * sizeof(float), hipMemcpyDeviceToDevice);
        if (ce == hipSuccess) ce = hipMemcpy(nh, h->traj_hs, (size_t)h->traj_cap * sizeof(float), hipMemcpyDeviceToDevice);
        if (ce != hipSuccess) {
            (void)hipFree(nt); (void)hipFree(nh);
            return fail(h, CNF_ERR_HIP, hipGetErrorString(ce));
        }
        (void)hipFree(h->traj); (void)hipFree(h->traj_hs);
    }
    h->traj = nt; h->traj_hs = nh; h->traj_cap = cap;
    return CNF_OK;
}
static cnf_status traj_slot(cnf_handle h, int n, float** out) {
    cnf_status s = traj_reserve(h, n + 1);
    if (s != CNF_OK) return s;
    *out = h->traj + (size_t)n * traj_slot_floats(h);
    return CNF_OK;
}

// Post-processing (and loss sums) the caller wants behind the solve.  The streamed driver enqueues it itself, right behind
// the attempt its step estimate says is the last one -- the kernel checks `done` and does nothing if the estimate was
// short -- so that it does not wait for the host to notice the end of the solve; `launched` tells the caller it ran.
struct PostHook {
    float* logpx; float* regs; float* sums5;
    const float* xs = nullptr;     // also assemble u0 from these columns, in the launch that sets the initial state
    bool launched = false;
};
static void enqueue_post(cnf_handle h, int train, const StepState* state, const PostHook& ph, int B, bool need_done,
                         hipStream_t st) {
    launch_post_state(h->nd, train, state, h->U[0], h->U[1], ph.logpx, ph.regs, B, st, need_done, ph.sums5, h->post_part,
                      reinterpret_cast<unsigned*>(h->d_sums + 9));
}
static cnf_status solve_core(cnf_handle h, int mode, const float* u0, const float* eps, float* u_out, int B,
                             const cnf_solve_opts* opts, cnf_solve_stats* stats, void* stream, Recorder* rec,
                             bool final_sync = true, PostHook* post = nullptr);

// One-launch solves of this process: one at a time (two would each hold CUs the other is waiting for), except that
// submitted launches may queue up behind each other on ONE stream.
static std::mutex g_persist_mu;
static int g_submitted_inflight = 0;               // (both under g_persist_mu)
static hipStream_t g_submitted_stream = nullptr;

// The outcome of the one-launch solve with launch index `seq`: its final state arrives in its slot of the host mirror.
// *aborted: a wait inside the kernel ran out (the state says so itself: n_partials < 0) -- nothing of the launch is used,
// the abort word is cleared (after the stream has drained: launches queued behind it read it too) and the caller runs
// the solve again on the streamed driver.
static cnf_status finish_one_launch(cnf_handle h, unsigned seq, int slot, hipStream_t st, StepState* fin, bool* aborted) {
    const volatile cnf_ctx::HostMirror* hm = h->h_mirror + slot;
    unsigned sq = 0;
    for (long spins = 0;; ++spins) {
        if (cnf_mirror_read(hm, fin, &sq) && sq == seq) break;
        if (spins < 4096) _mm_pause();
        else sched_yield();
        if (spins % 100000 == 99999) {
            hipError_t qe = hipStreamQuery(st);
            if (qe != hipSuccess && qe != hipErrorNotReady) HIPCHK(h, qe);
            if (qe == hipSuccess && !(cnf_mirror_read(hm, fin, &sq) && sq == seq))
                return fail(h, CNF_ERR_HIP, "the solve kernel finished without publishing a state");
        }
    }
    *aborted = fin->n_partials < 0;
    if (*aborted) {
        // A workgroup dispatched late may have run after workgroup 0 advanced the index base and left words tagged with
        // the NEXT launch's first meeting: with the stream drained, the meeting words are cleared along with the abort word.
        HIPCHK(h, hipStreamSynchronize(st));
        HIPCHK(h, hipMemset(h->partials, 0, 8 * MAX_PARTIALS * sizeof(float)));
        HIPCHK(h, hipMemset(h->d_sums + 11, 0, sizeof(float)));
    } else if (!fin->done && !fin->nonfinite) return fail(h, CNF_ERR_MAXITERS, "maxiters reached before t1");
    return CNF_OK;
}

extern "C" cnf_status cnf_solve_tsit5(cnf_handle h, int mode, const float* u0,
                                      const float* eps, float* u_out, int B,
                                      const cnf_solve_opts* opts, cnf_solve_stats* stats,
                                      void* stream) {
    if (h && !u_out) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    return solve_core(h, mode, u0, eps, u_out, B, opts, stats, stream, nullptr);
}

static cnf_status solve_core(cnf_handle h, int mode, const float* u0, const float* eps, float* u_out, int B,
                             const cnf_solve_opts* opts, cnf_solve_stats* stats, void* stream, Recorder* rec,
                             bool final_sync, PostHook* post) {
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!u0 || !opts) return fail(h, CNF_ERR_BAD_ARG, "null pointer");      // u_out may be null: the state stays in U[cur]
    const int train = mode == CNF_MODE_TRAIN;
    if (train && !eps) return fail(h, CNF_ERR_BAD_ARG, "eps is required in TrainMode");
    if (!(opts->t0 == opts->t0) || !(opts->t1 == opts->t1) || opts->t0 == opts->t1)
        return fail(h, CNF_ERR_BAD_ARG, "empty or NaN time span");
    if (!opts->adaptive && !(opts->dt > 0.f)) return fail(h, CNF_ERR_BAD_ARG, "fixed stepping needs dt > 0");
    if (opts->adaptive && (!(opts->abstol >= 0.f) || !(opts->reltol >= 0.f) || (opts->abstol == 0.f && opts->reltol == 0.f)))
        return fail(h, CNF_ERR_BAD_ARG, "bad tolerances");
    if (opts->dt < 0.f) return fail(h, CNF_ERR_BAD_ARG, "dt must be >= 0 (direction comes from the span)");
    if (opts->maxiters < 1) return fail(h, CNF_ERR_BAD_ARG, "maxiters must be >= 1");
    if (stats) memset(stats, 0, sizeof *stats);
    if (B == 0) return CNF_OK;
    int k;
    if ((s = resolve_kernel(h, mode, B, opts->kernel, &k)) != CNF_OK) return s;
    if ((s = ensure_capacity(h, B)) != CNF_OK) return s;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int D = rows_of(h, mode);
    const size_t n = (size_t)D * B;
    int launches = 0;

    const bool use_mfma = k == CNF_KERNEL_MFMA && mfma_supported(h->mfma, h->nd, train, B);
    h->trace_on = k == CNF_KERNEL_MFMA && !use_mfma;       // generic driver + an auxiliary MFMA kernel per stage
    h->aux_train = train != 0; h->aux_eps = eps;
    if (h->trace_on && (s = ensure_adj_images(h, (hipStream_t)stream)) != CNF_OK) return s;
    // lock-step over shards only matters when the controller decides something
    const bool lockstep = (h->shard_reduce != nullptr || h->shard_comm != nullptr) && opts->adaptive;
    // number of error partials = blocks of whichever kernel writes them
    int nblk = (int)((n + 255) / 256);
    if (nblk > 256) nblk = 256;
    if (use_mfma) nblk = mfma_grid_for(h->mfma, B, rec != nullptr, train != 0);
    // TestMode on k_trace3s: the trace launches write the error partials themselves (one pair per workgroup)
    const bool trace_fused = h->trace_on && !lockstep && !rec && trace_fused_ok(h, B);
    if (trace_fused) nblk = trace_fused_grid(B);

    // initial state
    StepState* init = &h->h_state[2];
    memset(init, 0, sizeof *init);
    init->t = init->t0 = opts->t0;
    init->t1 = opts->t1;
    init->tdir = opts->t1 >= opts->t0 ? 1.f : -1.f;
    init->dt = opts->dt;
    init->qold = 1e-4f;
    init->abstol = opts->abstol; init->reltol = opts->reltol;
    init->adaptive = opts->adaptive ? 1 : 0;
    init->n_partials = nblk;
    {
        float rem = fabsf(init->t1 - init->t);
        float hh = init->dt < rem ? init->dt : rem;
        init->h = init->tdir * hh;
    }
    const bool hairer = opts->adaptive && opts->dt == 0.f;
    // The whole solve in ONE launch where the handle and the batch allow it (k_solve3b / k_solve3jb): the weights and the
    // Runge-Kutta rows stay on the CUs for all attempts, the workgroups exchange two floats per attempt.
    // ... or of a small two-layer network, one wave per 16-sample tile, registers only (k_solve_wave, cnf_wave.hip)
    // (a one-layer tanh network has no other MFMA kernel: AUTO resolves to GENERIC for it, and the wave kernels take it
    // through its appended identity layer unless GENERIC was asked for)
    const bool wave_k = k == CNF_KERNEL_MFMA || (h->nd_wave.id2 && opts->kernel != CNF_KERNEL_GENERIC);
    const bool wave_ok = wave_k && (!rec || (rec->wg && post && post->xs && wave_grad_supported(h->nd_wave, B, train != 0))) &&
                         wave_solve_supported(h->nd_wave, train != 0, B);
    if (rec && rec->wg && !(wave_ok && !lockstep && !h->no_persist)) { rec->wg_failed = true; return CNF_OK; }
    // ... or of config 5's network at eight columns per CU (k_solve_bcast, cnf_bcast.hip)
    // (the gradient's recorded forward too, where one tile per workgroup holds the batch)
    const bool bcast_rec = rec && !rec->wg && train && bcast_store_floats(B, h->device) == 0;
    const bool bcast_ok = k == CNF_KERNEL_MFMA && (!rec || bcast_rec) && !wave_ok && bcast_solve_supported(h->nd, train != 0, B, h->device);
    if (bcast_ok && !lockstep && !h->no_persist) {
        if (!h->d_bimg) HIPCHK(h, hipMalloc(&h->d_bimg, bcast_img_floats() * sizeof(float)));
        if (!h->bimg_valid) { bcast_pack(h->nd, h->d_params, h->d_bimg, st); HIPCHK(h, hipGetLastError()); h->bimg_valid = true; }
        const size_t need = bcast_store_floats(B, h->device);        // several tiles per workgroup: their rows live in global memory
        if (need > h->bstore_floats) {
            HIPCHK(h, hipStreamSynchronize(st));
            if (h->d_bstore) { (void)hipFree(h->d_bstore); h->d_bstore = nullptr; h->bstore_floats = 0; }
            HIPCHK(h, hipMalloc(&h->d_bstore, need * sizeof(float)));
            h->bstore_floats = need;
        }
    }
    // ... or of the 32-128-128-32 network in TestMode (exact trace: k_trace3s<SOLVE>, cnf_trace.hip)
    const bool tsolve_ok = trace_fused && !wave_ok && !bcast_ok && !use_mfma && (!h->nd.n_cond) &&
                           trace_solve_supported(h->nd, adj_mfma_layout(h->nd, grad_layout(h->nd)), B, h->device);
    if ((use_mfma || wave_ok || bcast_ok || tsolve_ok) && !lockstep && !h->no_persist) {
        // one such kernel at a time in this process: two of them would each hold CUs the other is waiting for.  Launches
        // queued on ONE stream run one after the other by themselves (submitted inferences); a launch on another stream
        // waits for those first.
        std::unique_lock<std::mutex> persist_lock(g_persist_mu);
        if (g_submitted_inflight > 0 && g_submitted_stream != st) HIPCHK(h, hipStreamSynchronize(g_submitted_stream));
        const unsigned base = h->mirror_base;
        const int mslot = 1 + (int)(h->one_launch_count % 3);
        Solve3Args sv{};
        sv.part = h->partials; sv.base_dev = reinterpret_cast<unsigned*>(h->d_sums + 10); sv.abort_flag = reinterpret_cast<int*>(h->d_sums + 11);
        sv.t_out = h->time_kernel ? reinterpret_cast<unsigned long long*>(h->d_sums + 12) : nullptr;
        sv.maxiters = (int)opts->maxiters; sv.hairer = hairer ? 1 : 0; sv.init = *init;
        // How long a wait inside the kernel lasts before it gives up: bounded in TIME (the kernel's 100 MHz clock) -- 2 ms per
        // tile a workgroup carries (a meeting's arrivals are microseconds apart when every workgroup is placed; a workgroup
        // that is not placed because another tenant holds its CU makes the launch give up after that long, and the streamed
        // solve starts).  CNF_SOLVE_WAIT_US overrides; CNF_SOLVE_POLL_LIMIT bounds the number of polls instead (tests: 1).
        static const int spin_limit = [] { const char* e = getenv("CNF_SOLVE_POLL_LIMIT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0x7fffffff; }();
        static const int wait_us = [] { const char* e = getenv("CNF_SOLVE_WAIT_US"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 2000; }();
        sv.spin_limit = h->poll_limit > 0 ? h->poll_limit : spin_limit;
        const int wus = h->wait_us > 0 ? h->wait_us : wait_us;
        sv.wait_ticks = wus < 20000000 ? 100u * (unsigned)wus : 2000000000u;   // (per tile: the launcher scales it)
        sv.trace = h->step_trace; sv.trace_cap = h->step_trace_cap;
        // (k_trace3s<SOLVE> works on the integrator's buffers: u0 is assembled / copied into U[0] in front of it, the
        // post-processing follows it -- three launches per inference)
        if (tsolve_ok) {
            if (post && post->xs) launch_build_u0(post->xs, h->U[0], h->nd.nvars, D, B, st);
            else if (u0 != h->U[0]) HIPCHK(h, hipMemcpyAsync(h->U[0], u0, n * sizeof(float), hipMemcpyDeviceToDevice, st));
            ++launches;
        }
        const bool fused_io = post && post->xs && !tsolve_ok;   // inference: u0 from the data columns and the post-processing in the launch
        if (fused_io) { sv.xs = post->xs; sv.logpx = post->logpx; sv.regs = post->regs; sv.sums5 = post->sums5; if (rec) sv.u_out = u_out; }
        else { sv.u0 = u0; sv.u_out = tsolve_ok ? nullptr : u_out; }   // (the launcher reads u0 in place or copies it into U[0])
        // gradient path: every attempt files u_n and its stage states in the slot of step `naccept` (as the streamed
        // recording does); the store is sized beforehand and the solve repeated if it took more steps than fit
        float* dump = nullptr; size_t slot = 0; int dcap = 0;
        if (rec && !rec->wg) {
            if ((s = traj_reserve(h, 64)) != CNF_OK) return s;
            slot = traj_slot_floats(h); dcap = h->traj_cap;
            dump = h->traj + n;                      // stage area of slot 0; u_n sits one array before
        }
        s = CNF_ERR_UNSUPPORTED;
        if (wave_ok)
            s = wave_solve_launch(h->nd_wave, train != 0, h->d_params, h->nd.n_cond > 0 ? h->d_cond : nullptr, h->cbs, h->d_state, h->U[0],
                                  eps, B, st, h->d_mirror + mslot, base, sv, rec ? rec->wg : nullptr);
        // (the gradient in the launch of the solve exists on the wave kernel only: if that launch could not start, the caller
        // runs the streamed gradient path)
        if (rec && rec->wg && s != CNF_OK) { rec->wg_failed = true; return CNF_OK; }
        if (tsolve_ok) {
            const GradLayout g = grad_layout(h->nd);
            const AdjMfmaLayout m = adj_mfma_layout(h->nd, g);
            TraceArgs ta{};
            ta.B = B; ta.U[0] = h->U[0]; ta.U[1] = h->U[1]; ta.K1[0] = h->K1[0]; ta.K1[1] = h->K1[1];
            for (int i = 0; i < 5; ++i) ta.Ks[i] = h->Ks[i];
            ta.norm_kind = -1; ta.st_mut = h->d_state; ta.n_total = (float)n;
            ta.mirror = h->d_mirror + mslot; ta.seq = base;
            s = launch_trace_solve(h->nd, g, m, h->d_adj_img, ta, sv, st) == hipSuccess ? CNF_OK : CNF_ERR_UNSUPPORTED;
            if (s != CNF_OK) (void)hipGetLastError();
        }
        if (bcast_ok) {
            BcastRecord brec{dump, n, slot, dcap, h->traj_hs};
            Solve3Args svb = sv;
            if (rec && u_out) svb.u_out = u_out;         // (this kernel writes the final state where it is wanted)
            s = bcast_solve_launch(h->nd, train != 0, h->d_params, h->d_bimg, h->d_state, h->U[0], eps, B, st, h->d_mirror + mslot, base, svb, h->device,
                                   h->d_bstore, h->nd.n_cond > 0 ? h->d_cond : nullptr, h->cbs, rec ? &brec : nullptr);
            if (s == CNF_OK && rec && u_out) sv.u_out = u_out;
        }
        if (s == CNF_ERR_UNSUPPORTED && use_mfma)
            s = mfma_solve_persistent(h->mfma, h->nd, train, h->d_state, h->U, eps, B, st, h->d_mirror + mslot, base, sv, h->device,
                                      dump, n, slot, dcap, h->traj_hs, h->K1);
        if (s == CNF_OK) {
            ++launches;
            h->mirror_base = base + 1;
            ++h->one_launch_count;
            h->last_state = h->d_state;
            HIPCHK(h, hipGetLastError());
            if (h->submitting && rec && rec->wg) {
                // a submitted gradient: the launch is on its way; cnf_loss_grad_collect reads its outcome from its mirror slot
                cnf_ctx::Submitted sub;
                sub.launched = true; sub.grad = true; sub.seq = base; sub.slot = mslot; sub.hairer = hairer; sub.mode = mode; sub.B = B;
                sub.k = CNF_KERNEL_MFMA; sub.opts = *opts; sub.st = st;
                h->submitted.push_back(sub);
                h->sub_taken = true;
                ++g_submitted_inflight; g_submitted_stream = st;
                if (post) post->launched = true;
                rec->wg_submitted = true;
                return CNF_OK;
            }
            if (h->submitting && fused_io && !rec && !u_out) {
                // submitted: the launch is on its way; cnf_inference_collect reads its outcome from its mirror slot
                cnf_ctx::Submitted sub;
                sub.launched = true; sub.seq = base; sub.slot = mslot; sub.hairer = hairer; sub.mode = mode; sub.B = B; sub.k = k;
                sub.xs = post->xs; sub.eps = eps; sub.logpx = post->logpx; sub.regs = post->regs; sub.sums5 = post->sums5;
                sub.opts = *opts; sub.st = st;
                h->submitted.push_back(sub);
                h->sub_taken = true;
                ++g_submitted_inflight; g_submitted_stream = st;
                post->launched = true;
                return CNF_OK;
            }
            StepState fin{};
            bool aborted = false;
            if ((s = finish_one_launch(h, base, mslot, st, &fin, &aborted)) != CNF_OK) return s;
            const int attempts = fin.naccept + fin.nreject;
            if (!aborted) {
                if (fused_io) post->launched = true;
                else if (post) { enqueue_post(h, train, h->d_state, *post, B, false, st); ++launches; post->launched = true; }
                if (u_out && !sv.u_out) { launch_copy_final(h->d_state, h->U[0], h->U[1], u_out, n, st); ++launches; }
                HIPCHK(h, hipGetLastError());
                if (rec) {                               // step sizes back from the device
                    rec->n = fin.naccept;
                    rec->overflow = !rec->wg && fin.naccept > h->traj_cap;
                    rec->hs.assign((size_t)(rec->overflow ? 0 : fin.naccept), 0.f);
                    if (!rec->overflow && fin.naccept > 0)
                        HIPCHK(h, hipMemcpyAsync(rec->hs.data(), rec->wg ? rec->wg->hs_out : h->traj_hs, (size_t)fin.naccept * sizeof(float),
                                                 hipMemcpyDeviceToHost, st));
                    if (rec->wg) rec->wg_done = true;
                    else final_sync = true;
                }
                if (final_sync) HIPCHK(h, hipStreamSynchronize(st));
                if (stats) {
                    stats->nf = (hairer ? 2 : 1) + 6 * attempts;
                    stats->naccept = fin.naccept;
                    stats->nreject = fin.nreject;
                    stats->t_final = fin.t;
                    stats->dt_last = fin.dt;
                    stats->kernel_used = wave_ok ? CNF_KERNEL_MFMA : k;
                    stats->launches = launches;
                }
                if (fin.nonfinite) {
                    // (the step sizes above may still be on their way into rec->hs, which the caller drops on this return)
                    if (rec && rec->wg) HIPCHK(h, hipStreamSynchronize(st));
                    return fail(h, CNF_ERR_NONFINITE, "solver state became NaN/Inf");
                }
                return CNF_OK;
            }
            // A workgroup did not arrive within the wait bound: something else holds CUs (another stream or process, a CU
            // mask).  The launch has ended (every wait is bounded); nothing of its result is used.  The solve runs again from
            // u0 on the streamed driver below, which needs no co-residency.
            ++h->fallbacks;
            if (rec && rec->wg) { rec->wg_failed = true; return CNF_OK; }      // (the caller runs the streamed gradient path)
            // (u0 == h->U[0] was solved in place; it can only be rebuilt when the caller's data columns are still at hand --
            // `inference_impl` passes them in post->xs, the streamed driver below assembles u0 from them again)
            if (!fused_io && u0 == h->U[0] && !(post && post->xs))
                return fail(h, CNF_ERR_HIP, "one-launch solve: a workgroup did not arrive and u0 was solved in place (set CNF_PERSISTENT=0)");
        } else if (s != CNF_ERR_UNSUPPORTED) return fail(h, s, "one-launch solve failed to start");
        persist_lock.unlock();
    }
    if (post && post->xs) launch_build_u0(post->xs, h->U[0], h->nd.nvars, D, B, st, h->d_state, init);   // (u0 == h->U[0])
    else launch_set_state(h->d_state, *init, st);
    if (u0 != h->U[0]) HIPCHK(h, hipMemcpyAsync(h->U[0], u0, n * sizeof(float), hipMemcpyDeviceToDevice, st));

    // k1 = f(u0).  With the automatic initial dt on the fused path, the two norms and their controller phases
    // ride in the RHS launches themselves (the last workgroup to finish runs the phase): 2 launches, not 4.
    unsigned* ticket = reinterpret_cast<unsigned*>(h->d_sums + 8);
    bool fused_init = false;
    if (use_mfma && hairer && !lockstep) {
        s = mfma_rhs_init0(h->mfma, h->nd, train, h->d_state, h->U[0], eps, h->K1[0], h->partials, ticket, B, st);
        if (s == CNF_OK) fused_init = true;
        else if (s != CNF_ERR_UNSUPPORTED) return fail(h, s, "MFMA RHS launch failed");
    }
    if (fused_init) {
    } else if (use_mfma) {
        s = mfma_rhs(h->mfma, h->nd, train, h->U[0], eps, h->K1[0], B, st);
        if (s != CNF_OK) return fail(h, s, "MFMA RHS launch failed");
    } else if (h->trace_on && trace_fused && hairer) {       // k1 = f(u0) with the first norm of the automatic initial dt
        TraceFuse f; f.norm_kind = 0;
        if ((s = launch_trace(h, h->U[0], h->K1[0], true, false, B, st, 0, nullptr, false, &f)) != CNF_OK) return s;
        fused_init = true;
    } else if (h->trace_on) {
        if ((s = launch_trace(h, h->U[0], h->K1[0], false, false, B, st)) != CNF_OK) return s;
    } else {
        RhsArgs a{};
        a.B = B; a.S = h->cap_B; a.train = train; a.ws = h->ws; a.eps = eps;
        a.cond = h->mfma.cond; a.cbs = h->cbs;
        a.u = h->U[0]; a.du = h->K1[0];
        launch_rhs_generic(h->nd, h->d_params, a, st);
    }
    launches += 1;
    int nf = 1;

    NormArgs na{};
    na.st = h->d_state; na.n = n; na.partials = h->partials;
    for (int i = 0; i < 2; ++i) { na.U[i] = h->U[i]; na.K1[i] = h->K1[i]; }
    for (int i = 0; i < 5; ++i) na.Ks[i] = h->Ks[i];

    if (hairer && fused_init && h->trace_on) {
        // f1 = f(u0 + h*f0) -> Ks[0] with the second norm and the controller phase that sets dt, in the trace launch
        const float one = 1.f;
        TraceFuse f; f.norm_kind = 1;
        if ((s = launch_trace(h, nullptr, h->Ks[0], true, false, B, st, 1, &one, false, &f)) != CNF_OK) return s;
        launches += 1;
        nf += 1;
    } else if (hairer && fused_init) {
        // f1 = f(u0 + h*f0) -> Ks[0], with the second norm and the controller phase that sets dt
        s = mfma_rhs_stage(h->mfma, h->nd, train, h->d_state, h->U, h->K1, h->Ks, eps, 1, B, st, h->d_state,
                           h->partials, ticket);
        if (s != CNF_OK) return fail(h, s, "MFMA RHS launch failed");
        launches += 1;
        nf += 1;
    } else if (hairer) {
        // automatic initial dt (Hairer; OrdinaryDiffEq's ode_determine_initdt, third party)
        na.kind = 0;
        if (lockstep) {
            launch_norm_partials(na, nblk, st);
            if ((s = lockstep_controller(h, h->d_state, h->partials, 0, (float)n, st)) != CNF_OK) return s;
        } else {        // norm + controller in one launch (the last block to finish runs the controller)
            na.ticket = reinterpret_cast<unsigned*>(h->d_sums + 8); na.st_mut = h->d_state; na.ctrl_phase = 0; na.n_total = (float)n;
            launch_norm_partials(na, nblk, st);
            na.ticket = nullptr;
        }
        // f1 = f(u0 + h*f0) -> Ks[0]
        if (use_mfma) {
            s = mfma_rhs_stage(h->mfma, h->nd, train, h->d_state, h->U, h->K1, h->Ks, eps, 1, B, st);
            if (s != CNF_OK) return fail(h, s, "MFMA RHS launch failed");
        } else if (h->trace_on) {
            const float one = 1.f;
            if ((s = launch_trace(h, nullptr, h->Ks[0], true, false, B, st, 1, &one)) != CNF_OK) return s;
        } else {
            RhsArgs a{};
            a.st = h->d_state; a.B = B; a.S = h->cap_B; a.train = train; a.ws = h->ws; a.eps = eps;
            a.cond = h->mfma.cond; a.cbs = h->cbs;
            for (int i = 0; i < 2; ++i) { a.U[i] = h->U[i]; a.K1[i] = h->K1[i]; }
            for (int i = 0; i < 5; ++i) a.Ks[i] = h->Ks[i];
            a.nk = 1; a.coef[0] = 1.f; a.du = h->Ks[0];
            launch_rhs_generic(h->nd, h->d_params, a, st);
        }
        na.kind = 1;
        if (lockstep) {
            launch_norm_partials(na, nblk, st);
            if ((s = lockstep_controller(h, h->d_state, h->partials, 1, (float)n, st)) != CNF_OK) return s;
        } else {
            na.ticket = reinterpret_cast<unsigned*>(h->d_sums + 8); na.st_mut = h->d_state; na.ctrl_phase = 1; na.n_total = (float)n;
            launch_norm_partials(na, nblk, st);
            na.ticket = nullptr;
        }
        launches += lockstep ? 5 : 3;
        nf += 1;
    }
    HIPCHK(h, hipGetLastError());

    StepState* cur_state = h->d_state;   // slot holding the live integrator state
    int pp = 0;                          // partials buffer the NEXT launch reads
    if (lockstep || (rec && !use_mfma)) {
        // one attempt at a time.  Lock-step: every shard must see the same global error norm before
        // the next attempt is sized.  Recording (gradient path): the host files the state after
        // every accepted step.
        StepState* snap = &h->h_state[0];
        float h_attempt = 0.f;
        int seen_accept = 0;
        if (rec) {
            float* slot0;
            if ((s = traj_slot(h, 0, &slot0)) != CNF_OK) return s;
            HIPCHK(h, hipMemcpyAsync(slot0, h->U[0], n * sizeof(float), hipMemcpyDeviceToDevice, st));
            HIPCHK(h, hipMemcpyAsync(snap, h->d_state, sizeof(StepState), hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipStreamSynchronize(st));
            h_attempt = snap->h;
            rec->hs.clear(); rec->n = 0;
        }
        // Lock-step over an RCCL communicator: nothing of an attempt leaves the stream (step kernel, the reduction of the
        // partials, the all-reduce of three floats, the controller), so FOUR attempts are enqueued per look at the state --
        // every shard holds the same state bit for bit, sees `done` in the same attempt and enqueues the same number of
        // collectives; attempts queued past the end find `done` and change nothing.
        const int chunk = (lockstep && h->shard_comm && use_mfma && !rec) ? 4 : 1;
        for (long it = 0;; it += chunk) {
            if (it >= (long)opts->maxiters) return fail(h, CNF_ERR_MAXITERS, "maxiters reached before t1");
          for (int c = 0; c < chunk && it + c < (long)opts->maxiters; ++c) {
            float* dump = nullptr;
            if (rec) {                                   // stage states of the attempt from u_{seen_accept}
                float* slot;
                if ((s = traj_slot(h, seen_accept, &slot)) != CNF_OK) return s;
                dump = slot + n;
            }
            if (use_mfma) {
                s = mfma_step(h->mfma, h->nd, train, h->d_state, h->d_state + 1, h->U, h->K1, h->Ks, eps,
                              h->partials, h->partials, false, false, B, st, dump, n);
                if (s != CNF_OK) return fail(h, s, "MFMA step launch failed");
                launches += 1;
            } else {
                launches += enqueue_attempt_generic(h, train, eps, B, nblk, st, false, dump, n);
            }
            if (lockstep) {
                if ((s = lockstep_controller(h, h->d_state, h->partials, 2, (float)n, st)) != CNF_OK) return s;
                launches += 2;
            } else {
                launch_controller(h->d_state, h->partials, 2, (float)n, st);
                launches += 1;
            }
          }
            HIPCHK(h, hipMemcpyAsync(snap, h->d_state, sizeof(StepState), hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipStreamSynchronize(st));
            if (rec && snap->naccept > seen_accept) {
                seen_accept = snap->naccept;
                float* slot;
                if ((s = traj_slot(h, seen_accept, &slot)) != CNF_OK) return s;
                HIPCHK(h, hipMemcpyAsync(slot, h->U[snap->cur], n * sizeof(float), hipMemcpyDeviceToDevice, st));
                rec->hs.push_back(h_attempt);
                rec->n = seen_accept;
            }
            h_attempt = snap->h;
            if (snap->done) break;
        }
        h->last_state = h->d_state;
        if (u_out) {
            launch_copy_final(h->d_state, h->U[0], h->U[1], u_out, n, st);
            launches += 1;
        }
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(st));
        if (stats) {
            stats->nf = nf + 6 * (snap->naccept + snap->nreject);
            stats->naccept = snap->naccept;
            stats->nreject = snap->nreject;
            stats->t_final = snap->t;
            stats->dt_last = snap->dt;
            stats->kernel_used = k;
            stats->launches = launches;
        }
        if (snap->nonfinite) return fail(h, CNF_ERR_NONFINITE, "solver state became NaN/Inf");
        return CNF_OK;
    }
    // Streamed solve: launches are kept a few attempts ahead of the last state the host has seen.  The controller of an
    // attempt runs on the device -- inside the NEXT launch of the fused step kernel, or in the last block of the error-norm
    // kernel on the per-stage paths -- and whoever ran it mirrors the new state to pinned host memory under the launch
    // index; the host polls that index: no events, no copies, no stand-alone controller, no chunk boundaries.
    // Launches queued past the end find `done` and exit at once.
    bool done = false;
    StepState fin{};
    long post_at = -1, seen_done = -1;  // launches enqueued when the post-processing was last enqueued; the launch that published `done`
    {
        const int AHEAD = 3;
        const volatile cnf_ctx::HostMirror* hm = h->h_mirror;
        const unsigned base = h->mirror_base;
        long sent = 0, seen = 0;       // launches (fused) / attempts (per-stage) enqueued; newest mirror index read
        const long max_sent = (long)opts->maxiters + (use_mfma ? 1 : 0);   // fused: one more launch runs the last controller
        // gradient path: every attempt files u_n and its stage states in the slot of step `naccept`, indexed on
        // the device; the store is sized beforehand and the solve repeated if it took more steps than fit
        float* dump = nullptr; size_t slot = 0; int dcap = 0;
        if (rec) {
            if ((s = traj_reserve(h, 64)) != CNF_OK) return s;
            slot = traj_slot_floats(h); dcap = h->traj_cap;
            dump = h->traj + n;                      // stage area of slot 0; u_n sits one array before
        }
        // attempts the newest snapshot says are still needed, (t1 - t)/dt rounded up (the controller rarely shrinks dt
        // near the end); -1 = no snapshot yet.  Launches beyond that would only find `done` and exit.
        long need = -1;
        for (;;) {
            while (sent < max_sent && sent - seen < AHEAD && !done &&
                   (need < 0 || sent - seen < need + (use_mfma ? 1 : 0))) {      // fused: + the launch that runs the last controller
                if (use_mfma) {
                    // launch i applies the controller of attempt i-1 and publishes under index i
                    const bool apply = sent > 0;
                    StepState* st_next = cur_state == h->d_state ? h->d_state + 1 : h->d_state;
                    s = mfma_step(h->mfma, h->nd, train, cur_state, st_next, h->U, h->K1, h->Ks, eps,
                                  h->partials + 2 * MAX_PARTIALS * pp, h->partials + 2 * MAX_PARTIALS * (pp ^ 1), apply,
                                  false, B, st, dump, n, h->d_mirror, base + (unsigned)sent, slot, dcap, h->traj_hs);
                    if (s != CNF_OK) return fail(h, s, "MFMA step launch failed");
                    if (apply) cur_state = st_next;
                    pp ^= 1;
                    ++launches;
                } else {
                    // attempt i ends with its own controller and publishes under index i + 1
                    launches += enqueue_attempt_generic(h, train, eps, B, nblk, st, true, nullptr, 0, h->d_mirror,
                                                        base + (unsigned)sent + 1);
                }
                ++sent;
            }
            h->mirror_base = base + (unsigned)sent + 1;
            // everything the estimate asks for is in the queue: the post-processing goes right behind it (fused path: the
            // state slot the newest launch writes is the one a `done` would be published in)
            if (post && !rec && use_mfma && !done && need >= 0 && sent - seen >= need + 1 && post_at != sent) {
                enqueue_post(h, train, cur_state, *post, B, true, st);
                post_at = sent;
                ++launches;
            }
            if (done) break;
            if (seen >= (long)opts->maxiters) {
                HIPCHK(h, hipStreamSynchronize(st));
                return fail(h, CNF_ERR_MAXITERS, "maxiters reached before t1");
            }
            // wait for a consistent snapshot newer than the last one read.  The spin backs off: pause first, yield the
            // core once the wait outlasts a few launches.
            unsigned sq = 0;
            StepState snap;
            for (long spins = 0;; ++spins) {
                if (cnf_mirror_read(hm, &snap, &sq) && (int)(sq - base) > (int)seen) break;
                if (spins < 4096) _mm_pause();
                else sched_yield();
                if (spins % 100000 == 99999) {                     // a faulted kernel would never publish
                    hipError_t qe = hipStreamQuery(st);
                    if (qe != hipSuccess && qe != hipErrorNotReady) HIPCHK(h, qe);
                    if (qe == hipSuccess && !(cnf_mirror_read(hm, &snap, &sq) && (int)(sq - base) > (int)seen))
                        return fail(h, CNF_ERR_HIP, "step kernels finished without publishing a state");
                    if (qe == hipSuccess) break;
                }
            }
            seen = (long)(sq - base);
            if (snap.done) { fin = snap; done = true; seen_done = seen; }
            need = 1;
            if (snap.dt > 0.f) {
                const double left = std::fabs((double)snap.t1 - (double)snap.t) / (double)snap.dt;
                need = left > 1e6 ? 1000000 : (long)std::ceil(left - 1e-6);
                if (need < 1) need = 1;
            }
        }
    }
    h->last_state = cur_state;
    if (post) {
        // the speculative launch counts if it was enqueued behind the launch that published `done` (mirror index
        // `seen`, i.e. launch number seen on the fused path: the launches are numbered from 0)
        if (!(post_at > seen_done)) { enqueue_post(h, train, cur_state, *post, B, false, st); ++launches; }
        post->launched = true;
    }
    if (rec) {            // streamed recording: step sizes back from the device
        rec->n = fin.naccept;
        rec->overflow = fin.naccept > h->traj_cap;
        rec->hs.assign((size_t)(rec->overflow ? 0 : fin.naccept), 0.f);
        if (!rec->overflow && fin.naccept > 0)
            HIPCHK(h, hipMemcpyAsync(rec->hs.data(), h->traj_hs, (size_t)fin.naccept * sizeof(float),
                                     hipMemcpyDeviceToHost, st));
        final_sync = true;
    }
    if (u_out) {
        launch_copy_final(cur_state, h->U[0], h->U[1], u_out, n, st);
        launches += 1;
    }
    HIPCHK(h, hipGetLastError());
    // the step count is already known from the last mirror; the copy is stream-ordered work.  Callers that
    // hand u_out to the host wait here, cnf_inference goes straight on to the post-processing kernel.
    if (final_sync) HIPCHK(h, hipStreamSynchronize(st));
    nf += 6 * (fin.naccept + fin.nreject);
    if (stats) {
        stats->nf = nf;
        stats->naccept = fin.naccept;
        stats->nreject = fin.nreject;
        stats->t_final = fin.t;
        stats->dt_last = fin.dt;
        stats->kernel_used = k;
        stats->launches = launches;
    }
    if (fin.nonfinite) return fail(h, CNF_ERR_NONFINITE, "solver state became NaN/Inf");
    return CNF_OK;
}

extern "C" cnf_status cnf_solve_tsit5_host(cnf_handle h, int mode, const float* u0, const float* eps,
                                           float* u_out, int B, const cnf_solve_opts* opts,
                                           cnf_solve_stats* stats) {
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!u0 || !u_out || !opts) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (B == 0) { if (stats) memset(stats, 0, sizeof *stats); return CNF_OK; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t D = rows_of(h, mode), n_in = h->nd.n_in;
    if ((s = ensure_stage(h, (2 * D + n_in) * B)) != CNF_OK) return s;
    float* u_d = h->stage;
    float* o_d = u_d + D * B;
    float* e_d = eps ? o_d + D * B : nullptr;
    HIPCHK(h, hipMemcpy(u_d, u0, D * B * sizeof(float), hipMemcpyHostToDevice));
    if (eps) HIPCHK(h, hipMemcpy(e_d, eps, n_in * B * sizeof(float), hipMemcpyHostToDevice));
    s = cnf_solve_tsit5(h, mode, u_d, e_d, o_d, B, opts, stats, nullptr);
    if (s == CNF_OK) {
        hipError_t e = hipMemcpy(u_out, o_d, D * B * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) s = fail(h, CNF_ERR_HIP, hipGetErrorString(e));
    }
    return s;
}

// ---------------------------------------------------------------------------------------
// assembly / post-processing / loss (a6, a8, a9)
// ---------------------------------------------------------------------------------------
extern "C" cnf_status cnf_build_u0(cnf_handle h, int mode, const float* xs, float* u0, int B,
                                   void* stream) {
    if (!h || !xs || !u0) return CNF_ERR_BAD_ARG;
    if (mode != CNF_MODE_TEST && mode != CNF_MODE_TRAIN) return fail(h, CNF_ERR_BAD_ARG, "unknown mode");
    if (B < 0) return fail(h, CNF_ERR_BAD_SHAPE, "negative batch");
    if (B == 0) return CNF_OK;
    HIPCHK(h, hipSetDevice(h->device));
    launch_build_u0(xs, u0, h->nd.nvars, rows_of(h, mode), B, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CNF_OK;
}

extern "C" cnf_status cnf_inference_post(cnf_handle h, int mode, const float* u_final,
                                         float* logpx, float* regs, int B, void* stream) {
    if (!h || !u_final || !logpx || !regs) return CNF_ERR_BAD_ARG;
    if (mode != CNF_MODE_TEST && mode != CNF_MODE_TRAIN) return fail(h, CNF_ERR_BAD_ARG, "unknown mode");
    if (B < 0) return fail(h, CNF_ERR_BAD_SHAPE, "negative batch");
    if (B == 0) return CNF_OK;
    HIPCHK(h, hipSetDevice(h->device));
    launch_post(h->nd, mode == CNF_MODE_TRAIN, u_final, logpx, regs, B, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CNF_OK;
}

static cnf_status inference_impl(cnf_handle h, int mode, const float* xs, const float* eps,
                                 float* logpx, float* regs, float* u_final, float* sums5, int B,
                                 const cnf_solve_opts* opts, cnf_solve_stats* stats, void* stream) {
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!xs || !logpx || !regs || !opts) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (stats) memset(stats, 0, sizeof *stats);
    if (B == 0) {
        if (sums5) HIPCHK(h, hipMemsetAsync(sums5, 0, 5 * sizeof(float), (hipStream_t)stream));
        return CNF_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    if ((s = ensure_capacity(h, B)) != CNF_OK) return s;
    // no allocations and no copies on this path: u0 is assembled in the integrator's own state buffer, and the
    // post-processing reads the final state from wherever the integrator left it
    PostHook ph{logpx, regs, sums5, xs};
    if (s == CNF_OK) s = solve_core(h, mode, h->U[0], eps, u_final, B, opts, stats, stream, nullptr, false, &ph);
    if (s == CNF_OK && !ph.launched) {                      // (the one-attempt-at-a-time drivers leave it to the caller)
        enqueue_post(h, mode == CNF_MODE_TRAIN, h->last_state, ph, B, false, (hipStream_t)stream);   // stream-ordered
        HIPCHK(h, hipGetLastError());
    }
    return s;
}

// ---- submitted inferences: enqueue now, collect later (the GPU goes from one solve straight into the next) ----
// The end of one submitted launch: wait for its outcome; a launch that ran out of a wait is run again on the streamed driver.
static cnf_status finish_submission(cnf_handle h, const cnf_ctx::Submitted& sub, cnf_solve_stats* stats) {
    HIPCHK(h, hipSetDevice(h->device));
    StepState fin{};
    bool aborted = false;
    cnf_status s;
    {
        std::unique_lock<std::mutex> lock(g_persist_mu);
        s = finish_one_launch(h, sub.seq, sub.slot, sub.st, &fin, &aborted);
        --g_submitted_inflight;
    }
    if (s != CNF_OK) return s;
    if (!aborted) {
        if (stats) {
            const int attempts = fin.naccept + fin.nreject;
            stats->nf = (sub.hairer ? 2 : 1) + 6 * attempts;
            stats->naccept = fin.naccept; stats->nreject = fin.nreject;
            stats->t_final = fin.t; stats->dt_last = fin.dt;
            stats->kernel_used = sub.k; stats->launches = 1;
        }
        if (fin.nonfinite) return fail(h, CNF_ERR_NONFINITE, "solver state became NaN/Inf");
        return CNF_OK;
    }
    ++h->fallbacks;
    if (sub.grad)         // (its gradient was written as zeros and its loss as NaN by k_grad_finish: the caller runs the batch again)
        return fail(h, CNF_ERR_UNSUPPORTED, "the submitted gradient launch gave up (a wait ran out, or more steps than its store holds): "
                                            "zeros were delivered; run the batch again with cnf_loss_grad");
    const bool was = h->collecting;
    h->collecting = true; h->no_persist = true;
    s = inference_impl(h, sub.mode, sub.xs, sub.eps, sub.logpx, sub.regs, nullptr, sub.sums5, sub.B, &sub.opts, stats, sub.st);
    h->no_persist = false; h->collecting = was;
    return s;
}

// The oldest submitted inference, completed and taken off the queue.
static cnf_status collect_one(cnf_handle h, cnf_solve_stats* stats) {
    cnf_ctx::Submitted sub = h->submitted.front();
    h->submitted.pop_front();
    if (stats) *stats = sub.stats;
    if (!sub.launched) return sub.status;
    return finish_submission(h, sub, stats);
}

static cnf_status settle_submitted(cnf_handle h) {
    const bool was = h->collecting;
    h->collecting = true;                     // (the fallback's own calls must not drain the queue they are part of)
    for (auto& sub : h->submitted) {
        if (!sub.launched) continue;
        cnf_solve_stats st{};
        sub.status = finish_submission(h, sub, &st);
        sub.stats = st;
        sub.launched = false;
    }
    h->collecting = was;
    return CNF_OK;                            // (each outcome, errors included, is its collect call's)
}

// a handle destroyed with submissions outstanding gives up its claim on the process-wide launch order
static void release_submitted(cnf_handle h) {
    std::unique_lock<std::mutex> lock(g_persist_mu);
    for (const auto& sub : h->submitted)
        if (sub.launched) --g_submitted_inflight;
    h->submitted.clear();
}

extern "C" cnf_status cnf_inference_submit(cnf_handle h, int mode, const float* xs, const float* eps, float* logpx,
                                           float* regs, float* sums5, int B, const cnf_solve_opts* opts, void* stream) {
    if (!h) return CNF_ERR_BAD_ARG;
    if (h->submitted.size() >= 3) return fail(h, CNF_ERR_BAD_ARG, "three inferences are submitted already: collect one first");
    h->collecting = true; h->submitting = true; h->sub_taken = false;
    cnf_solve_stats st{};
    const cnf_status s = inference_impl(h, mode, xs, eps, logpx, regs, nullptr, sums5, B, opts, &st, stream);
    h->submitting = false; h->collecting = false;
    if (!h->sub_taken) {                      // completed on the spot (another driver, an empty batch, or an error)
        cnf_ctx::Submitted sub;
        sub.status = s; sub.stats = st;
        h->submitted.push_back(sub);
    }
    return CNF_OK;                            // (the outcome, errors included, is the matching collect call's)
}

extern "C" cnf_status cnf_inference_collect(cnf_handle h, cnf_solve_stats* stats) {
    if (!h) return CNF_ERR_BAD_ARG;
    if (stats) memset(stats, 0, sizeof *stats);
    if (h->submitted.empty()) return fail(h, CNF_ERR_BAD_ARG, "no inference is submitted");
    h->collecting = true;
    const cnf_status s = collect_one(h, stats);
    h->collecting = false;
    return s;
}

extern "C" int cnf_inference_pending(cnf_handle h) { return h ? (int)h->submitted.size() : -1; }

extern "C" cnf_status cnf_inference(cnf_handle h, int mode, const float* xs, const float* eps,
                                    float* logpx, float* regs, float* u_final, int B,
                                    const cnf_solve_opts* opts, cnf_solve_stats* stats,
                                    void* stream) {
    return inference_impl(h, mode, xs, eps, logpx, regs, u_final, nullptr, B, opts, stats, stream);
}

extern "C" cnf_status cnf_inference_sums(cnf_handle h, int mode, const float* xs, const float* eps, float* logpx,
                                         float* regs, float* sums5, int B, const cnf_solve_opts* opts,
                                         cnf_solve_stats* stats, void* stream) {
    if (h && !sums5) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    // the sums ride in the post-processing launch (block partials, the last block adds them in block order)
    return inference_impl(h, mode, xs, eps, logpx, regs, nullptr, sums5, B, opts, stats, stream);
}

extern "C" cnf_status cnf_inference_host(cnf_handle h, int mode, const float* xs,
                                         const float* eps, float* logpx, float* regs,
                                         float* u_final, int B, const cnf_solve_opts* opts,
                                         cnf_solve_stats* stats) {
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!xs || !logpx || !regs || !opts) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (B == 0) { if (stats) memset(stats, 0, sizeof *stats); return CNF_OK; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t D = rows_of(h, mode), n_in = h->nd.n_in, nv = h->nd.nvars;
    const size_t out_f = (size_t)B * (4 + D);
    if ((s = ensure_stage(h, (nv + n_in) * B + out_f)) != CNF_OK) return s;
    float* xs_d = h->stage;
    float* out_d = xs_d + nv * B;
    float* e_d = eps ? out_d + out_f : nullptr;
    HIPCHK(h, hipMemcpy(xs_d, xs, nv * B * sizeof(float), hipMemcpyHostToDevice));
    if (eps) HIPCHK(h, hipMemcpy(e_d, eps, n_in * B * sizeof(float), hipMemcpyHostToDevice));
    float* lp = out_d; float* rg = out_d + B; float* uf = out_d + 4 * (size_t)B;
    s = cnf_inference(h, mode, xs_d, e_d, lp, rg, uf, B, opts, stats, nullptr);
    if (s == CNF_OK) {
        hipError_t e = hipMemcpy(logpx, lp, (size_t)B * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(regs, rg, 3 * (size_t)B * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess && u_final) e = hipMemcpy(u_final, uf, D * B * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) s = fail(h, CNF_ERR_HIP, hipGetErrorString(e));
    }
    return s;
}

extern "C" cnf_status cnf_loss_sums(cnf_handle h, const float* logpx, const float* regs, int B,
                                    float* sums5, void* stream) {
    if (!h || !logpx || !regs || !sums5) return CNF_ERR_BAD_ARG;
    if (B < 0) return fail(h, CNF_ERR_BAD_SHAPE, "negative batch");
    HIPCHK(h, hipSetDevice(h->device));
    launch_loss_sums(logpx, regs, B, sums5, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return CNF_OK;
}

extern "C" cnf_status cnf_loss_allreduce(cnf_handle h, cnf_comm comm, float* sums5, void* stream) {
    if (!h || !comm || !sums5) return CNF_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (cnf_comm_allreduce(comm, sums5, 5, stream) != CNF_OK) return fail(h, CNF_ERR_RCCL, cnf_comm_last_error());
    return CNF_OK;
}

extern "C" cnf_status cnf_set_shard_comm(cnf_handle h, cnf_comm comm) {
    if (!h) return CNF_ERR_BAD_ARG;
    h->shard_comm = comm;
    return CNF_OK;
}

extern "C" cnf_status cnf_loss_from_sums(cnf_handle h, int mode, const float* sums5, float* loss) {
    if (!h || !sums5 || !loss) return CNF_ERR_BAD_ARG;
    const double cnt = sums5[4];
    if (!(cnt > 0)) return fail(h, CNF_ERR_BAD_SHAPE, "empty batch");
    if (mode == CNF_MODE_TRAIN)      // src/icnf.jl:489
        *loss = (float)((-(double)sums5[0] + (double)h->lam[0] * sums5[1] + (double)h->lam[1] * sums5[2] +
                         (double)h->lam[2] * sums5[3]) / cnt);
    else                             // src/base_icnf.jl:496
        *loss = (float)(-(double)sums5[0] / cnt);
    return CNF_OK;
}

// ---------------------------------------------------------------------------------------
// Gradient of the TrainMode loss w.r.t. the parameters.
// Reference: MLJModelInterface.fit differentiates loss(icnf, TrainMode(), xs, ps, st) with Enzyme
// through the solve (src/exts/mlj_ext/core_icnf.jl:59-73, src/icnf.jl:481-490).
// ---------------------------------------------------------------------------------------
static cnf_status ensure_grad_capacity(cnf_handle h, int B) {
    HIPCHK(h, hipSetDevice(h->device));
    const GradLayout g = grad_layout(h->nd);
    if (!h->d_PT) HIPCHK(h, hipMalloc(&h->d_PT, h->n_params * sizeof(float)));
    if (!h->d_adj_img) {
        const AdjMfmaLayout m = adj_mfma_layout(h->nd, g);
        HIPCHK(h, hipMalloc(&h->d_adj_img, (size_t)m.img_floats * sizeof(float)));
    }
    if ((size_t)B <= h->grad_cap_B) return CNF_OK;
    HIPCHK(h, hipDeviceSynchronize());
    if (h->grad_arena) { (void)hipFree(h->grad_arena); h->grad_arena = nullptr; }
    if (h->traj) { (void)hipFree(h->traj); (void)hipFree(h->traj_hs); h->traj = nullptr; h->traj_hs = nullptr; h->traj_cap = 0; }
    h->grad_cap_B = 0;
    const size_t cap = ((size_t)B + 63) & ~(size_t)63;
    const size_t D = (size_t)h->nd.n_in + 3, n_in = h->nd.n_in;
    // the four factor arrays hold the 6 stages of `fsteps` steps: one batch contraction per fsteps steps
    // (K = 6 fsteps B); as many steps as fit a 1 GiB budget, at most 32
    const size_t per_step = 12 * ((size_t)g.sum_in + g.sum_out) * cap;       // floats
    size_t fsteps = ((size_t)1 << 28) / per_step;
    if (fsteps < 1) fsteps = 1;
    if (fsteps > 32) fsteps = 32;
    h->grad_fsteps = (int)fsteps;
    const size_t total = 5 * D * cap + 7 * n_in * cap + fsteps * per_step +
                         ((size_t)GRAD_MAX_KSPLIT + 1) * h->n_params;
    HIPCHK(h, hipMalloc(&h->grad_arena, total * sizeof(float)));
    float* p = h->grad_arena;
    for (int i = 0; i < 5; ++i) { h->g_US[i] = p; p += D * cap; }
    for (int i = 0; i < 6; ++i) { h->g_W[i] = p; p += n_in * cap; }
    h->g_lam = p; p += n_in * cap;
    h->g_HS = p; p += fsteps * 6 * (size_t)g.sum_in * cap;
    h->g_TS = p; p += fsteps * 6 * (size_t)g.sum_in * cap;
    h->g_AB = p; p += fsteps * 6 * (size_t)g.sum_out * cap;
    h->g_PB = p; p += fsteps * 6 * (size_t)g.sum_out * cap;
    h->g_part = p; p += (size_t)GRAD_MAX_KSPLIT * h->n_params;
    h->g_grad = p;
    h->grad_cap_B = cap;
    return CNF_OK;
}

// loss_and_grad of small batches of a small two-layer tanh network (or a one-layer one through its appended identity layer):
// the solve, the loss sums and the whole discrete adjoint in ONE launch, one wave per 16 samples (k_solve_wave<GRAD>,
// cnf_wave.hip), then the sum of the waves' partials.  TrainMode (VJP compute mode) and TestMode (exact trace).  *done = false:
// not this network / batch, or the launch gave up (a wait ran out, more steps than its store holds) -- nothing was written.
static cnf_status wave_loss_grad(cnf_handle h, int mode, const float* xs, const float* eps, int B, const cnf_solve_opts* opts,
                                 float* loss_out, float* grad, cnf_solve_stats* stats, void* stream, bool* done,
                                 float* loss_dev = nullptr /* submit: the loss goes here (device), nothing is waited for */) {
    *done = false;
    cnf_status s = CNF_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool train = mode == CNF_MODE_TRAIN;
    if (opts->kernel == CNF_KERNEL_GENERIC || !wave_grad_supported(h->nd_wave, B, train)) return CNF_OK;
    {
        const size_t per_step = wave_grad_traj_floats(h->nd_wave, B);
        int cap = WV_GCAP;
        while (cap > 64 && per_step * cap > ((size_t)1 << 28)) cap /= 2;       // <= 1 GiB of trajectory (B = 8192: 256 steps)
        // behind the trajectory: the step sizes, and one partial of the flat gradient per wave when there are more waves than
        // the arena's GRAD_MAX_KSPLIT partials
        const int waves = wave_grad_waves(B);
        const size_t part_f = waves > GRAD_MAX_KSPLIT ? (size_t)waves * h->n_params : 0;
        // TrainMode / VJP at small batches: the forward pass also files its evaluations' intermediates (the backward pass then
        // skips half of its products), while at least 256 steps of them fit in 256 MiB
        size_t rich_step = wave_grad_rich_floats(h->nd_wave, B, train);
        if (rich_step * 256 > ((size_t)1 << 26)) rich_step = 0;
        if (rich_step) { while (rich_step * cap > ((size_t)1 << 26)) cap /= 2; }
        const size_t need = per_step * cap + WV_GCAP + part_f + rich_step * cap;
        if (need > h->wg_traj_floats) {
            HIPCHK(h, hipDeviceSynchronize());
            if (h->wg_traj) { (void)hipFree(h->wg_traj); h->wg_traj = nullptr; h->wg_traj_floats = 0; }
            HIPCHK(h, hipMalloc(&h->wg_traj, need * sizeof(float)));
            h->wg_traj_floats = need;
        }
        WaveGradArgs wg;
        wg.traj = h->wg_traj; wg.traj_cap = cap; wg.hs_out = h->wg_traj + per_step * cap;
        wg.gpart = part_f ? h->wg_traj + per_step * cap + WV_GCAP : h->g_part;
        wg.lam_out = h->g_lam; wg.n_params = (int)h->n_params;
        wg.ys = h->nd.n_cond > 0 ? h->d_ys : nullptr;
        wg.rich = rich_step ? h->wg_traj + per_step * cap + WV_GCAP + part_f : nullptr;
        wg.lam1 = h->lam[0]; wg.lam2 = h->lam[1]; wg.lam3 = h->lam[2];
        Recorder rec;
        rec.wg = &wg;
        cnf_solve_stats sst{};
        PostHook ph{h->tmp_logpx, h->tmp_regs, h->d_sums, xs};
        if ((s = solve_core(h, mode, h->U[0], eps, nullptr, B, opts, &sst, stream, &rec, false, &ph)) != CNF_OK) return s;
        if (rec.wg_submitted) {
            HIPCHK(h, launch_grad_finish(wg.gpart, grad, (int)h->n_params, waves, h->d_state, h->d_sums, h->lam[0], h->lam[1], h->lam[2],
                                         train ? 1 : 0, loss_dev, st));
            h->grad_last_B = B;
            *done = true;
            return CNF_OK;
        }
        if (rec.wg_done) {
            HIPCHK(h, launch_grad_reduce(wg.gpart, grad, (int)h->n_params, waves, st));
            float* sums = reinterpret_cast<float*>(&h->h_state[2]);
            HIPCHK(h, hipMemcpyAsync(sums, h->d_sums, 5 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipStreamSynchronize(st));
            h->last_hs = rec.hs;
            h->grad_last_B = B;
            sst.launches += 1;
            if ((s = cnf_loss_from_sums(h, mode, sums, loss_out)) != CNF_OK) return s;
            if (stats) *stats = sst;
            *done = true;
        }
    }
    return CNF_OK;
}

extern "C" cnf_status cnf_loss_grad(cnf_handle h, const float* xs, const float* eps, int B,
                                    const cnf_solve_opts* opts, float* loss_out, float* grad,
                                    cnf_solve_stats* stats, void* stream) {
    const int mode = CNF_MODE_TRAIN;
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!xs || !eps || !opts || !loss_out || !grad) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (B < 1) return fail(h, CNF_ERR_BAD_SHAPE, "the loss is a mean over the batch: B must be >= 1");
    const GradLayout gl = grad_layout(h->nd);
    if (!grad_supported(h->nd, gl)) return fail(h, CNF_ERR_UNSUPPORTED, "network too wide for the gradient kernels");
    if ((s = ensure_capacity(h, B)) != CNF_OK) return s;
    if ((s = ensure_grad_capacity(h, B)) != CNF_OK) return s;
    hipStream_t st = (hipStream_t)stream;
    const NetDesc& nd = h->nd;
    const int n_in = nd.n_in, D = n_in + 3;
    const size_t n = (size_t)D * B;
    const AdjMfmaLayout am = adj_mfma_layout(nd, gl);
    // the pullback kernel follows the kernel choice of the solve: GENERIC -> VALU, otherwise MFMA when it fits
    const bool adj_mfma = opts->kernel != CNF_KERNEL_GENERIC && adj_mfma_supported(nd, am);
    if (!h->pt_valid) {
        HIPCHK(h, launch_transpose_params(nd, h->d_params, h->d_PT, st));
        HIPCHK(h, launch_pack_adj_images(nd, gl, am, h->d_params, h->d_adj_img, st));
        h->pt_valid = true;
    }

    {   // small batches of a small two-layer tanh network: everything in one launch (wave_loss_grad above)
        bool done = false;
        if ((s = wave_loss_grad(h, mode, xs, eps, B, opts, loss_out, grad, stats, stream, &done)) != CNF_OK || done) return s;
    }

    // ---- forward: u0, recorded solve, loss ------------------------------------------------------
    // (as an inference does: u0 is assembled from the data columns and the post-processing and the five loss sums are formed
    // inside the one-launch solves -- the recording forms included --, behind the solve otherwise; the final state goes
    // straight to fsol where the kernel can write it there)
    Recorder rec;
    cnf_solve_stats sst{};
    float* fsol = h->g_US[1];
    PostHook ph{h->tmp_logpx, h->tmp_regs, h->d_sums, xs};
    for (;;) {
        ph.launched = false;
        if ((s = solve_core(h, mode, h->U[0], eps, fsol, B, opts, &sst, stream, &rec, true, &ph)) != CNF_OK) return s;
        if (!rec.overflow) break;
        if ((s = traj_reserve(h, rec.n + 8)) != CNF_OK) return s;       // more steps than slots: grow, solve again
    }
    h->last_hs = rec.hs;
    if (!ph.launched) {                                    // (the one-attempt-at-a-time drivers leave it to the caller)
        enqueue_post(h, 1, h->last_state, ph, B, false, st);
        HIPCHK(h, hipGetLastError());
    }
    // (the five sums travel to the host behind the backward pass: the loss VALUE is not needed to start it)
    float* sums = reinterpret_cast<float*>(&h->h_state[2]);             // pinned; the initial-state slot is free by now
    HIPCHK(h, hipMemcpyAsync(sums, h->d_sums, 5 * sizeof(float), hipMemcpyDeviceToHost, st));

    // ---- backward: discrete adjoint of the recorded steps ---------------------------------------
    // capacity is in samples of cap_B; with B <= cap_B at least grad_fsteps steps fit
    // (CNF_GRAD_FSTEPS=n: contract after every n steps instead -- measurements: fewer steps per contraction keep the factor rows
    // in the infinity cache, more amortise its launch)
    static const int fsteps_env = [] { const char* e = getenv("CNF_GRAD_FSTEPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    int fsteps = (int)std::min<size_t>(32, (size_t)h->grad_fsteps * h->grad_cap_B / (size_t)B);
    if (fsteps_env > 0 && fsteps_env < fsteps) fsteps = fsteps_env;
    int ksplit = 1, filed = 0;
    HIPCHK(h, hipMemsetAsync(h->g_part, 0, (size_t)GRAD_MAX_KSPLIT * h->n_params * sizeof(float), st));
    HIPCHK(h, launch_final_cotangent(nd, h->lam[2], fsol, h->g_lam, B, st));
    const float invB = 1.0f / (float)B;
    const float lam_l = invB, lam_E = h->lam[0] * invB, lam_n = h->lam[1] * invB;   // constant scalar rows
    static const float A[6][5] = {
        {0, 0, 0, 0, 0},
        {TS_A21, 0, 0, 0, 0},
        {TS_A31, TS_A32, 0, 0, 0},
        {TS_A41, TS_A42, TS_A43, 0, 0},
        {TS_A51, TS_A52, TS_A53, TS_A54, 0},
        {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65}};
    static const float Bw[6] = {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76};
    if (adj_mfma && adj3b_supported(nd) && h->mfma.d_img3b) {
        // The headline shape: the steps whose factor rows fit the arena in ONE launch on split-bf16 products (cnf_adj3b.hip;
        // the image is the forward kernels'), then one contraction over them -- 2 launches per run of steps instead of one
        // per step and one per run.
        if ((s = traj_reserve(h, rec.n)) != CNF_OK) return s;
        const int run = std::min(fsteps, ADJ3B_MAX_STEPS);
        for (int hi = rec.n - 1; hi >= 0; hi -= run) {
            const int lo = std::max(0, hi - run + 1), cnt = hi - lo + 1;
            Adj3bSteps M{};
            M.traj = h->traj; M.slot_stride = traj_slot_floats(h); M.n = n;
            M.step_hi = hi; M.step_lo = lo;
            for (int j = 0; j < cnt; ++j) M.hs[j] = rec.hs[hi - j];
            M.eps = eps; M.lam = h->g_lam; M.lam_out = h->g_lam;
            M.HS = h->g_HS; M.TS = h->g_TS; M.AB = h->g_AB; M.PB = h->g_PB;
            M.lam_l = lam_l; M.lam_E = lam_E; M.lam_n = lam_n;
            for (int i = 0; i < 6; ++i) M.bw[i] = Bw[i];
            for (int m = 0; m < 6; ++m) for (int d = 0; d < 5; ++d) M.kc[m][d] = m - 1 - d >= 0 ? A[m][m - 1 - d] : 0.f;
            M.B = B;
            if (adj3b_split(B, cnt)) {          // (batches that leave CUs idle: two launches, the parked state in between)
                const size_t need = adj3b_park_floats(B, cnt);
                if (need > h->park_floats) {
                    HIPCHK(h, hipStreamSynchronize(st));
                    if (h->d_park) { (void)hipFree(h->d_park); h->d_park = nullptr; h->park_floats = 0; }
                    if (hipMalloc(&h->d_park, need * sizeof(float)) == hipSuccess) h->park_floats = need;
                    else { (void)hipGetLastError(); h->d_park = nullptr; }      // (no room for the parked state: one launch per run)
                }
                M.park = h->d_park;
            }
            HIPCHK(h, launch_adj3b(nd, gl, h->mfma.d_img3b, M, st));
            int ks, ch;
            grad_ksplit(nd, gl, 6 * cnt * B, &ks, &ch);
            if (ks > ksplit) ksplit = ks;
            HIPCHK(h, launch_wgrad(nd, gl, h->g_AB, h->g_PB, h->g_HS, h->g_TS, h->g_part, (int)h->n_params, 6 * cnt * B, ks, ch, st));
        }
    } else {
    // (MFMA pullback: the steps of a run are gathered and launched together -- as two launches over the whole run where the batch
    // leaves CUs idle, else one launch per step)
    int run0 = 0;                                          // first entry of the current run in h_steps
    if (adj_mfma && rec.n > h->steps_cap) {
        HIPCHK(h, hipStreamSynchronize(st));
        if (h->d_steps) { (void)hipFree(h->d_steps); h->d_steps = nullptr; }
        if (h->h_steps) { (void)hipHostFree(h->h_steps); h->h_steps = nullptr; }
        h->steps_cap = 0;
        const int cap = rec.n + 32;
        HIPCHK(h, hipMalloc(&h->d_steps, (size_t)cap * sizeof(AdjStepArgs)));
        HIPCHK(h, hipHostMalloc(&h->h_steps, (size_t)cap * sizeof(AdjStepArgs)));
        h->steps_cap = cap;
    }
    for (int step = rec.n - 1; step >= 0; --step) {
        float* un;
        if ((s = traj_slot(h, step, &un)) != CNF_OK) return s;
        const float hs = rec.hs[step];
        // stage states: U_1 = u_n, U_2..U_6 were filed behind it by the forward pass
        const float* US[6];
        for (int i = 0; i < 6; ++i) US[i] = un + (size_t)i * n;
        AdjStepArgs S{};
        for (int i = 5; i >= 0; --i) {
            AdjArgs a{};
            a.P = h->d_params; a.PT = h->d_PT; a.ustage = US[i]; a.eps = eps;
            a.ys = nd.n_cond > 0 ? h->d_ys : nullptr;
            a.lam = h->g_lam;
            a.nw = 0;
            for (int m = i + 1; m < 6; ++m) { a.w[a.nw] = h->g_W[m]; a.wc[a.nw] = A[m][i]; ++a.nw; }
            for (int k = a.nw; k < 5; ++k) { a.w[k] = h->g_lam; a.wc[k] = 0.f; }     // unused slots: a readable array, weight 0
            a.cb = Bw[i]; a.hstep = hs;
            a.c_l = hs * Bw[i] * lam_l; a.c_E = hs * Bw[i] * lam_E; a.c_n = hs * Bw[i] * lam_n;
            a.w_out = h->g_W[i];
            // every stage evaluation files its factors behind the earlier ones: rows [slot B, (slot + 1) B)
            const size_t slot = (size_t)filed * 6 + i;
            a.HS = h->g_HS + slot * B * gl.sum_in; a.TS = h->g_TS + slot * B * gl.sum_in;
            a.AB = h->g_AB + slot * B * gl.sum_out; a.PB = h->g_PB + slot * B * gl.sum_out;
            a.B = B;
            if (adj_mfma) S.st[i] = a;
            else HIPCHK(h, launch_adj(nd, gl, a, st));
        }
        if (adj_mfma) {        // the six stage pullbacks and the lambda update of this step in ONE launch
            S.first = 5; S.last = 0; S.B = B; S.lam_update = 1; S.lam_out = h->g_lam;
            for (int m = 0; m < 6; ++m) for (int d = 0; d < 5; ++d) S.kc[m][d] = m - 1 - d >= 0 ? A[m][m - 1 - d] : 0.f;
            h->h_steps[rec.n - 1 - step] = S;
        } else {
            StageK ws{};
            ws.nk = 6;
            for (int i = 0; i < 6; ++i) ws.k[i] = h->g_W[i];
            HIPCHK(h, launch_lambda_update(h->g_lam, ws, (size_t)n_in * B, st));
        }
        // Wbar += sum over the filed stage evaluations and their samples: one contraction with K = 6 filed B
        if (++filed == fsteps || step == 0) {
            if (adj_mfma) {                                // the run's pullbacks
                const int cnt = filed;
                // (two launches per SUB-run of steps whose parked rows stay in the infinity cache -- measured at config 5's network:
                // whole runs of 10-22 steps, 0.2-0.8 GB of rows, cost 2 % at B = 256 and 6 % at 2048 against step by step)
                const size_t per_step = adj_mfma_scratch_floats(am, (size_t)B, 1);
                int nsub = (int)std::max<size_t>(1, std::min<size_t>((size_t)cnt, ((size_t)48 << 20) / (per_step * sizeof(float))));
                bool two = adj_mfma_run_split(nd, am, B, nsub);
                if (two) {
                    const size_t need = per_step * nsub;
                    if (need > h->sc_floats) {
                        HIPCHK(h, hipStreamSynchronize(st));
                        if (h->d_sc) { (void)hipFree(h->d_sc); h->d_sc = nullptr; h->sc_floats = 0; }
                        if (hipMalloc(&h->d_sc, need * sizeof(float)) == hipSuccess) h->sc_floats = need;
                        else { (void)hipGetLastError(); two = false; }          // (no room for the parked rows: one launch per step)
                    }
                }
                if (two) {
                    HIPCHK(h, hipMemcpyAsync(h->d_steps + run0, h->h_steps + run0, (size_t)cnt * sizeof(AdjStepArgs), hipMemcpyHostToDevice, st));
                    for (int j = 0; j < cnt; j += nsub)
                        HIPCHK(h, launch_adj_mfma_run(nd, gl, am, h->d_adj_img, h->d_steps + run0 + j, h->h_steps + run0 + j, std::min(nsub, cnt - j),
                                                      B, h->d_sc, st));
                } else {
                    for (int j = 0; j < cnt; ++j) HIPCHK(h, launch_adj_mfma_step(nd, gl, am, h->d_adj_img, h->h_steps[run0 + j], st));
                }
                run0 += cnt;
            }
            int ks, ch;
            grad_ksplit(nd, gl, 6 * filed * B, &ks, &ch);
            if (ks > ksplit) ksplit = ks;
            HIPCHK(h, launch_wgrad(nd, gl, h->g_AB, h->g_PB, h->g_HS, h->g_TS, h->g_part, (int)h->n_params,
                                   6 * filed * B, ks, ch, st));
            filed = 0;
        }
    }
    }
    HIPCHK(h, launch_grad_reduce(h->g_part, grad, (int)h->n_params, ksplit, st));
    h->grad_last_B = B;                                    // (g_lam now holds d loss / d u(t0): cnf_grad_x)
    HIPCHK(h, hipStreamSynchronize(st));
    if ((s = cnf_loss_from_sums(h, mode, sums, loss_out)) != CNF_OK) return s;
    if (stats) *stats = sst;
    return CNF_OK;
}

// TestMode: loss(icnf, TestMode(), xs, ps, st) = -mean(logpx) (src/base_icnf.jl:489-497) and its gradient w.r.t. the flat
// parameters through the exact-trace solve -- what the reference's call tests and its benchmark suite differentiate besides
// the TrainMode loss (test/call_tests.jl `diff_loss` with omode = TestMode(); benchmark/benchmarks.jl:60-99 "AD-1-order" /
// "test").  Implemented for the networks k_solve_wave<GRAD> takes (two tanh layers or one, n_in <= 16, <= 64 hidden units,
// n_in + n_cond <= 16, B <= 8192); CNF_ERR_UNSUPPORTED otherwise.
extern "C" cnf_status cnf_loss_grad_test(cnf_handle h, const float* xs, int B, const cnf_solve_opts* opts, float* loss_out,
                                         float* grad, cnf_solve_stats* stats, void* stream) {
    const int mode = CNF_MODE_TEST;
    cnf_status s = check_call(h, mode, B);
    if (s != CNF_OK) return s;
    if (!xs || !opts || !loss_out || !grad) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (B < 1) return fail(h, CNF_ERR_BAD_SHAPE, "the loss is a mean over the batch: B must be >= 1");
    if ((s = ensure_capacity(h, B)) != CNF_OK) return s;
    if ((s = ensure_grad_capacity(h, B)) != CNF_OK) return s;
    bool done = false;
    if ((s = wave_loss_grad(h, mode, xs, nullptr, B, opts, loss_out, grad, stats, stream, &done)) != CNF_OK) return s;
    if (done) return CNF_OK;
    // ---- every other network: the recorded exact-trace solve, then k_adj_test (cnf_gradt.hip) over all of its steps in one launch ----
    hipStream_t st = (hipStream_t)stream;
    const NetDesc& nd = h->nd;
    const int n_in = nd.n_in, D = n_in + 1;
    float* u0 = h->g_US[0];
    launch_build_u0(xs, u0, nd.nvars, D, B, st);
    // the recorded forward pass files u_n after every accepted step (host-driven, one attempt at a time: solve_core's recording
    // branch of the streamed driver).  Networks whose TestMode runs inside the fused step kernels (two layers, closed-form trace)
    // take the generic right-hand side there: that branch is the one place where a TestMode solve records.
    cnf_solve_opts ropts = *opts;
    if (mfma_supported(h->mfma, nd, false, B)) ropts.kernel = CNF_KERNEL_GENERIC;
    Recorder rec;
    cnf_solve_stats sst{};
    float* fsol = h->g_US[1];
    for (;;) {
        if ((s = solve_core(h, mode, u0, nullptr, fsol, B, &ropts, &sst, stream, &rec)) != CNF_OK) return s;
        if (!rec.overflow) break;
        if ((s = traj_reserve(h, rec.n + 8)) != CNF_OK) return s;
    }
    h->last_hs = rec.hs;
    launch_post(nd, 0, fsol, h->tmp_logpx, h->tmp_regs, B, st);
    launch_loss_sums(h->tmp_logpx, h->tmp_regs, B, h->d_sums, st);
    float* sums = reinterpret_cast<float*>(&h->h_state[2]);
    HIPCHK(h, hipMemcpyAsync(sums, h->d_sums, 5 * sizeof(float), hipMemcpyDeviceToHost, st));
    // the step sizes to the device (behind the steps in the trajectory store's step-size array), scratch and partials of the kernel
    if ((s = traj_reserve(h, rec.n + 1)) != CNF_OK) return s;
    if (rec.n > 0) HIPCHK(h, hipMemcpyAsync(h->traj_hs, rec.hs.data(), (size_t)rec.n * sizeof(float), hipMemcpyHostToDevice, st));
    const int G = adj_test_workgroups(B);
    const size_t per_wg = adj_test_scratch_floats(nd) + h->n_params;
    if ((size_t)G * per_wg > h->gt_floats) {
        HIPCHK(h, hipStreamSynchronize(st));
        if (h->d_gt) { (void)hipFree(h->d_gt); h->d_gt = nullptr; h->gt_floats = 0; }
        HIPCHK(h, hipMalloc(&h->d_gt, (size_t)G * per_wg * sizeof(float)));
        h->gt_floats = (size_t)G * per_wg;
    }
    float* first;
    if ((s = traj_slot(h, 0, &first)) != CNF_OK) return s;
    AdjTestArgs ta{};
    ta.P = h->d_params; ta.traj = first; ta.slot_stride = traj_slot_floats(h); ta.hs = h->traj_hs; ta.nsteps = rec.n;
    ta.ys = nd.n_cond > 0 ? h->d_ys : nullptr; ta.lam_l = 1.0f / (float)B; ta.lam_out = h->g_lam;
    ta.gpart = h->d_gt; ta.scratch = h->d_gt + (size_t)G * h->n_params; ta.scratch_per_wg = adj_test_scratch_floats(nd);
    ta.B = B; ta.n_params = (int)h->n_params;
    if (launch_adj_test(nd, ta, st) != hipSuccess) { (void)hipGetLastError(); return fail(h, CNF_ERR_UNSUPPORTED, "network too wide for the TestMode adjoint kernel"); }
    HIPCHK(h, launch_grad_reduce(h->d_gt, grad, (int)h->n_params, G, st));
    h->grad_last_B = B;                                    // (g_lam holds d loss / d z(t0): cnf_grad_x)
    HIPCHK(h, hipStreamSynchronize(st));
    if ((s = cnf_loss_from_sums(h, mode, sums, loss_out)) != CNF_OK) return s;
    if (stats) { *stats = sst; stats->launches += 2; }
    return CNF_OK;
}

extern "C" cnf_status cnf_loss_grad_test_host(cnf_handle h, const float* xs, int B, const cnf_solve_opts* opts, float* loss_out,
                                              float* grad, cnf_solve_stats* stats) {
    cnf_status s = check_call(h, CNF_MODE_TEST, B);
    if (s != CNF_OK) return s;
    if (!xs || !grad) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (B < 1) return fail(h, CNF_ERR_BAD_SHAPE, "the loss is a mean over the batch: B must be >= 1");
    HIPCHK(h, hipSetDevice(h->device));
    if ((s = ensure_grad_capacity(h, B)) != CNF_OK) return s;
    const size_t nx = (size_t)h->nd.nvars * B;
    if ((s = ensure_stage(h, nx)) != CNF_OK) return s;
    HIPCHK(h, hipMemcpy(h->stage, xs, nx * sizeof(float), hipMemcpyHostToDevice));
    if ((s = cnf_loss_grad_test(h, h->stage, B, opts, loss_out, h->g_grad, stats, nullptr)) != CNF_OK) return s;
    HIPCHK(h, hipMemcpy(grad, h->g_grad, h->n_params * sizeof(float), hipMemcpyDeviceToHost));
    return CNF_OK;
}

// ---- submitted gradients: a training loop that never waits for the GPU ---------------------------------------------------------
// cnf_loss_grad_submit enqueues what cnf_loss_grad (TrainMode) / cnf_loss_grad_test (TestMode) compute -- solve, loss, discrete
// adjoint, the sum of the partials -- and returns; the loss (one float) and the gradient are left in DEVICE memory, stream-
// ordered, for whatever the caller enqueues next (the optimiser's update, then cnf_set_params_async with the new parameters and
// the next submission).  cnf_loss_grad_collect (oldest first; the queue is the one of cnf_inference_submit: at most three in
// flight, all on one stream) reports how the launch ended; a launch that gave up has delivered ZEROS and a NaN loss and is
// reported as CNF_ERR_UNSUPPORTED.  Only where the gradient runs in the launch of the solve (k_solve_wave<GRAD>); otherwise
// CNF_ERR_UNSUPPORTED at once, and the caller uses the synchronous call.
extern "C" cnf_status cnf_loss_grad_submit(cnf_handle h, int mode, const float* xs, const float* eps, int B, const cnf_solve_opts* opts,
                                           float* loss_dev, float* grad, void* stream) {
    if (!h) return CNF_ERR_BAD_ARG;
    if (h->submitted.size() >= 3) return fail(h, CNF_ERR_BAD_ARG, "three launches are submitted already: collect one first");
    h->collecting = true;
    cnf_status s = check_call(h, mode, B);
    if (s == CNF_OK && (!xs || !opts || !loss_dev || !grad || (mode == CNF_MODE_TRAIN && !eps))) s = fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (s == CNF_OK && B < 1) s = fail(h, CNF_ERR_BAD_SHAPE, "the loss is a mean over the batch: B must be >= 1");
    if (s == CNF_OK) s = ensure_capacity(h, B);
    if (s == CNF_OK) s = ensure_grad_capacity(h, B);
    bool done = false;
    if (s == CNF_OK) {
        h->submitting = true; h->sub_taken = false;
        s = wave_loss_grad(h, mode, xs, eps, B, opts, nullptr, grad, nullptr, stream, &done, loss_dev);
        h->submitting = false;
    }
    h->collecting = false;
    if (s != CNF_OK) return s;
    if (!done) return fail(h, CNF_ERR_UNSUPPORTED, "no in-launch gradient for this network / batch: use cnf_loss_grad");
    return CNF_OK;
}

extern "C" cnf_status cnf_loss_grad_collect(cnf_handle h, cnf_solve_stats* stats) { return cnf_inference_collect(h, stats); }

// cnf_set_params without the host waits: the copy (and the packing launches) are enqueued on `stream` and the call returns.
// For a caller whose every launch on this handle goes to that one stream (then stream order is all the synchronisation there
// is to do): the parameter update between two submitted gradients.
extern "C" cnf_status cnf_set_params_async(cnf_handle h, const float* flat_dev, size_t n, void* stream) {
    if (!h || !flat_dev) return CNF_ERR_BAD_ARG;
    if (n != h->n_params) return fail(h, CNF_ERR_BAD_SHAPE, "parameter count does not match the layer sizes");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    if (g_submitted_inflight > 0 && g_submitted_stream != s)
        return fail(h, CNF_ERR_BAD_ARG, "cnf_set_params_async: launches are in flight on another stream");
    // A submitted INFERENCE that gives up is run again by its collect call -- with the parameters (and conditioning) the handle
    // holds then.  Changing them under it would silently change its result: such submissions are settled first (host wait).
    // Submitted gradients are not re-run (a launch that gave up reports CNF_ERR_UNSUPPORTED), so they stay in flight.
    for (const auto& sub : h->submitted)
        if (sub.launched && !sub.grad) { const cnf_status ss = settle_submitted(h); if (ss != CNF_OK) return ss; break; }
    HIPCHK(h, hipMemcpyAsync(h->d_params, flat_dev, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    cnf_status ms = mfma_plan_pack(h->mfma, h->nd, h->d_params, s);
    if (ms != CNF_OK) return fail(h, ms, "MFMA weight packing failed");
    h->have_params = true;
    h->pt_valid = false; h->img_valid = false; h->bimg_valid = false;
    h->cond_B = 0;
    return CNF_OK;
}

// d loss / d xs of the last cnf_loss_grad call: the adjoint state at t0 is d loss / d u(t0), and u0 = (xs; zeros) -- its first
// nvars rows, [B][nvars] as xs is laid out.  (The backward sweep leaves it in g_lam; nothing is recomputed.)
extern "C" cnf_status cnf_grad_x(cnf_handle h, float* gx, int B, void* stream) {
    if (!h || !gx) return CNF_ERR_BAD_ARG;
    if (B < 1 || B != h->grad_last_B || !h->g_lam) return fail(h, CNF_ERR_BAD_ARG, "cnf_grad_x: no gradient of a batch of this size has been computed");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync(gx, (size_t)h->nd.nvars * sizeof(float), h->g_lam, (size_t)h->nd.n_in * sizeof(float),
                               (size_t)h->nd.nvars * sizeof(float), (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return CNF_OK;
}

extern "C" int cnf_grad_steps(cnf_handle h, float* hs, int cap) {
    if (!h) return -1;
    const int n = (int)h->last_hs.size();
    if (hs) for (int i = 0; i < n && i < cap; ++i) hs[i] = h->last_hs[i];
    return n;
}

extern "C" cnf_status cnf_loss_grad_host(cnf_handle h, const float* xs, const float* eps, int B,
                                         const cnf_solve_opts* opts, float* loss_out, float* grad,
                                         cnf_solve_stats* stats) {
    cnf_status s = check_call(h, CNF_MODE_TRAIN, B);
    if (s != CNF_OK) return s;
    if (!xs || !eps || !grad) return fail(h, CNF_ERR_BAD_ARG, "null pointer");
    if (B < 1) return fail(h, CNF_ERR_BAD_SHAPE, "the loss is a mean over the batch: B must be >= 1");
    HIPCHK(h, hipSetDevice(h->device));
    if ((s = ensure_grad_capacity(h, B)) != CNF_OK) return s;
    const size_t nx = (size_t)h->nd.nvars * B, ne = (size_t)h->nd.n_in * B;
    if ((s = ensure_stage(h, nx + ne)) != CNF_OK) return s;
    float* x_d = h->stage;
    float* e_d = x_d + nx;
    hipError_t e = hipMemcpy(x_d, xs, nx * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(e_d, eps, ne * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) s = fail(h, CNF_ERR_HIP, hipGetErrorString(e));
    if (s == CNF_OK) s = cnf_loss_grad(h, x_d, e_d, B, opts, loss_out, h->g_grad, stats, nullptr);
    if (s == CNF_OK) {
        e = hipMemcpy(grad, h->g_grad, h->n_params * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) s = fail(h, CNF_ERR_HIP, hipGetErrorString(e));
    }
    return s;
}
