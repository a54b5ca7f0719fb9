// Shared device/host definitions for libcnfhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CNF_MAX_LAYERS 8

// Network description, passed to kernels by value (kernarg segment -> scalar loads).
struct NetDesc {
    int n_layers;
    int dims[CNF_MAX_LAYERS + 1];   // n_in, h1, ..., n_out
    int acts[CNF_MAX_LAYERS];
    int w_off[CNF_MAX_LAYERS];      // float offset of layer l's weight in the flat vector
    int b_off[CNF_MAX_LAYERS];      // float offset of layer l's bias
    int n_in;                       // nvars + naugs
    int nvars, naugs;
    int norm_z, norm_j, norm_z_aug; // !iszero(lambda) (src/base_icnf.jl:42-51)
    int jvp;                        // 0: VJP (DIVecJacMatrixMode), 1: JVP (DIJacVecMatrixMode)
    int max_dim, sum_dims;          // max_l dims[l]; sum_{l=0..L} dims[l]
    int n_cond;                     // conditioning rows; dims[0] stays n_in, the first layer's weight has
                                    // n_in + n_cond columns (column-major: the z columns come first)
    int wy_off;                     // float offset of the conditioning columns of layer 0's weight
    int id2;                        // layer 1 is an identity map (W = I, b = 0) appended behind the parameters of a ONE-layer
                                    // network so that the two-layer wave kernels take it (cnf_abi.hip, cnf_wave.hip)
};

// Device-resident integrator state: lets the step controller run on the GPU so that a
// solve needs no host round trip per step.
struct StepState {
    float t, dt, qold;      // current time, proposed step size (>0), PI controller memory
    float t0, t1, tdir;     // span and direction (+1/-1)
    float h;                // signed step of the attempt in flight (clipped to t1)
    float abstol, reltol;
    float eest;             // last error estimate
    float d0, d1;           // initial-dt intermediates
    int adaptive;
    int cur;                // which of the two (u, k1) buffer sets holds the current state
    int done;               // reached t1
    int naccept, nreject;
    int nonfinite;
    int n_partials;         // entries of the error-partials array
};

#ifdef __HIPCC__
// Integrator state -> pinned host mirror: one 8-byte system-scope store per state word, {tag = launch index, word}
// (protocol and reader: cnf_mirror.h).  No wait, no fence: the reader validates the tags, and a release fence would
// write back this XCD's whole L2 on every launch.
__device__ __forceinline__ void mirror_store(void* mirror, unsigned seq, const StepState& z) {
    if (!mirror) return;
    static_assert(sizeof(StepState) % 4 == 0, "copied as 32-bit words");
    const unsigned* src = reinterpret_cast<const unsigned*>(&z);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(mirror);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(StepState) / 4); ++i)
        __hip_atomic_store(dst + i, ((unsigned long long)seq << 32) | src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

// Tsit5 (Tsitouras 2011); same digits as oracle/cnf_oracle.py.
#define TS_A21 0.161f
#define TS_A31 -0.008480655492356989f
#define TS_A32 0.335480655492357f
#define TS_A41 2.8971530571054935f
#define TS_A42 -6.359448489975075f
#define TS_A43 4.3622954328695815f
#define TS_A51 5.325864828439257f
#define TS_A52 -11.748883564062828f
#define TS_A53 7.4955393428898365f
#define TS_A54 -0.09249506636175525f
#define TS_A61 5.86145544294642f
#define TS_A62 -12.92096931784711f
#define TS_A63 8.159367898576159f
#define TS_A64 -0.071584973281401f
#define TS_A65 -0.028269050394068383f
#define TS_A71 0.09646076681806523f
#define TS_A72 0.01f
#define TS_A73 0.4798896504144996f
#define TS_A74 1.379008574103742f
#define TS_A75 -3.290069515436081f
#define TS_A76 2.324710524099774f
#define TS_BT1 -0.00178001105222577714f
#define TS_BT2 -0.0008164344596567469f
#define TS_BT3 0.007880878010261995f
#define TS_BT4 -0.1447110071732629f
#define TS_BT5 0.5823571654525552f
#define TS_BT6 -0.45808210592918697f
#define TS_BT7 0.015151515151515152f

// Tsit5 stage coefficient row s (0-based stage index 1..6 -> a_{s+1, 1..s}).
__host__ __device__ inline void tsit5_row(int s, float* a) {
    const float A[7][6] = {
        {0, 0, 0, 0, 0, 0},
        {TS_A21, 0, 0, 0, 0, 0},
        {TS_A31, TS_A32, 0, 0, 0, 0},
        {TS_A41, TS_A42, TS_A43, 0, 0, 0},
        {TS_A51, TS_A52, TS_A53, TS_A54, 0, 0},
        {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65, 0},
        {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76}};
    for (int j = 0; j < 6; ++j) a[j] = A[s][j];
}

#ifdef __HIPCC__
// ---- step controller (shared by k_controller and the fused MFMA step kernel) -----------------
// OrdinaryDiffEq-style PI controller for Tsit5 (third party in the
// reference, restated from the published scheme, mirrored by the oracle).
__device__ __forceinline__ void ctrl_set_attempt_h(StepState* st) {
    float rem = fabsf(st->t1 - st->t);
    float h = st->dt < rem ? st->dt : rem;
    st->h = st->tdir * h;
}
// p0 = sum (err/sc)^2 over all D*B entries, p1 = non-finite count of the attempt just made
__device__ __forceinline__ void ctrl_after_step(StepState* st, float p0, float p1, float n_total) {
    const float habs = fabsf(st->h);
    bool accept = true;
    float q = 1.f, q11 = 1.f, eest = 0.f;
    if (p1 > 0.f) st->nonfinite = 1;
    if (st->adaptive) {
        eest = sqrtf(p0 / n_total);
        if (!(eest == eest)) { st->nonfinite = 1; eest = 1e30f; }
        accept = eest <= 1.0f;
        const float beta1 = 7.f / 50.f, beta2 = 2.f / 25.f, gamma = 0.9f;
        const float qmin = 0.2f, qmax = 10.f;
        q11 = powf(fmaxf(eest, 1e-30f), beta1);
        q = q11 / powf(st->qold, beta2);
        q = fmaxf(1.f / qmax, fminf(1.f / qmin, q / gamma));
        st->eest = eest;
        if (accept) {
            if (q >= 1.0f && q <= 1.2f) q = 1.f;
            st->qold = fmaxf(eest, 1e-4f);
            st->dt = habs / q;
        } else {
            st->dt = habs / fminf(1.f / qmin, q11 / gamma);
        }
    }
    if (accept) {
        st->naccept += 1;
        st->t = st->t + st->h;
        st->cur ^= 1;
        float tol = 100.f * 1.1920929e-7f * fmaxf(1.f, fabsf(st->t1));
        if (fabsf(st->t1 - st->t) <= tol) { st->t = st->t1; st->done = 1; }
    } else {
        st->nreject += 1;
    }
    if (st->nonfinite) st->done = 1;
    if (!st->done) ctrl_set_attempt_h(st);
}

// phases of the device controller: 0/1 = the two norms of the automatic initial dt (Hairer; OrdinaryDiffEq's
// ode_determine_initdt, third party), 2 = after a step attempt
__device__ __forceinline__ void ctrl_phase(StepState* st, int phase, float p0, float p1, float n_total) {
    const float span = fabsf(st->t1 - st->t0);
    if (phase == 0) {
        float d0 = sqrtf(p0 / n_total), d1 = sqrtf(p1 / n_total);
        float dt0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
        dt0 = fminf(dt0, span);
        st->d0 = dt0;   // keep dt0
        st->d1 = d1;
        st->h = st->tdir * dt0;
    } else if (phase == 1) {
        float dt0 = st->d0, d1 = st->d1;
        float d2 = sqrtf(p0 / n_total) / dt0;
        float m = fmaxf(d1, d2);
        float dt1 = (m <= 1e-15f) ? fmaxf(1e-6f, dt0 * 1e-3f) : powf(0.01f / m, 0.2f);
        st->dt = fminf(fminf(100.f * dt0, dt1), span);
        ctrl_set_attempt_h(st);
    } else {
        ctrl_after_step(st, p0, p1, n_total);
    }
}

// ---- activations: value and derivative w.r.t. the pre-activation ------------------------
__device__ __forceinline__ float cnf_sigmoid(float a) { return 1.0f / (1.0f + __expf(-a)); }

// tanh accurate to ~2 ulp over the whole range without the cancellation of
// 1 - 2/(exp(2a)+1) near 0 (needed: outputs are compared at 1e-4 relative).
__device__ __forceinline__ float cnf_tanh(float a) {
    float x = fabsf(a);
    float r;
    if (x < 0.3f) {
                float x2 = x * x;
        // tanh(x) = x * (1 + x2*p(x2)) ; Taylor up to x^11 (truncation < 2e-9 rel for x<0.3)
        float p = -0.0088632355f;                  // -1382/155925
        p = fmaf(p, x2, 0.021869488f);             // 62/2835
        p = fmaf(p, x2, -0.053968254f);            // -17/315
        p = fmaf(p, x2, 0.13333333f);              // 2/15
        p = fmaf(p, x2, -0.33333334f);             // -1/3
        r = fmaf(x * x2, p, x);
    } else {
        float e = __expf(2.0f * x);                // v_exp_f32 path
        r = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    }
    return copysignf(r, a);
}

// tanh(a) = 1 - 2/(exp(2a) + 1): v_mul, v_exp, v_add, v_rcp, v_fma.  Absolute error
// <= 2e-7 (one rounding of values near 1), which is what every other fp32 activation value
// carries; measured against the parity metric it is indistinguishable from libm's tanh.
__device__ __forceinline__ float tanh_fast(float a) {
    const float t = __builtin_amdgcn_exp2f(a * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}

__device__ __forceinline__ void cnf_act(int kind, float a, float& h, float& d) {
    switch (kind) {
        case 0: h = a; d = 1.0f; break;
        case 1: h = cnf_tanh(a); d = fmaf(-h, h, 1.0f); break;
        case 2: { float s = cnf_sigmoid(a); h = s; d = s * (1.0f - s); } break;
        case 3: { h = (a > 15.0f) ? a : log1pf(__expf(a)); d = cnf_sigmoid(a); } break;
        case 4: h = fmaxf(a, 0.0f); d = (a > 0.0f) ? 1.0f : 0.0f; break;
        case 5: { float s = cnf_sigmoid(a); h = a * s; d = s * (1.0f + a * (1.0f - s)); } break;
        default: { float e = __expf(fminf(a, 0.0f)); h = (a > 0.0f) ? a : e - 1.0f; d = (a > 0.0f) ? 1.0f : e; } break;
    }
}
#endif
