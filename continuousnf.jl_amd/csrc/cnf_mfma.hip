// Fused MFMA kernels of libcnfhip (gfx950): the whole augmented RHS -- Dense+activation
// forward, the reverse (VJP) sweep, the eps^T J eps trace estimate and the regulariser
// rows -- and, one level up, a whole Tsit5 step (6 RHS evaluations, the stage
// combinations, the embedded error estimate) in ONE launch, with the stage vectors in
// registers and every activation in LDS.  HBM traffic per step and sample tile: read u, k1
// and eps, write u_new and k7.
//
// Reference functions taken over (file:line under /root/reference): augmented_f
// Matrix/Train/VJP src/icnf.jl:318-350 (and its in-place twin :352-382); the Dense layers
// and the pullback are Lux/Enzyme there (third party), restated per SURVEY.md Appendix B.
//
// Work decomposition (one workgroup = 8 waves = 512 lanes, 2 waves per SIMD, 1 per CU):
//   - a workgroup owns a tile of 32 samples = two teams of 4 waves, one per 16-sample MFMA
//     column tile (team = wave >> 2);
//   - every layer is a GEMM  Out[o][s] = sum_k W[o][k] X[k][s]  on v_mfma_f32_16x16x4_f32:
//     A = a 16-row slab of W (forward) or of W^T (reverse), B = the team's 16 samples of X;
//     wave fg of a team takes the 16-row output tiles fg, fg+4 (then +8, ...), with
//     fg = (wave + 2*team) & 3 so that layers with fewer than 4 tiles use different SIMDs
//     for the two teams;
//   - the accumulator layout (lane = sample, 4 consecutive rows per lane) is exactly what
//     one ds_write_b128 stores into the [sample][feature] activation image, and what one
//     ds_read_b128 fetches as the next layer's B operand for 4 consecutive MFMA k-steps:
//     k-step c of block u contracts feature 16u + 4*(lane>>4) + c on both operands;
//   - the reverse sweep writes g_l over h_l in place (sigma' is recomputed from h), so one
//     image per layer serves both sweeps;
//   - weights live in LDS for the whole launch when they fit (one padded row-major copy per
//     layer: read as b128 along rows forward, as 4 x b32 down columns in reverse); larger
//     networks read row fragments straight from L2 (plus a transposed copy for reverse) on
//     16-sample workgroups (one team of 8 waves), static layouts as ONE fragment stream per wave
//     that runs ahead across tile, sweep and evaluation boundaries (rhs_tile_stream[_jvp]);
//   - the Runge-Kutta state of the z rows sits in registers in the accumulator layout of the
//     lanes that produce zdot; the three scalar rows sit in LDS.
//
// Two layout flavours drive the same kernel template: RtLayout (every size a run-time
// value: any network whose images fit in LDS) and StLayoutX<...> / StLayoutJ<...> (sizes and
// activations are compile-time constants: LDS offsets become instruction immediates, layer loops
// unroll, no integer address arithmetic is left in the GEMM phases).
// The headline shape 32-128-128-32 has kernels of its own with the weight fragments resident in
// registers (cnf_step3.hip: k_step3, k_step3j); launch() below routes to them.
#include "cnf_mfma_dev.h"
#include "cnf_step3.h"

#include <cstdlib>
#include <type_traits>


// k_mfma geometry: 2 teams (one per 16-sample column tile) of MF_WPT = 4 waves: 512 lanes,
// 2 waves per SIMD, <= 256 VGPRs.
#define MF_WPT 4
#define MF_WPT_NARROW 8       // one team of 8 waves = 16-sample workgroups (networks that stream their weights from L2)
#define MF_KTHREADS (MF_WPT * 128)


// ---- layouts -------------------------------------------------------------------------------
__host__ __device__ constexpr int pad_to(int x, int residue, int modulus) {  // smallest y >= x, y % modulus == residue
    return x + ((residue - x) % modulus + modulus) % modulus;
}
__host__ __device__ constexpr int sw_of(int p_in) { return pad_to(p_in, 4, 16); }  // weight row stride
__host__ __device__ constexpr int sx_of(int p) { return pad_to(p, 8, 16); }        // activation row stride

// run-time layout: thin accessor wrapper over the plan's MfmaLayout
struct RtLayout {
    static constexpr bool kStatic = false;
    MfmaLayout m;
    __device__ __forceinline__ int L() const { return m.L; }
    __device__ __forceinline__ int P(int l) const { return m.P[l]; }
    __device__ __forceinline__ int act(int l) const { return m.acts[l]; }
    __device__ __forceinline__ int SW(int l) const { return m.SW[l]; }
    __device__ __forceinline__ int SX(int l) const { return m.SX[l]; }
    __device__ __forceinline__ int w_off(int l) const { return m.w_off[l]; }
    __device__ __forceinline__ int b_off(int l) const { return m.b_off[l]; }
    __device__ __forceinline__ int x_off(int l) const { return m.x_off[l]; }
    __device__ __forceinline__ int img_floats() const { return m.core_img; }
    __device__ __forceinline__ bool wlds() const { return m.wlds != 0; }
    __device__ __forceinline__ bool jvp() const { return m.jvp != 0; }
    __device__ __forceinline__ int tx_off(int l) const { return m.tx_off[l]; }
    __device__ __forceinline__ int SWT(int l) const { return m.SWT[l]; }
    __device__ __forceinline__ int wt_off(int l) const { return m.wt_off[l]; }
    __device__ __forceinline__ int eps_off() const { return m.eps_off; }
    __device__ __forceinline__ int du_off() const { return m.du_off; }
    __device__ __forceinline__ int red_off() const { return m.red_off; }
    __device__ __forceinline__ int total_floats() const { return m.total_floats; }
    __device__ __forceinline__ int bar_off() const { return m.total_floats - 16; }
    __device__ __forceinline__ int sc_off() const { return m.sc_off; }
    __device__ __forceinline__ int n_in() const { return m.n_in; }
    __device__ __forceinline__ int norm_z() const { return m.norm_z; }
    __device__ __forceinline__ int norm_j() const { return m.norm_j; }
};

// compile-time layout: same formulas as mfma_plan_init, evaluated by the compiler
template <bool WLDS, int ACT, int... PD>
struct StLayoutX {
    static constexpr bool kStatic = true;
    static constexpr int kL = sizeof...(PD) - 1;
    static constexpr int kNB = WLDS ? MF_NB : 16;      // samples per workgroup tile (mfma_plan_init: place_lds)
    int n_in_, norm_z_, norm_j_;
    __host__ __device__ static constexpr int pd(int l) { constexpr int a[] = {PD...}; return a[l]; }
    __host__ __device__ static constexpr int L() { return kL; }
    __host__ __device__ static constexpr int P(int l) { return pd(l); }
    __host__ __device__ static constexpr int act(int) { return ACT; }
    __host__ __device__ static constexpr int SW(int l) { return sw_of(pd(l)); }
    __host__ __device__ static constexpr int SX(int l) { return sx_of(pd(l)); }
    __host__ __device__ static constexpr int w_off(int l) {
        int off = 0;
        for (int i = 0; i < l; ++i) off += pd(i + 1) * sw_of(pd(i));
        return off;
    }
    __host__ __device__ static constexpr int b_off(int l) {
        int off = w_off(kL);
        for (int i = 0; i < l; ++i) off += pd(i + 1);
        return off;
    }
    __host__ __device__ static constexpr bool wlds() { return WLDS; }
    __host__ __device__ static constexpr bool jvp() { return false; }
    __host__ __device__ static constexpr int tx_off(int) { return 0; }
    __host__ __device__ static constexpr int SWT(int l) { return sw_of(pd(l + 1)); }
    __host__ __device__ static constexpr int wt_off(int l) {            // transposed images (WLDS == false)
        int off = (b_off(kL) + 3) & ~3;
        for (int i = 0; i < l; ++i) off += pd(i) * sw_of(pd(i + 1));
        return off;
    }
    __host__ __device__ static constexpr int img_floats() {
        return WLDS ? (b_off(kL) + 3) & ~3 : (wt_off(kL) + 3) & ~3;
    }
    __host__ __device__ static constexpr int x_off(int l) {
        int off = WLDS ? img_floats() : 0;
        for (int i = 0; i < l; ++i) off += kNB * sx_of(pd(i));
        return off;
    }
    __host__ __device__ static constexpr int eps_off() { return x_off(kL + 1); }
    __host__ __device__ static constexpr int du_off() { return eps_off() + kNB * sx_of(pd(0)); }
    __host__ __device__ static constexpr int red_off() { return du_off() + kNB * sx_of(pd(0)); }
    __host__ __device__ static constexpr int red_floats() {
        return 3 * (pd(0) / 16) * MF_NB < 256 ? 256 : 3 * (pd(0) / 16) * MF_NB;
    }
    __host__ __device__ static constexpr int sc_off() { return red_off() + red_floats(); }
    __host__ __device__ static constexpr int bar_off() { return sc_off() + MF_NB * 24; }
    __host__ __device__ static constexpr int total_floats() { return bar_off() + 16; }
    __device__ __forceinline__ int n_in() const { return n_in_; }
    __device__ __forceinline__ int norm_z() const { return norm_z_; }
    __device__ __forceinline__ int norm_j() const { return norm_j_; }
};

template <int ACT, int... PD>
using StLayout = StLayoutX<true, ACT, PD...>;

// streamed weights + tangent images (JVP compute mode): same formulas as mfma_plan_init with jvp = 1, wlds = 0
template <int ACT, int... PD>
struct StLayoutJ : StLayoutX<false, ACT, PD...> {
    using B_ = StLayoutX<false, ACT, PD...>;
    __host__ __device__ static constexpr bool jvp() { return true; }
    __host__ __device__ static constexpr int tx_off(int l) {           // tau_l, l = 1 .. L-1 (tau_0 is the eps image)
        int off = B_::du_off() + B_::kNB * sx_of(B_::pd(0));
        for (int i = 1; i < l; ++i) off += B_::kNB * sx_of(B_::pd(i));
        return off;
    }
    __host__ __device__ static constexpr int red_off() { return tx_off(B_::kL); }
    __host__ __device__ static constexpr int sc_off() { return red_off() + B_::red_floats(); }
    __host__ __device__ static constexpr int bar_off() { return sc_off() + MF_NB * 24; }
    __host__ __device__ static constexpr int total_floats() { return bar_off() + 16; }
};

// layer loops: unrolled with compile-time indices for static layouts, plain loops otherwise
template <int I, int N, class F>
__device__ __forceinline__ void static_for_up(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_up<I + 1, N>(f); }
}
template <int I, class F>
__device__ __forceinline__ void static_for_down(F&& f) {
    if constexpr (I >= 0) { f(std::integral_constant<int, I>{}); static_for_down<I - 1>(f); }
}
template <class LY, class F>
__device__ __forceinline__ void for_layers_up(const LY& ly, F&& f) {
    if constexpr (LY::kStatic) static_for_up<0, LY::kL>(f);
    else for (int l = 0; l < ly.L(); ++l) f(l);
}
template <class LY, class F>
__device__ __forceinline__ void for_layers_down(const LY& ly, F&& f) {
    if constexpr (LY::kStatic) static_for_down<LY::kL - 1>(f);
    else for (int l = ly.L() - 1; l >= 0; --l) f(l);
}

// ---- activation helpers ------------------------------------------------------------------
__device__ __forceinline__ void act_fast(int kind, float a, float& h, float& d) {
    if (kind == 1) { h = tanh_fast(a); d = fmaf(-h, h, 1.0f); }
    else cnf_act(kind, a, h, d);
}
__device__ __forceinline__ void act4(int kind, const f32x4& a, f32x4& h, f32x4& d) {
    float h0, h1, h2, h3, d0, d1, d2, d3;
    act_fast(kind, a.x, h0, d0); act_fast(kind, a.y, h1, d1);
    act_fast(kind, a.z, h2, d2); act_fast(kind, a.w, h3, d3);
    h = f32x4{h0, h1, h2, h3};
    d = f32x4{d0, d1, d2, d3};
}
__device__ __forceinline__ float d_from_h(int kind, float h) {
    switch (kind) {
        case 0: return 1.0f;
        case 1: return fmaf(-h, h, 1.0f);
        case 2: return h * (1.0f - h);
        case 3: return 1.0f - __expf(-h);            // softplus: sigma'(a) = 1 - exp(-softplus(a))
        case 4: return h > 0.0f ? 1.0f : 0.0f;
        default: return h > 0.0f ? 1.0f : h + 1.0f;  // elu
    }
}

__device__ __forceinline__ f32x4 mfma4(const f32x4& a, const f32x4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// GEMM bodies, fully unrolled over NU k-blocks (16 features each) as a rolling software
// pipeline: the LDS operands of k-block u+2 are requested right after the MFMAs of block u
// have been issued, so LDS traffic is spread over the phase (no burst of every wave's loads
// at the phase start) and each request has a whole block of MFMAs (256 cycles) to land.
//  NTL == 2: acc0/acc1 are two output tiles sharing the B operand;
//  NTL == 1: one output tile, k-steps alternate between acc0 and acc1 (two independent
//            MFMA chains; the caller adds them).
template <int NTL>
__device__ __forceinline__ void mfma_block(f32x4& acc0, f32x4& acc1, const f32x4& b, const f32x4& a0,
                                           const f32x4& a1) {
    if (NTL == 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], b[c], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b[c], acc1, 0, 0, 0);
        }
    } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[3], b[3], acc1, 0, 0, 0);
    }
}
// Forward:  xb = region_in + (16*sw + s)*SX + 4q;  wa = W + (16*ot + s)*SW + 4q
// AHEAD = how many k-blocks the operand requests run ahead of the MFMAs: 2 for weights in
// LDS, 4 for weights read straight from HBM/L2 (longer latency).
template <int NU, int NTL, int AHEAD = 2>
__device__ __forceinline__ void fwd_body(f32x4& acc0, f32x4& acc1, const float* xb,
                                         const float* wa0, const float* wa1) {
    constexpr int R = AHEAD + 1;   // ring: a prefetch never lands in registers that the
                                   // MFMAs issued just before it still have to read
    f32x4 b[R], a0[R], a1[R];
#pragma unroll
    for (int u = 0; u < AHEAD && u < NU; ++u) {
        b[u] = *(const f32x4*)(xb + 16 * u);
        a0[u] = *(const f32x4*)(wa0 + 16 * u);
        if (NTL == 2) a1[u] = *(const f32x4*)(wa1 + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<NTL>(acc0, acc1, b[u % R], a0[u % R], a1[u % R]);
        __builtin_amdgcn_sched_barrier(0);
        if (u + AHEAD < NU) {
            b[(u + AHEAD) % R] = *(const f32x4*)(xb + 16 * (u + AHEAD));
            a0[(u + AHEAD) % R] = *(const f32x4*)(wa0 + 16 * (u + AHEAD));
            if (NTL == 2) a1[(u + AHEAD) % R] = *(const f32x4*)(wa1 + 16 * (u + AHEAD));
        }
    }
}
// Reverse: Out[k][s] = sum_o W[o][k] G[o][s].
//  gb = region_{l+1} + (16*sw + s)*SX + 4q;  wc = W + (4q)*SW + 16*kt + s (walked down a column)
template <int NU, int NTL>
__device__ __forceinline__ void bwd_body(f32x4& acc0, f32x4& acc1, int SW, const float* gb,
                                         const float* wc0, const float* wc1) {
    f32x4 b[3], a0[3], a1[3];
#pragma unroll
    for (int u = 0; u < 2 && u < NU; ++u) {
        b[u] = *(const f32x4*)(gb + 16 * u);
        const float* p0 = wc0 + 16 * u * SW;
        a0[u] = f32x4{p0[0], p0[SW], p0[2 * SW], p0[3 * SW]};
        if (NTL == 2) {
            const float* p1 = wc1 + 16 * u * SW;
            a1[u] = f32x4{p1[0], p1[SW], p1[2 * SW], p1[3 * SW]};
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<NTL>(acc0, acc1, b[u % 3], a0[u % 3], a1[u % 3]);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < NU) {
            b[(u + 2) % 3] = *(const f32x4*)(gb + 16 * (u + 2));
            const float* p0 = wc0 + 16 * (u + 2) * SW;
            a0[(u + 2) % 3] = f32x4{p0[0], p0[SW], p0[2 * SW], p0[3 * SW]};
            if (NTL == 2) {
                const float* p1 = wc1 + 16 * (u + 2) * SW;
                a1[(u + 2) % 3] = f32x4{p1[0], p1[SW], p1[2 * SW], p1[3 * SW]};
            }
        }
    }
}

// run-time k-block count: blocks of 8, then an exact tail
template <int NTL, int AHEAD = 2>
__device__ __forceinline__ void gemm_fwd(f32x4& acc0, f32x4& acc1, int U, const float* xb,
                                         const float* wa0, const float* wa1) {
    while (U > 8) {
        fwd_body<8, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1);
        xb += 128; wa0 += 128; wa1 += 128; U -= 8;
    }
    switch (U) {
        case 1: fwd_body<1, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
        case 2: fwd_body<2, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
        case 3: fwd_body<3, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
        case 4: fwd_body<4, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
        case 5: fwd_body<5, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
        case 6: fwd_body<6, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
        case 7: fwd_body<7, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
        default: fwd_body<8, NTL, AHEAD>(acc0, acc1, xb, wa0, wa1); break;
    }
}
template <int NTL>
__device__ __forceinline__ void gemm_bwd(f32x4& acc0, f32x4& acc1, int U, int SW, const float* gb,
                                         const float* wc0, const float* wc1) {
    while (U > 8) {
        bwd_body<8, NTL>(acc0, acc1, SW, gb, wc0, wc1);
        gb += 128; wc0 += 128 * SW; wc1 += 128 * SW; U -= 8;
    }
    switch (U) {
        case 1: bwd_body<1, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
        case 2: bwd_body<2, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
        case 3: bwd_body<3, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
        case 4: bwd_body<4, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
        case 5: bwd_body<5, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
        case 6: bwd_body<6, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
        case 7: bwd_body<7, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
        default: bwd_body<8, NTL>(acc0, acc1, SW, gb, wc0, wc1); break;
    }
}

// Phase barrier of the workgroup (both teams).
__device__ __forceinline__ void phase_barrier() { __syncthreads(); }


// ---- epilogues ---------------------------------------------------------------------------------
// forward tile: bias + activation; hidden layers store h.  The last layer keeps zdot in
// registers (returned through zd: the lane that computes rows 4q..4q+3 of sample s also owns
// those rows of the Runge-Kutta state), stores g_L = eps .* sigma'_L to region_L and the
// |zdot|^2 partial of its 16 rows to RED[0].
template <class LY>
__device__ __forceinline__ void fwd_epilogue(const LY& ly, float* lds, const float* wimg, int l, bool last, int ot,
                                             f32x4 acc, int row, int q, f32x4& zd, bool test = false,
                                             const float* cbrow = nullptr) {
    const int r0 = 16 * ot + 4 * q;
    // conditional models: layer 0 takes this sample's bias W1[:, n_in:] ys + b1
    const f32x4 bv = (l == 0 && cbrow) ? *(const f32x4*)(cbrow + r0) : *(const f32x4*)(wimg + ly.b_off(l) + r0);
    const int act = ly.act(l);
    float h0, h1, h2, h3, d0, d1, d2, d3;
    act_fast(act, acc.x + bv.x, h0, d0);
    act_fast(act, acc.y + bv.y, h1, d1);
    act_fast(act, acc.z + bv.z, h2, d2);
    act_fast(act, acc.w + bv.w, h3, d3);
    float* out = lds + ly.x_off(l + 1) + row * ly.SX(l + 1) + r0;
    if (!last) {
        *(f32x4*)out = f32x4{h0, h1, h2, h3};
    } else {
        const int n_in = ly.n_in();
        const f32x4 ev = *(const f32x4*)(lds + ly.eps_off() + row * ly.SX(0) + r0);
        zd = f32x4{r0 + 0 < n_in ? h0 : 0.f, r0 + 1 < n_in ? h1 : 0.f,
                   r0 + 2 < n_in ? h2 : 0.f, r0 + 3 < n_in ? h3 : 0.f};
        // Hutchinson: g_L = eps .* sigma'_L; exact trace (2 layers): the image holds sigma'_L itself
        *(f32x4*)out = test ? f32x4{r0 + 0 < n_in ? d0 : 0.f, r0 + 1 < n_in ? d1 : 0.f, r0 + 2 < n_in ? d2 : 0.f,
                                    r0 + 3 < n_in ? d3 : 0.f}
                            : f32x4{ev.x * d0, ev.y * d1, ev.z * d2, ev.w * d3};
        const float e2 = quad_sum(zd.x * zd.x + zd.y * zd.y + zd.z * zd.z + zd.w * zd.w);
        if (q == 0) lds[ly.red_off() + ot * MF_NB + row] = e2;
    }
}
// reverse tile: hidden layers scale by sigma'(h) in place; layer 0 reduces eJ = W_1^T g_1 to
// the trace and norm partials (src/icnf.jl:334, :343)
template <class LY>
__device__ __forceinline__ void bwd_epilogue(const LY& ly, float* lds, int l, int kt, f32x4 acc, int row,
                                             int q) {
    const int r0 = 16 * kt + 4 * q;
    if (l > 0) {
        const int pact = ly.act(l - 1);
        float* out = lds + ly.x_off(l) + row * ly.SX(l) + r0;
        const f32x4 hv = *(const f32x4*)out;
        *(f32x4*)out = f32x4{acc.x * d_from_h(pact, hv.x), acc.y * d_from_h(pact, hv.y),
                             acc.z * d_from_h(pact, hv.z), acc.w * d_from_h(pact, hv.w)};
    } else {
        const int nt0 = ly.P(0) >> 4;
        const f32x4 ev = *(const f32x4*)(lds + ly.eps_off() + row * ly.SX(0) + r0);
        float ld = -(acc.x * ev.x + acc.y * ev.y + acc.z * ev.z + acc.w * ev.w);
        float n2 = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
        ld = quad_sum(ld);
        n2 = quad_sum(n2);
        if (q == 0) {
            lds[ly.red_off() + (nt0 + kt) * MF_NB + row] = ld;
            lds[ly.red_off() + (2 * nt0 + kt) * MF_NB + row] = n2;
        }
    }
}

// ---- the RHS on the tile resident in LDS ---------------------------------------------------
// In: region_0 holds z ([sample][feature]); EPS holds eps.  Out: zd0/zd1 = zdot of the (up
// to two) 16-row tiles this wave owns, RED = per-tile partial sums of
// (|zdot|^2, -eps.(J^T eps), |J^T eps|^2).  Ends with a barrier.
#ifdef MF_STAMPS
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                      stamps[i] += t_ - tlast; tlast = t_; } while (0)
#define STAMP_ARGS , unsigned long long* stamps, unsigned long long& tlast
#define STAMP_PASS , stamps, tlast
#else
#define STAMP(i) do {} while (0)
#define STAMP_ARGS
#define STAMP_PASS
#endif
template <int WPT, class LY, class F>
__device__ __forceinline__ void rhs_tile(const LY& ly, float* lds, const float* wimg, int lane, int wave,
                                         f32x4& zd0, f32x4& zd1,
                                         F&& after_zdot, const float* cimg, int SWC,
                                         const float* cbrow STAMP_ARGS) {
    // wimg: where the weight image is read from -- the LDS copy, or (networks too large for
    // LDS) the HBM/L2-resident image, with a row-major transposed copy for the reverse sweep
    // team = column tile; the feature-group index is rotated by 2 for team 1 so that the
    // narrow layers (fewer than 4 output tiles) of the two teams land on different SIMDs
    const int s = lane & 15, q = lane >> 4, team = wave / WPT, fg = (wave + (WPT / 2) * team) % WPT;
    const int row = 16 * team + s;
    if constexpr (!LY::kStatic) {
        if (ly.jvp() && !cimg) {
            // ---- forward-mode sweep (DIJacVecMatrixMode, src/icnf.jl:384-420): h_l and
            // tau_l = sigma'_l .* (W_l tau_{l-1}) side by side, tau_0 = eps; no reverse sweep.
            // Same row fragments of W_l against two B images.
            const int L = ly.L(), nt0 = ly.P(0) >> 4, n_in = ly.n_in();
            for (int l = 0; l < L; ++l) {
                const int ntiles = ly.P(l + 1) >> 4, SW = ly.SW(l), U = ly.P(l) >> 4;
                const float* xh = lds + ly.x_off(l) + row * ly.SX(l) + 4 * q;
                const float* xt = lds + (l == 0 ? ly.eps_off() : ly.tx_off(l)) + row * ly.SX(l) + 4 * q;
                const float* W = wimg + ly.w_off(l);
                const bool last = l == L - 1;
                const int act = ly.act(l);
                for (int t0 = fg; t0 < ntiles; t0 += 2 * WPT) {
                    const int t1 = t0 + WPT;
                    const bool two = ntiles > WPT && t1 < ntiles;   // constant false for layers of <= WPT tiles
                    f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = ah0, at0 = ah0, at1 = ah0;
                    const float* wa0 = W + (16 * t0 + s) * SW + 4 * q;
                    const float* wa1 = W + (16 * t1 + s) * SW + 4 * q;
                    if (ly.wlds()) {
                        if (two) { gemm_fwd<2>(ah0, ah1, U, xh, wa0, wa1); gemm_fwd<2>(at0, at1, U, xt, wa0, wa1); }
                        else { gemm_fwd<1>(ah0, ah1, U, xh, wa0, wa1); ah0 += ah1; gemm_fwd<1>(at0, at1, U, xt, wa0, wa1); at0 += at1; }
                    } else {
                        if (two) { gemm_fwd<2, 4>(ah0, ah1, U, xh, wa0, wa1); gemm_fwd<2, 4>(at0, at1, U, xt, wa0, wa1); }
                        else { gemm_fwd<1, 4>(ah0, ah1, U, xh, wa0, wa1); ah0 += ah1; gemm_fwd<1, 4>(at0, at1, U, xt, wa0, wa1); at0 += at1; }
                    }
                    for (int n = 0; n < (two ? 2 : 1); ++n) {
                        const int ot = n ? t1 : t0;
                        const f32x4 ah = n ? ah1 : ah0, at = n ? at1 : at0;
                        const int r0 = 16 * ot + 4 * q;
                        const f32x4 bv = (l == 0 && cbrow) ? *(const f32x4*)(cbrow + r0)
                                                           : *(const f32x4*)(wimg + ly.b_off(l) + r0);
                        f32x4 h, d;
                        act4(act, ah + bv, h, d);
                        const f32x4 tau = d * at;
                        if (!last) {
                            *(f32x4*)(lds + ly.x_off(l + 1) + row * ly.SX(l + 1) + r0) = h;
                            *(f32x4*)(lds + ly.tx_off(l + 1) + row * ly.SX(l + 1) + r0) = tau;
                        } else {
                            const f32x4 ev = *(const f32x4*)(lds + ly.eps_off() + row * ly.SX(0) + r0);
                            const f32x4 zd = {r0 + 0 < n_in ? h.x : 0.f, r0 + 1 < n_in ? h.y : 0.f,
                                              r0 + 2 < n_in ? h.z : 0.f, r0 + 3 < n_in ? h.w : 0.f};
                            const f32x4 tm = {r0 + 0 < n_in ? tau.x : 0.f, r0 + 1 < n_in ? tau.y : 0.f,
                                              r0 + 2 < n_in ? tau.z : 0.f, r0 + 3 < n_in ? tau.w : 0.f};
                            if (n) zd1 = zd; else zd0 = zd;
                            const float e2 = quad_sum(zd.x * zd.x + zd.y * zd.y + zd.z * zd.z + zd.w * zd.w);
                            const float ld = quad_sum(-(ev.x * tm.x + ev.y * tm.y + ev.z * tm.z + ev.w * tm.w));   // icnf.jl:404
                            const float n2 = quad_sum(tm.x * tm.x + tm.y * tm.y + tm.z * tm.z + tm.w * tm.w);      // icnf.jl:413
                            if (q == 0) {
                                lds[ly.red_off() + ot * MF_NB + row] = e2;
                                lds[ly.red_off() + (nt0 + ot) * MF_NB + row] = ld;
                                lds[ly.red_off() + (2 * nt0 + ot) * MF_NB + row] = n2;
                            }
                        }
                    }
                }
                if (last) after_zdot();
                phase_barrier();
            }
            return;
        }
    }
    // ---- forward ----
    for_layers_up(ly, [&](auto l) {
        const int ntiles = ly.P(l + 1) >> 4, SW = ly.SW(l);
        const float* xb = lds + ly.x_off(l) + row * ly.SX(l) + 4 * q;
        const float* W = wimg + ly.w_off(l);
        const bool last = l == ly.L() - 1;
        for (int t0 = fg; t0 < ntiles; t0 += 2 * WPT) {
            const int t1 = t0 + WPT;
            const bool two = ntiles > WPT && t1 < ntiles;   // constant false for layers of <= WPT tiles
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* wa0 = W + (16 * t0 + s) * SW + 4 * q;
            const float* wa1 = W + (16 * t1 + s) * SW + 4 * q;
            if constexpr (LY::kStatic) {
                constexpr int U = LY::P(decltype(l)::value) >> 4;
                constexpr int AH = LY::wlds() ? 2 : 4;
                if constexpr (U <= 8) {
                    if (two) fwd_body<U, 2, AH>(acc0, acc1, xb, wa0, wa1);
                    else { fwd_body<U, 1, AH>(acc0, acc1, xb, wa0, wa1); acc0 += acc1; }
                } else {       // long contractions: blocks of 8 k-blocks, not one giant unroll
                    if (two) gemm_fwd<2, AH>(acc0, acc1, U, xb, wa0, wa1);
                    else { gemm_fwd<1, AH>(acc0, acc1, U, xb, wa0, wa1); acc0 += acc1; }
                }
            } else {
                const int U = ly.P(l) >> 4;
                if (ly.wlds()) {
                    if (two) gemm_fwd<2>(acc0, acc1, U, xb, wa0, wa1);
                    else { gemm_fwd<1>(acc0, acc1, U, xb, wa0, wa1); acc0 += acc1; }
                } else {
                    if (two) gemm_fwd<2, 4>(acc0, acc1, U, xb, wa0, wa1);
                    else { gemm_fwd<1, 4>(acc0, acc1, U, xb, wa0, wa1); acc0 += acc1; }
                }
            }
            STAMP(16 + 3 * (int)l);
            fwd_epilogue(ly, lds, wimg, l, last, t0, acc0, row, q, zd0, cimg != nullptr, cbrow);
            if (two) fwd_epilogue(ly, lds, wimg, l, last, t1, acc1, row, q, zd1, cimg != nullptr, cbrow);
        }
        // zdot is known: the owner lanes can already form the NEXT stage state and put it
        // into region_0 (last read two barriers ago), which takes the stage combination and
        // one barrier off the critical path of every stage
        if (last) after_zdot();
        STAMP(17 + 3 * (int)l);
        phase_barrier();
        STAMP(18 + 3 * (int)l);
    });
    if (cimg) {
        // ---- exact trace of a 2-layer net (TestMode; src/icnf.jl:148-164 with utils.jl:1-36 in closed
        // form): tr J = sum_k sigma'_1[k] * (C sigma'_2)[k],  C = W_1 .* W_2^T.  One GEMM with the rows
        // of C (image in HBM/L2) against the sigma'_2 image, dotted with sigma'_1 recomputed from h_1.
        const int ntiles = ly.P(1) >> 4, U = ly.P(2) >> 4;
        const float* gb = lds + ly.x_off(2) + row * ly.SX(2) + 4 * q;
        const int pact = ly.act(0);
        float trp = 0.f;
        for (int t0 = fg; t0 < ntiles; t0 += 2 * WPT) {
            const int t1 = t0 + WPT;
            const bool two = ntiles > WPT && t1 < ntiles;   // constant false for layers of <= WPT tiles
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* ra = cimg + (16 * t0 + s) * SWC + 4 * q;
            const float* rb = cimg + (16 * t1 + s) * SWC + 4 * q;
            if (two) gemm_fwd<2, 4>(acc0, acc1, U, gb, ra, rb);
            else { gemm_fwd<1, 4>(acc0, acc1, U, gb, ra, rb); acc0 += acc1; }
            const f32x4 h0 = *(const f32x4*)(lds + ly.x_off(1) + row * ly.SX(1) + 16 * t0 + 4 * q);
            trp += acc0.x * d_from_h(pact, h0.x) + acc0.y * d_from_h(pact, h0.y) + acc0.z * d_from_h(pact, h0.z) +
                   acc0.w * d_from_h(pact, h0.w);
            if (two) {
                const f32x4 h1 = *(const f32x4*)(lds + ly.x_off(1) + row * ly.SX(1) + 16 * t1 + 4 * q);
                trp += acc1.x * d_from_h(pact, h1.x) + acc1.y * d_from_h(pact, h1.y) + acc1.z * d_from_h(pact, h1.z) +
                       acc1.w * d_from_h(pact, h1.w);
            }
        }
        trp = quad_sum(trp);
        if (q == 0) lds[ly.red_off() + fg * MF_NB + row] = -trp;      // one partial per wave of the team
        phase_barrier();
        return;
    }
    // ---- reverse (VJP): g_l = (W_{l+1}^T g_{l+1}) .* sigma'_l, in place over h_l ----
    for_layers_down(ly, [&](auto l) {
        const int ntiles = ly.P(l) >> 4, SW = ly.SW(l);
        const float* gb = lds + ly.x_off(l + 1) + row * ly.SX(l + 1) + 4 * q;
        const float* W = wimg + ly.w_off(l);
        for (int t0 = fg; t0 < ntiles; t0 += 2 * WPT) {
            const int t1 = t0 + WPT;
            const bool two = ntiles > WPT && t1 < ntiles;   // constant false for layers of <= WPT tiles
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* wc0 = W + (4 * q) * SW + 16 * t0 + s;
            const float* wc1 = W + (4 * q) * SW + 16 * t1 + s;
            if constexpr (LY::kStatic) {
                constexpr int U = LY::P(decltype(l)::value + 1) >> 4;
                if constexpr (LY::wlds()) {
                    if (two) bwd_body<U, 2>(acc0, acc1, SW, gb, wc0, wc1);
                    else { bwd_body<U, 1>(acc0, acc1, SW, gb, wc0, wc1); acc0 += acc1; }
                } else {
                    const float* WT = wimg + ly.wt_off(l);
                    const float* ra = WT + (16 * t0 + s) * ly.SWT(l) + 4 * q;
                    const float* rb = WT + (16 * t1 + s) * ly.SWT(l) + 4 * q;
                    if constexpr (U <= 8) {
                        if (two) fwd_body<U, 2, 4>(acc0, acc1, gb, ra, rb);
                        else { fwd_body<U, 1, 4>(acc0, acc1, gb, ra, rb); acc0 += acc1; }
                    } else {
                        if (two) gemm_fwd<2, 4>(acc0, acc1, U, gb, ra, rb);
                        else { gemm_fwd<1, 4>(acc0, acc1, U, gb, ra, rb); acc0 += acc1; }
                    }
                }
            } else {
                const int U = ly.P(l + 1) >> 4;
                if (ly.wlds()) {
                    if (two) gemm_bwd<2>(acc0, acc1, U, SW, gb, wc0, wc1);
                    else { gemm_bwd<1>(acc0, acc1, U, SW, gb, wc0, wc1); acc0 += acc1; }
                } else {
                    // rows of the transposed image: the same row-fragment GEMM as forward
                    const float* WT = wimg + ly.wt_off(l);
                    const float* ra = WT + (16 * t0 + s) * ly.SWT(l) + 4 * q;
                    const float* rb = WT + (16 * t1 + s) * ly.SWT(l) + 4 * q;
                    if (two) gemm_fwd<2, 4>(acc0, acc1, U, gb, ra, rb);
                    else { gemm_fwd<1, 4>(acc0, acc1, U, gb, ra, rb); acc0 += acc1; }
                }
            }
            STAMP(32 + 3 * (int)l);
            bwd_epilogue(ly, lds, l, t0, acc0, row, q);
            if (two) bwd_epilogue(ly, lds, l, t1, acc1, row, q);
        }
        STAMP(33 + 3 * (int)l);
        phase_barrier();
        STAMP(34 + 3 * (int)l);
    });
}

// ---- weight stream (static layouts whose weights stay in L2) --------------------------------------------
// Every wave's sequence of weight fragments is known at compile time and does not depend on data: tiles fg,
// fg+WPT, ... of every sweep, sweep after sweep, evaluation after evaluation.  So the fragments are fetched as ONE
// stream that runs MF_AH k-blocks ahead of the MFMAs ACROSS tile, sweep and evaluation boundaries: the last k-blocks
// of a tile request the first fragments of the wave's next tile (or of the next sweep's image), which then travel
// while the epilogue and the barrier run.  Without that every tile start exposes a full L2 round trip -- six per
// evaluation on BASELINE config 5.
#ifndef MF_AH
#define MF_AH 4
#endif
struct WPre { f32x4 a0[MF_AH], a1[MF_AH]; };

template <bool TWO>
__device__ __forceinline__ void wpre_load(WPre& p, const float* w0, const float* w1) {
#pragma unroll
    for (int j = 0; j < MF_AH; ++j) {
        p.a0[j] = *(const f32x4*)(w0 + 16 * j);
        if (TWO) p.a1[j] = *(const f32x4*)(w1 + 16 * j);
    }
}

// one tile (NTL == 1) or two tiles sharing the B operand (NTL == 2) of NU k-blocks; `pre` holds the first MF_AH
// fragments on entry and those of the next body (rows nw0 / nw1) on exit
template <int NU, int NTL, bool NEXT_TWO>
__device__ __forceinline__ void chain_body(f32x4& acc0, f32x4& acc1, const float* xb, const float* wa0,
                                           const float* wa1, WPre& pre, const float* nw0, const float* nw1) {
    static_assert(NU >= MF_AH, "a body must be at least as long as the prefetch distance");
    constexpr int R = MF_AH + 1;
    f32x4 a0[R], a1[R], b[3];
#pragma unroll
    for (int j = 0; j < MF_AH; ++j) { a0[j] = pre.a0[j]; if (NTL == 2) a1[j] = pre.a1[j]; }
    b[0] = *(const f32x4*)xb;
    if (NU > 1) b[1] = *(const f32x4*)(xb + 16);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        mfma_block<NTL>(acc0, acc1, b[u % 3], a0[u % R], a1[u % R]);
        __builtin_amdgcn_sched_barrier(0);
        if (u + MF_AH < NU) {
            a0[(u + MF_AH) % R] = *(const f32x4*)(wa0 + 16 * (u + MF_AH));
            if (NTL == 2) a1[(u + MF_AH) % R] = *(const f32x4*)(wa1 + 16 * (u + MF_AH));
        } else {
            const int j = u + MF_AH - NU;
            pre.a0[j] = *(const f32x4*)(nw0 + 16 * j);
            if (NEXT_TWO) pre.a1[j] = *(const f32x4*)(nw1 + 16 * j);
        }
        if (u + 2 < NU) b[(u + 2) % 3] = *(const f32x4*)(xb + 16 * (u + 2));
    }
}

// one sweep: this wave's NT tiles (fg, fg + WPT, ...) of a row-major image with NU k-blocks per row.
// pro(t) -> per-tile vector fetched BEFORE the MFMAs (bias); epi(t, acc, pro value).
template <int WPT, int NU, int NT, bool NEXT_TWO, class Pro, class Epi>
__device__ __forceinline__ void stream_sweep(const float* img, int SW, const float* xb, int fg, int s, int q,
                                             WPre& pre, const float* nw0, const float* nw1, Pro&& pro, Epi&& epi) {
    auto rowp = [&](int t) { return img + (16 * t + s) * SW + 4 * q; };
    static_for_up<0, (NT + 1) / 2>([&](auto bi) {
        constexpr int b = decltype(bi)::value;
        constexpr bool two = 2 * b + 1 < NT, has_next = 2 * b + 2 < NT, next_two_in = 2 * b + 3 < NT;
        const int t0 = fg + 2 * b * WPT, t1 = t0 + WPT;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 p0 = pro(t0), p1 = two ? pro(t1) : zero;
        f32x4 acc0 = zero, acc1 = zero;
        if constexpr (has_next)
            chain_body<NU, two ? 2 : 1, next_two_in>(acc0, acc1, xb, rowp(t0), rowp(t1), pre, rowp(t0 + 2 * WPT),
                                                     rowp(t0 + 3 * WPT));
        else
            chain_body<NU, two ? 2 : 1, NEXT_TWO>(acc0, acc1, xb, rowp(t0), rowp(t1), pre, nw0, nw1);
        if (!two) acc0 += acc1;
        epi(t0, acc0, p0);
        if (two) epi(t1, acc1, p1);
    });
}

// LDS-only workgroup barrier: __syncthreads() would also drain the weight fragments in flight
__device__ __forceinline__ void stream_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the same stream with TWO B operands per sample tile (state and tangent columns, JVP compute mode): 4 products per
// k-block when the wave has two tiles in flight
template <int NU, int NTL, bool NEXT_TWO>
__device__ __forceinline__ void chain_body2(f32x4 (&acc)[4], const float* xh, const float* xt, const float* wa0,
                                            const float* wa1, WPre& pre, const float* nw0, const float* nw1) {
    static_assert(NU >= MF_AH, "a body must be at least as long as the prefetch distance");
    constexpr int R = MF_AH + 1;
    f32x4 a0[R], a1[R], bh[3], bt[3];
#pragma unroll
    for (int j = 0; j < MF_AH; ++j) { a0[j] = pre.a0[j]; if (NTL == 2) a1[j] = pre.a1[j]; }
    bh[0] = *(const f32x4*)xh; bt[0] = *(const f32x4*)xt;
    if (NU > 1) { bh[1] = *(const f32x4*)(xh + 16); bt[1] = *(const f32x4*)(xt + 16); }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u % R][c], bh[u % 3][c], acc[0], 0, 0, 0);     // tile 0, state
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u % R][c], bt[u % 3][c], acc[1], 0, 0, 0);     // tile 0, tangent
            if (NTL == 2) {
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u % R][c], bh[u % 3][c], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u % R][c], bt[u % 3][c], acc[3], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u + MF_AH < NU) {
            a0[(u + MF_AH) % R] = *(const f32x4*)(wa0 + 16 * (u + MF_AH));
            if (NTL == 2) a1[(u + MF_AH) % R] = *(const f32x4*)(wa1 + 16 * (u + MF_AH));
        } else {
            const int j = u + MF_AH - NU;
            pre.a0[j] = *(const f32x4*)(nw0 + 16 * j);
            if (NEXT_TWO) pre.a1[j] = *(const f32x4*)(nw1 + 16 * j);
        }
        if (u + 2 < NU) { bh[(u + 2) % 3] = *(const f32x4*)(xh + 16 * (u + 2)); bt[(u + 2) % 3] = *(const f32x4*)(xt + 16 * (u + 2)); }
    }
}

// rhs_tile for the streamed layouts in the JVP compute mode (DIJacVecMatrixMode, src/icnf.jl:384-420): one forward sweep of
// h_l and tau_l = sigma'_l .* (W_l tau_{l-1}), tau_0 = eps; ldot = -eps.tau_L, ndot^2 = |tau_L|^2.  Same contract as rhs_tile.
template <int WPT, class LY, class F>
__device__ __forceinline__ void rhs_tile_stream_jvp(const LY& ly, float* lds, const float* wimg, int lane, int wave,
                                                    f32x4& zd0, f32x4& zd1, F&& after_zdot, const float* cbrow, WPre& pre) {
    constexpr int L = LY::kL;
    const int s = lane & 15, q = lane >> 4, team = wave / WPT, fg = (wave + (WPT / 2) * team) % WPT;
    const int row = 16 * team + s;
    const int n_in = ly.n_in(), nt0 = LY::P(0) >> 4;
    auto rowp = [&](const float* img, int SW, int t) { return img + (16 * t + s) * SW + 4 * q; };
    static_for_up<0, L>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr bool last = l == L - 1;
        constexpr int NU = LY::P(l) / 16, NT = LY::P(l + 1) / 16 / WPT;
        constexpr int ln = last ? 0 : l + 1;                               // the sweep that follows (next evaluation's first)
        constexpr bool NEXT_TWO = LY::P(ln + 1) / 16 / WPT >= 2;
        const float* img = wimg + LY::w_off(l);
        const float* nimg = wimg + LY::w_off(ln);
        const float* xh = lds + LY::x_off(l) + row * LY::SX(l) + 4 * q;
        const float* xt = lds + (l == 0 ? LY::eps_off() : LY::tx_off(l)) + row * LY::SX(l) + 4 * q;
        static_for_up<0, (NT + 1) / 2>([&](auto bi) {
            constexpr int b = decltype(bi)::value;
            constexpr bool two = 2 * b + 1 < NT, has_next = 2 * b + 2 < NT, next_two_in = 2 * b + 3 < NT;
            const int t0 = fg + 2 * b * WPT, t1 = t0 + WPT;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            auto biasv = [&](int t) {
                const int r0 = 16 * t + 4 * q;
                return (l == 0 && cbrow) ? *(const f32x4*)(cbrow + r0) : *(const f32x4*)(wimg + LY::b_off(l) + r0);
            };
            const f32x4 bv0 = biasv(t0), bv1 = two ? biasv(t1) : zero;
            f32x4 acc[4] = {zero, zero, zero, zero};
            if constexpr (has_next)
                chain_body2<NU, two ? 2 : 1, next_two_in>(acc, xh, xt, rowp(img, LY::SW(l), t0), rowp(img, LY::SW(l), t1), pre,
                                                          rowp(img, LY::SW(l), t0 + 2 * WPT), rowp(img, LY::SW(l), t0 + 3 * WPT));
            else
                chain_body2<NU, two ? 2 : 1, NEXT_TWO>(acc, xh, xt, rowp(img, LY::SW(l), t0), rowp(img, LY::SW(l), t1), pre,
                                                       rowp(nimg, LY::SW(ln), fg), rowp(nimg, LY::SW(ln), fg + WPT));
            auto epi = [&](int t, const f32x4& ah, const f32x4& at, const f32x4& bv) {
                const int r0 = 16 * t + 4 * q;
                f32x4 h, d;
                act4(LY::act(l), ah + bv, h, d);
                const f32x4 tau = d * at;
                if constexpr (!last) {
                    *(f32x4*)(lds + LY::x_off(l + 1) + row * LY::SX(l + 1) + r0) = h;
                    *(f32x4*)(lds + LY::tx_off(l + 1) + row * LY::SX(l + 1) + r0) = tau;
                } else {
                    const f32x4 ev = *(const f32x4*)(lds + LY::eps_off() + row * LY::SX(0) + r0);
                    const f32x4 zd = {r0 + 0 < n_in ? h.x : 0.f, r0 + 1 < n_in ? h.y : 0.f, r0 + 2 < n_in ? h.z : 0.f,
                                      r0 + 3 < n_in ? h.w : 0.f};
                    const f32x4 tm = {r0 + 0 < n_in ? tau.x : 0.f, r0 + 1 < n_in ? tau.y : 0.f, r0 + 2 < n_in ? tau.z : 0.f,
                                      r0 + 3 < n_in ? tau.w : 0.f};
                    if (t == fg) zd0 = zd; else zd1 = zd;
                    const float e2 = quad_sum(zd.x * zd.x + zd.y * zd.y + zd.z * zd.z + zd.w * zd.w);
                    const float ld = quad_sum(-(ev.x * tm.x + ev.y * tm.y + ev.z * tm.z + ev.w * tm.w));   // icnf.jl:404
                    const float n2 = quad_sum(tm.x * tm.x + tm.y * tm.y + tm.z * tm.z + tm.w * tm.w);      // icnf.jl:413
                    if (q == 0) {
                        lds[LY::red_off() + t * MF_NB + row] = e2;
                        lds[LY::red_off() + (nt0 + t) * MF_NB + row] = ld;
                        lds[LY::red_off() + (2 * nt0 + t) * MF_NB + row] = n2;
                    }
                }
            };
            epi(t0, acc[0], acc[1], bv0);
            if (two) epi(t1, acc[2], acc[3], bv1);
        });
        if (last) after_zdot();
        stream_barrier();
    });
}

template <class LY>
constexpr bool static_jvp() {
    if constexpr (LY::kStatic) return LY::jvp();
    else return false;
}

template <class LY, int WPT>
constexpr bool stream_ok() {
    if constexpr (!LY::kStatic) return false;
    else {
        if (LY::wlds()) return false;
        for (int l = 0; l <= LY::kL; ++l) if ((LY::P(l) / 16) % WPT) return false;
        for (int l = 0; l <= LY::kL; ++l) if (LY::P(l) / 16 < MF_AH) return false;
        return LY::P(LY::kL) / 16 <= 2 * WPT;
    }
}


// rhs_tile for the streamed layouts (same contract; `pre` carries the stream from call to call)
template <int WPT, class LY, class F>
__device__ __forceinline__ void rhs_tile_stream(const LY& ly, float* lds, const float* wimg, int lane, int wave,
                                                f32x4& zd0, f32x4& zd1, F&& after_zdot, const float* cimg, int SWC,
                                                const float* cbrow, WPre& pre) {
    constexpr int L = LY::kL;
    const int s = lane & 15, q = lane >> 4, team = wave / WPT, fg = (wave + (WPT / 2) * team) % WPT;
    const int row = 16 * team + s;
    const bool test = cimg != nullptr;
    // first rows of this wave in an image (tile fg and tile fg + WPT)
    auto first0 = [&](const float* img, int SW) { return img + (16 * fg + s) * SW + 4 * q; };
    auto first1 = [&](const float* img, int SW) { return img + (16 * (fg + WPT) + s) * SW + 4 * q; };
    // ---- forward ----
    static_for_up<0, L>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr bool last = l == L - 1;
        constexpr int NU = LY::P(l) / 16, NT = LY::P(l + 1) / 16 / WPT;
        // what follows this sweep: the next forward layer, or the first reverse sweep / the trace image
        constexpr int NT_next = last ? LY::P(L - 1) / 16 / WPT : LY::P(l + 2 > L ? L : l + 2) / 16 / WPT;
        const float* nimg = last ? (test ? cimg : wimg + LY::wt_off(L - 1)) : wimg + LY::w_off(last ? l : l + 1);
        const int nSW = last ? (test ? SWC : LY::SWT(L - 1)) : LY::SW(last ? l : l + 1);
        const float* xb = lds + LY::x_off(l) + row * LY::SX(l) + 4 * q;
        stream_sweep<WPT, NU, NT, (NT_next >= 2)>(wimg + LY::w_off(l), LY::SW(l), xb, fg, s, q, pre,
                                                  first0(nimg, nSW), first1(nimg, nSW),
            [&](int t) {
                const int r0 = 16 * t + 4 * q;
                return (l == 0 && cbrow) ? *(const f32x4*)(cbrow + r0) : *(const f32x4*)(wimg + LY::b_off(l) + r0);
            },
            [&](int t, const f32x4& acc, const f32x4& bv) {
                const int r0 = 16 * t + 4 * q;
                f32x4 h, d;
                act4(LY::act(l), acc + bv, h, d);
                float* out = lds + LY::x_off(l + 1) + row * LY::SX(l + 1) + r0;
                if constexpr (!last) *(f32x4*)out = h;
                else {
                    const int n_in = ly.n_in();
                    const f32x4 ev = *(const f32x4*)(lds + LY::eps_off() + row * LY::SX(0) + r0);
                    const f32x4 zd = {r0 + 0 < n_in ? h.x : 0.f, r0 + 1 < n_in ? h.y : 0.f, r0 + 2 < n_in ? h.z : 0.f,
                                      r0 + 3 < n_in ? h.w : 0.f};
                    if (t == fg) zd0 = zd; else zd1 = zd;
                    *(f32x4*)out = test ? f32x4{r0 + 0 < n_in ? d.x : 0.f, r0 + 1 < n_in ? d.y : 0.f,
                                                r0 + 2 < n_in ? d.z : 0.f, r0 + 3 < n_in ? d.w : 0.f}
                                        : ev * d;
                    const float e2 = quad_sum(zd.x * zd.x + zd.y * zd.y + zd.z * zd.z + zd.w * zd.w);
                    if (q == 0) lds[LY::red_off() + t * MF_NB + row] = e2;
                }
            });
        if (last) after_zdot();
        stream_barrier();
    });
    const float* w0first0 = first0(wimg + LY::w_off(0), LY::SW(0));
    const float* w0first1 = first1(wimg + LY::w_off(0), LY::SW(0));
    constexpr bool F0_TWO = LY::P(1) / 16 / WPT >= 2;
    if (test) {
        // ---- exact trace of a 2-layer net: tr J = sum_k sigma'_1[k] (C sigma'_2)[k], C = W_1 .* W_2^T (rows in L2)
        if constexpr (L == 2) {
            constexpr int NU = LY::P(2) / 16, NT = LY::P(1) / 16 / WPT;
            const float* gb = lds + LY::x_off(2) + row * LY::SX(2) + 4 * q;
            float trp = 0.f;
            stream_sweep<WPT, NU, NT, F0_TWO>(cimg, SWC, gb, fg, s, q, pre, w0first0, w0first1,
                [&](int) { return f32x4{0.f, 0.f, 0.f, 0.f}; },
                [&](int t, const f32x4& acc, const f32x4&) {
                    const f32x4 h = *(const f32x4*)(lds + LY::x_off(1) + row * LY::SX(1) + 16 * t + 4 * q);
                    trp += acc.x * d_from_h(LY::act(0), h.x) + acc.y * d_from_h(LY::act(0), h.y) +
                           acc.z * d_from_h(LY::act(0), h.z) + acc.w * d_from_h(LY::act(0), h.w);
                });
            trp = quad_sum(trp);
            if (q == 0) lds[LY::red_off() + fg * MF_NB + row] = -trp;
        }
        stream_barrier();
        return;
    }
    // ---- reverse (VJP): rows of the transposed images ----
    static_for_down<L - 1>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int NU = LY::P(l + 1) / 16, NT = LY::P(l) / 16 / WPT;
        constexpr int NT_next = l > 0 ? LY::P(l > 0 ? l - 1 : 0) / 16 / WPT : LY::P(1) / 16 / WPT;
        const float* nw0 = l > 0 ? first0(wimg + LY::wt_off(l > 0 ? l - 1 : 0), LY::SWT(l > 0 ? l - 1 : 0)) : w0first0;
        const float* nw1 = l > 0 ? first1(wimg + LY::wt_off(l > 0 ? l - 1 : 0), LY::SWT(l > 0 ? l - 1 : 0)) : w0first1;
        const float* gb = lds + LY::x_off(l + 1) + row * LY::SX(l + 1) + 4 * q;
        stream_sweep<WPT, NU, (NT < 1 ? 1 : NT), (NT_next >= 2)>(wimg + LY::wt_off(l), LY::SWT(l), gb, fg, s, q, pre, nw0, nw1,
            [&](int) { return f32x4{0.f, 0.f, 0.f, 0.f}; },
            [&](int t, const f32x4& acc, const f32x4&) { bwd_epilogue(ly, lds, l, t, acc, row, q); });
        stream_barrier();
    });
}

// One workgroup = one 32-sample tile (two teams of 16 samples).  The Runge-Kutta state is
// kept in registers in the MFMA accumulator layout: the lane that produces rows 4q..4q+3
// of zdot for sample s (wave fg owns the 16-row tiles fg and fg+4 of the n_in rows) holds
// u and k1..k7 of exactly those rows, so the stage combination needs no data movement and
// lands in region_0 with one ds_write_b128.  The three scalar rows (dlogp, E, n) of sample s
// live in lane s of the team's wave fg == 0; their RHS values are read back from the RED
// partials one barrier later, off the critical path.
// STEP = true: one Tsit5 step attempt (mode 2); STEP = false: one RHS evaluation (modes 0, 1).
// Two instantiations so that profiles name the step kernel and the plain RHS kernel apart.
// Streamed solve: block 0 copies the integrator state to pinned host memory after every controller run and
// then publishes the launch index; the host polls that word instead of waiting on events and copies.

template <class LY, bool STEP, int WPT = MF_WPT>
__global__ void __launch_bounds__(MF_KTHREADS, MF_KTHREADS / 256) k_mfma(LY ly, MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const StepState* st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsc = a.test ? 1 : 3;                      // scalar rows: dlogp (+ E, n in TrainMode)
    const int n_in = ly.n_in(), D = n_in + nsc;
    const int mode = STEP ? 2 : a.mode;
    if (st && st->done) {
        // keep the state chain intact for the launches queued behind this one
        if (STEP && a.apply_ctrl && blockIdx.x == 0 && tid == 0) { *a.st_out = *st; publish_mirror(a, *st); }
        return;
    }
    // streamed layouts: the first weight fragments of this wave set off before anything else
    constexpr bool STREAM = stream_ok<LY, WPT>();
    WPre pre;
    if constexpr (STREAM) {
        const int tm = wave / WPT, fgs = (wave + (WPT / 2) * tm) % WPT;
        const float* w0 = a.img + LY::w_off(0) + (16 * fgs + (lane & 15)) * LY::SW(0) + 4 * (lane >> 4);
        wpre_load<(LY::P(1) / 16 / WPT >= 2)>(pre, w0, w0 + 16 * WPT * LY::SW(0));
    }
    // error partials of the previous attempt: requested first, consumed after the image fill
    float cp0 = 0.f, cp1 = 0.f;
    if (STEP && a.apply_ctrl) {
        const int np = st->n_partials;
        for (int i = tid; i < np; i += MF_KTHREADS) { cp0 += a.partials_in[2 * i]; cp1 += a.partials_in[2 * i + 1]; }
    }

    // weights + biases -> LDS (once per workgroup; 4 x 16 B in flight per lane), the rest of
    // LDS zeroed (padding columns of the activation images meet zero weights but must be finite)
    const float* wimg = ly.wlds() ? (const float*)lds : a.img;
    {
        const int n = ly.wlds() ? ly.img_floats() : 0;
        constexpr int FL = 16;                    // float4 requests in flight per lane
        for (int base = 0; base < n; base += FL * MF_KTHREADS * 4) {
            f32x4 v[FL];
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                const int i = base + (j * MF_KTHREADS + tid) * 4;
                v[j] = *(const f32x4*)(a.img + (i < n ? i : 0));
            }
#pragma unroll
            for (int j = 0; j < FL; ++j) {
                const int i = base + (j * MF_KTHREADS + tid) * 4;
                if (i < n) *(f32x4*)(lds + i) = v[j];
            }
        }
        for (int z = n + tid * 4; z < ly.total_floats(); z += MF_KTHREADS * 4)
            *(f32x4*)(lds + z) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    int cur = 0, nacc = 0;
    float hstep = 0.f, abstol = 0.f, reltol = 0.f;
    // Runge-Kutta state of this lane's z rows (accumulator layout)
    f32x4 uz0 = {0.f, 0.f, 0.f, 0.f}, uz1 = uz0, kz0[7], kz1[7], un0 = uz0, un1 = uz0;
    if (STEP && a.apply_ctrl) {
        // In-kernel step controller: every workgroup reduces the same partials in the same
        // order and takes the same accept/reject decision; block 0 publishes the new state
        // for the next launch (kernel boundary = release/acquire).
        float* sc = lds + ly.bar_off() + 4;      // 12 spare words behind the team counters
        __syncthreads();                          // zero fill done before the scratch is used
        for (int off = 32; off > 0; off >>= 1) { cp0 += __shfl_down(cp0, off, 64); cp1 += __shfl_down(cp1, off, 64); }
        float* red = lds + ly.red_off();
        if (lane == 0) { red[wave] = cp0; red[16 + wave] = cp1; }
        __syncthreads();
        if (tid == 0) {
            float p0 = 0.f, p1 = 0.f;
            for (int w = 0; w < MF_KTHREADS / 64; ++w) { p0 += red[w]; p1 += red[16 + w]; }
            StepState ns = *st;
            ctrl_after_step(&ns, p0, p1, a.n_total);
            if (blockIdx.x == 0) { *a.st_out = ns; publish_mirror(a, ns); }
            sc[0] = __int_as_float(ns.cur); sc[1] = ns.h; sc[2] = ns.abstol; sc[3] = ns.reltol;
            sc[4] = __int_as_float(ns.done); sc[5] = __int_as_float(ns.naccept);
        }
        __syncthreads();
        cur = __float_as_int(sc[0]); hstep = sc[1]; abstol = sc[2]; reltol = sc[3];
        nacc = __float_as_int(sc[5]);
        if (__float_as_int(sc[4])) return;        // the controller just finished the solve
        __syncthreads();                          // scratch (RED) is free again
    } else if (st) {
        cur = st->cur; hstep = st->h; abstol = st->abstol; reltol = st->reltol; nacc = st->naccept;
    }
    // where this attempt files its stage states (gradient path), if anywhere
    float* dumpb = a.dump;
    if (a.dump && a.dump_step_stride) {
        dumpb = nacc < a.dump_cap ? a.dump + (size_t)nacc * a.dump_step_stride : nullptr;
        if (dumpb && blockIdx.x == 0 && tid == 0) a.hs_out[nacc] = hstep;
    }

    constexpr int TNB = 16, NBT = 16 * (MF_KTHREADS / 64 / WPT);   // samples per team / per workgroup tile
    const int team = wave / WPT;
    const int s = lane & 15, q = lane >> 4, fg = (wave + (WPT / 2) * team) % WPT;
    const int nt0 = ly.P(0) >> 4;
    const bool own0 = fg < nt0, own1 = nt0 > WPT && fg + WPT < nt0;   // z-row tiles fg, fg+WPT
    const bool sown = fg == 0 && q == 0;                  // scalar rows of sample s
    const int r00 = 16 * fg + 4 * q, r01 = r00 + 16 * WPT;   // first owned row of each tile
    const int nv0 = own0 ? n_in - r00 : 0, nv1 = own1 ? n_in - r01 : 0;   // valid rows (may be <= 0 or > 4)
    const int row = TNB * team + s;
    // (explicit selects: indexing the kernel-argument arrays with a run-time value makes the compiler fetch the
    // pointer from the argument segment through a vector load -- a memory round trip in front of the state loads)
    const float* Uin = mode == 0 ? a.u : (cur ? a.U[1] : a.U[0]);
    const float* K1in = mode == 0 ? nullptr : (cur ? a.K1[1] : a.K1[0]);
    float errsum = 0.f, badcnt = 0.f;
#ifdef MF_STAMPS
    unsigned long long stamps[48] = {0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tlast;
#endif
    __syncthreads();       // image filled, counters zeroed
    STAMP(15);

    const int ntile = (a.B + NBT - 1) / NBT;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b0 = tile * NBT + TNB * team;                 // first sample of this team
        const int nvalid = max(0, min(TNB, a.B - b0));
        const bool live = s < nvalid;                             // this lane's sample exists
        const size_t gcol = (size_t)(b0 + s) * D;
        const float* cbrow = (a.cond && live) ? a.cond + (size_t)(b0 + s) * a.cbs : nullptr;
        // eps tile -> EPS[sample][feature]: the lanes that own the state rows also fetch the eps
        // rows (same addresses pattern; previous readers are this team's waves, ordered by the
        // last team barrier of the previous tile)
        float* sc = lds + ly.sc_off() + row * 24;
        auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
        auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
        {
            // every global read of the prologue is issued before any of them is consumed
            const float* safe = a.img;
            const float* ep = a.eps ? a.eps + (size_t)(b0 + s) * n_in : nullptr;
            const int ce0 = (own0 && ep && live) ? nv0 : 0, ce1 = (own1 && ep && live) ? nv1 : 0;
            const int cu0 = (own0 && live) ? nv0 : 0, cu1 = (own1 && live) ? nv1 : 0;
            const int ck0 = K1in ? cu0 : 0, ck1 = K1in ? cu1 : 0;
            const int cs0 = (sown && live) ? nsc : 0, cs1 = K1in ? cs0 : 0;
            const f32x4 re0 = ld4_issue(ep + r00, ce0, safe), re1 = ld4_issue(ep + r01, ce1, safe);
            const f32x4 ru0 = ld4_issue(Uin + gcol + r00, cu0, safe), ru1 = ld4_issue(Uin + gcol + r01, cu1, safe);
            const f32x4 rk0 = ld4_issue(K1in + gcol + r00, ck0, safe), rk1 = ld4_issue(K1in + gcol + r01, ck1, safe);
            const f32x4 rs0 = ld4_issue(Uin + gcol + n_in, cs0, safe), rs1 = ld4_issue(K1in + gcol + n_in, cs1, safe);
            if (own0) *(f32x4*)(lds + ly.eps_off() + row * ly.SX(0) + r00) = ld4_mask(re0, ce0);
            if (own1) *(f32x4*)(lds + ly.eps_off() + row * ly.SX(0) + r01) = ld4_mask(re1, ce1);
            {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 7; ++i) { kz0[i] = zero; kz1[i] = zero; }
                uz0 = ld4_mask(ru0, cu0); uz1 = ld4_mask(ru1, cu1);
                kz0[0] = ld4_mask(rk0, ck0); kz1[0] = ld4_mask(rk1, ck1);
                if (sown) { sc_set(0, ld4_mask(rs0, cs0)); sc_set(1, ld4_mask(rs1, cs1)); }
                if (mode == 2 && dumpb && a.dump_step_stride && live) {      // u_n of this step, one array before its stage states
                    float* un_slot = dumpb - a.dump_stride + gcol;
                    if (own0) st4(un_slot + r00, uz0, nv0);
                    if (own1) st4(un_slot + r01, uz1, nv1);
                }
            }
        }
        const int nstage = mode == 2 ? 6 : 1;
        un0 = uz0; un1 = uz1;
        // state of stage `stg` (z rows) -> region_0; the last stage's state is u_new (a7 = b)
        auto put_stage = [&](int stg) {
            // evaluation stg (1..6) runs at the Runge-Kutta stage state U_{stg+1}; U_2..U_6 are filed, U_7 = u_new is not
            float* dmp = (mode == 2 && dumpb && stg <= 5 && live) ? dumpb + (size_t)(stg - 1) * a.dump_stride + gcol : nullptr;
            if (own0) {
                if (mode == 1) un0 = uz0 + hstep * kz0[0];
                else if (mode == 2) un0 = uz0 + hstep * stage_acc4_rt(stg, kz0);
                *(f32x4*)(lds + ly.x_off(0) + row * ly.SX(0) + r00) = un0;
                if (dmp) st4(dmp + r00, un0, nv0);
            }
            if (own1) {
                if (mode == 1) un1 = uz1 + hstep * kz1[0];
                else if (mode == 2) un1 = uz1 + hstep * stage_acc4_rt(stg, kz1);
                *(f32x4*)(lds + ly.x_off(0) + row * ly.SX(0) + r01) = un1;
                if (dmp) st4(dmp + r01, un1, nv1);
            }
        };
        auto read_scalars = [&]() {
            if (a.test) {       // exact trace: 4 per-wave partials of -tr J
                float ld = 0.f;
                for (int w = 0; w < WPT; ++w) ld += lds[ly.red_off() + w * MF_NB + row];
                return f32x4{ld, 0.f, 0.f, 0.f};
            }
            float ld = 0.f, e2 = 0.f, n2 = 0.f;
            for (int t = 0; t < nt0; ++t) {
                e2 += lds[ly.red_off() + t * MF_NB + row];
                ld += lds[ly.red_off() + (nt0 + t) * MF_NB + row];
                n2 += lds[ly.red_off() + (2 * nt0 + t) * MF_NB + row];
            }
            return f32x4{ld, ly.norm_z() ? __builtin_sqrtf(e2) : 0.f, ly.norm_j() ? __builtin_sqrtf(n2) : 0.f, 0.f};
        };
        put_stage(1);
        phase_barrier();
        STAMP(0);
        for (int stg = 1; stg <= nstage; ++stg) {
            // scalar rows of the PREVIOUS evaluation, from its RED partials (RED[0] is
            // rewritten only in this evaluation's last forward epilogue, two barriers on)
            if (stg > 1 && sown) sc_set(stg, read_scalars());        // slot j holds k_j
            f32x4 zd0 = {0.f, 0.f, 0.f, 0.f}, zd1 = zd0;
            auto after_zdot = [&]() {
                if (mode == 2) {
                    set_k(kz0, stg, zd0); set_k(kz1, stg, zd1);
                    if (stg < nstage) put_stage(stg + 1);
                } else { kz0[1] = zd0; kz1[1] = zd1; }
            };
            if constexpr (STREAM && static_jvp<LY>())
                rhs_tile_stream_jvp<WPT>(ly, lds, wimg, lane, wave, zd0, zd1, after_zdot, cbrow, pre);
            else if constexpr (STREAM)
                rhs_tile_stream<WPT>(ly, lds, wimg, lane, wave, zd0, zd1, after_zdot, a.test ? a.cimg : nullptr, a.SWC,
                                     cbrow, pre);
            else
                rhs_tile<WPT>(ly, lds, wimg, lane, wave, zd0, zd1, after_zdot, a.test ? a.cimg : nullptr, a.SWC,
                              cbrow STAMP_PASS);
        }
        // scalar rows of the last evaluation (rhs_tile ended with a barrier)
        if (sown) {
            sc_set(mode == 2 ? 7 : 2, read_scalars());
        }
        // ---- outputs ----
        if (live) {
            if (mode == 0 || mode == 1) {
                float* out = (mode == 0 ? a.du : a.Ks0) + gcol;
                if (own0) st4(out + r00, kz0[1], nv0);
                if (own1) st4(out + r01, kz1[1], nv1);
                if (sown) st4(out + n_in, sc_get(2), nsc);
                if (a.init_phase >= 0) {       // norms of the automatic initial dt, over the rows this lane owns
                    auto acc = [&](const f32x4& u4, const f32x4& f0, const f32x4& f1, int nvalid) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (c < nvalid) {
                                const float sk = fmaf(fabsf(u4[c]), reltol, abstol);
                                if (a.init_phase == 0) {
                                    const float x = u4[c] / sk, y = f1[c] / sk;
                                    errsum = fmaf(x, x, errsum); badcnt = fmaf(y, y, badcnt);
                                } else {
                                    const float x = (f1[c] - f0[c]) / sk;
                                    errsum = fmaf(x, x, errsum);
                                }
                            }
                        }
                    };
                    if (own0) acc(uz0, kz0[0], kz0[1], nv0);
                    if (own1) acc(uz1, kz1[0], kz1[1], nv1);
                    if (sown) acc(sc_get(0), sc_get(1), sc_get(2), nsc);
                }
            } else {
                float* Un = (cur ? a.U[0] : a.U[1]) + gcol;
                float* K7 = (cur ? a.K1[0] : a.K1[1]) + gcol;
                if (own0) {
                    st4(Un + r00, un0, nv0); st4(K7 + r00, kz0[6], nv0);
                    err_acc(errsum, badcnt, kz0, uz0, un0, hstep, abstol, reltol, nv0);
                }
                if (own1) {
                    st4(Un + r01, un1, nv1); st4(K7 + r01, kz1[6], nv1);
                    err_acc(errsum, badcnt, kz1, uz1, un1, hstep, abstol, reltol, nv1);
                }
                if (sown) {
                    f32x4 ks[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) ks[j] = sc_get(1 + j);
                    const f32x4 us = sc_get(0);
                    const f32x4 uns = us + hstep * stage_acc4<6>(ks);
                    st4(Un + n_in, uns, nsc); st4(K7 + n_in, ks[6], nsc);
                    err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, nsc);
                }
            }
        }
        // this team's RED / EPS reads of this tile precede its next-tile writes
        phase_barrier();
    }
#ifdef MF_STAMPS
    STAMP(14);
    if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 2 || wave == 5) && mode == 2) {
        printf("wave %d total %llu pre %llu elem %llu tail %llu | F1 %llu %llu %llu | F2 %llu %llu %llu | F3 %llu %llu %llu | B3 %llu %llu %llu | B2 %llu %llu %llu | B1 %llu %llu %llu\n",
               wave, tlast - tstart, stamps[15], stamps[0], stamps[14], stamps[16], stamps[17], stamps[18], stamps[19], stamps[20],
               stamps[21], stamps[22], stamps[23], stamps[24], stamps[38], stamps[39], stamps[40], stamps[35], stamps[36],
               stamps[37], stamps[32], stamps[33], stamps[34]);
    }
#endif
    if (mode == 2 || a.init_phase >= 0) {
        // deterministic block reduction of the error partial (fixed tree, fixed order)
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) {
            errsum += __shfl_down(errsum, off, 64);
            badcnt += __shfl_down(badcnt, off, 64);
        }
        if (lane == 0) { lds[ly.red_off() + wave] = errsum; lds[ly.red_off() + 16 + wave] = badcnt; }
        __syncthreads();
        if (tid == 0) {
            float e = 0.f, b = 0.f;
            for (int w = 0; w < MF_KTHREADS / 64; ++w) { e += lds[ly.red_off() + w]; b += lds[ly.red_off() + 16 + w]; }
            float* pout = a.partials;
            if (mode != 2) {
                // initial-dt phase: partials through agent-scope atomics, then a ticket; whoever draws the last
                // one sums all partials (fixed order) and runs the controller phase -- no separate launches
                __hip_atomic_store(pout + 2 * blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pout + 2 * blockIdx.x + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lds[ly.red_off() + 40] = (tk == gridDim.x - 1) ? 1.f : 0.f;
                if (tk == gridDim.x - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                pout[2 * blockIdx.x] = e;
                pout[2 * blockIdx.x + 1] = b;
            }
        }
    }
    if (mode != 2 && a.init_phase >= 0) {
        __syncthreads();
        if (lds[ly.red_off() + 40] != 0.f) {         // this workgroup drew the last ticket: all its threads reduce
            const float* pin = a.partials;
            float q0 = 0.f, q1 = 0.f;
            for (int i = tid; i < (int)gridDim.x; i += MF_KTHREADS) {
                q0 += __hip_atomic_load(pin + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q1 += __hip_atomic_load(pin + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int off = 32; off > 0; off >>= 1) { q0 += __shfl_down(q0, off, 64); q1 += __shfl_down(q1, off, 64); }
            if (lane == 0) { lds[ly.red_off() + wave] = q0; lds[ly.red_off() + 16 + wave] = q1; }
            __syncthreads();
            if (tid == 0) {
                float p0 = 0.f, p1 = 0.f;
                for (int w = 0; w < MF_KTHREADS / 64; ++w) { p0 += lds[ly.red_off() + w]; p1 += lds[ly.red_off() + 16 + w]; }
                ctrl_phase(a.st_out, a.init_phase, p0, p1, a.n_total);
            }
        }
    }
}

// ---- weight image packing -------------------------------------------------------------------
__global__ void k_pack_image(MfmaLayout ly, NetDesc nd, const float* __restrict__ P,
                             float* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ly.img_floats) return;
    float v = 0.f;
    for (int l = 0; l < ly.L; ++l) {
        const int wsz = ly.P[l + 1] * ly.SW[l];
        if (i >= ly.w_off[l] && i < ly.w_off[l] + wsz) {
            const int o = (i - ly.w_off[l]) / ly.SW[l], k = (i - ly.w_off[l]) % ly.SW[l];
            if (o < nd.dims[l + 1] && k < nd.dims[l]) v = P[nd.w_off[l] + o + (size_t)k * nd.dims[l + 1]];
        }
        if (i >= ly.b_off[l] && i < ly.b_off[l] + ly.P[l + 1]) {
            const int o = i - ly.b_off[l];
            if (o < nd.dims[l + 1]) v = P[nd.b_off[l] + o];
        }
        if (!ly.wlds && i >= ly.wt_off[l] && i < ly.wt_off[l] + ly.P[l] * ly.SWT[l]) {
            const int k = (i - ly.wt_off[l]) / ly.SWT[l], o = (i - ly.wt_off[l]) % ly.SWT[l];
            if (o < nd.dims[l + 1] && k < nd.dims[l]) v = P[nd.w_off[l] + o + (size_t)k * nd.dims[l + 1]];
        }
    }
    if (ly.c_off >= 0 && i >= ly.c_off && i < ly.c_off + ly.P[1] * ly.SWC) {
        // C[k][i] = W_1[k][i] * W_2[i][k]   (layer 0: out k, in i; layer 1: out i, in k)
        const int k = (i - ly.c_off) / ly.SWC, c = (i - ly.c_off) % ly.SWC;
        if (k < nd.dims[1] && c < nd.dims[0])
            v = P[nd.w_off[0] + k + (size_t)c * nd.dims[1]] * P[nd.w_off[1] + c + (size_t)k * nd.dims[2]];
    }
    img[i] = v;
}

// ---- host side ----------------------------------------------------------------------------
// static instantiations: (activation, padded sizes...) -> kernel.  Variant ids >= 2.
using LyCfg3 = StLayout<CNF_ACT_TANH, 32, 128, 128, 32>;   // BASELINE configs 3/4
using LyCfg2 = StLayout<CNF_ACT_TANH, 16, 48, 16>;         // BASELINE config 2
using LyCfg1 = StLayout<CNF_ACT_TANH, 16, 16, 16>;         // BASELINE config 1 (2->6->2 padded)
using LyCfg5 = StLayoutX<false, CNF_ACT_TANH, 128, 384, 128>;   // BASELINE config 5: weights stay in HBM/L2
using LyCfg5J = StLayoutJ<CNF_ACT_TANH, 128, 384, 128>;         // the same in the JVP compute mode (tangent images)

template <class LY>
static bool matches(const MfmaLayout& m) {
    if (m.L != LY::kL) return false;
    for (int l = 0; l <= m.L; ++l)
        if (m.P[l] != LY::P(l)) return false;
    for (int l = 0; l < m.L; ++l)
        if (m.acts[l] != LY::act(l)) return false;
    return (m.wlds != 0) == LY::wlds() && m.core_img == LY::img_floats() && m.total_floats == LY::total_floats() &&
           m.red_off == LY::red_off() && m.x_off[m.L] == LY::x_off(LY::kL);
}

void mfma_plan_init(MfmaPlan& p, const NetDesc& nd) {
    p.variant = 0;
    p.shape3 = false;
    MfmaLayout& ly = p.ly;
    ly = MfmaLayout{};
    ly.L = nd.n_layers;
    for (int l = 0; l <= nd.n_layers; ++l) {
        ly.dims[l] = nd.dims[l];
        ly.P[l] = (nd.dims[l] + 15) & ~15;
    }
    int off = 0;
    for (int l = 0; l < nd.n_layers; ++l) {
        ly.acts[l] = nd.acts[l];
        if (nd.acts[l] == CNF_ACT_SWISH && l < nd.n_layers - 1) return;  // sigma' not recoverable from h
        ly.SW[l] = sw_of(ly.P[l]);
        ly.w_off[l] = off;
        off += ly.P[l + 1] * ly.SW[l];
    }
    for (int l = 0; l < nd.n_layers; ++l) { ly.b_off[l] = off; off += ly.P[l + 1]; }
    ly.img_floats = (off + 3) & ~3;
    ly.wlds = 1;
    ly.jvp = nd.jvp;
    // nb = samples per workgroup tile: 32 (two teams) with the weights in LDS, 16 (one team of 8 waves) without
    auto place_lds = [&](int start, int nb) {
        int o = start;
        for (int l = 0; l <= nd.n_layers; ++l) {
            ly.SX[l] = sx_of(ly.P[l]);
            ly.x_off[l] = o;
            o += nb * ly.SX[l];
        }
        ly.eps_off = o; o += nb * ly.SX[0];
        ly.du_off = o;  o += nb * ly.SX[0];
        if (ly.jvp)
            for (int l = 1; l < nd.n_layers; ++l) { ly.tx_off[l] = o; o += nb * ly.SX[l]; }
        ly.red_off = o;
        int red = 3 * (ly.P[0] >> 4) * MF_NB;
        o += red < 256 ? 256 : red;
        ly.sc_off = o; o += MF_NB * 24;   // scalar-row state
        o += 16;                      // team-barrier counters / controller scratch
        ly.total_floats = o;
    };
    place_lds(ly.img_floats, MF_NB);
    if ((size_t)ly.total_floats * sizeof(float) > MF_LDS_BYTES) {
        // weights do not fit in LDS next to the activation images: leave them in HBM/L2 and add
        // a row-major transposed copy per layer for the reverse sweep
        ly.wlds = 0;
        int o = ly.img_floats;
        for (int l = 0; l < nd.n_layers; ++l) {
            ly.SWT[l] = sw_of(ly.P[l + 1]);
            ly.wt_off[l] = o;
            o += ly.P[l] * ly.SWT[l];
        }
        ly.img_floats = (o + 3) & ~3;
        place_lds(0, 16);
    }
    ly.core_img = ly.img_floats;
    ly.c_off = -1; ly.SWC = 0;
    if (nd.n_layers == 2 && nd.dims[0] == nd.dims[2]) {     // exact-trace image, read from HBM/L2 in TestMode
        ly.SWC = sw_of(ly.P[2]);
        ly.c_off = ly.img_floats;
        ly.img_floats = (ly.c_off + ly.P[1] * ly.SWC + 3) & ~3;
    }
    ly.n_in = nd.n_in;
    ly.norm_z = nd.norm_z;
    ly.norm_j = nd.norm_j;
    ly.ept = 0;
    if ((size_t)ly.total_floats * sizeof(float) > MF_LDS_BYTES) return;   // activations alone exceed LDS
    if (ly.P[0] > 128) return;                                             // state tiles fg, fg+4 only
    p.variant = 1;
    p.shape3 = ly.L == 3 && ly.P[0] == 32 && ly.P[1] == 128 && ly.P[2] == 128 && ly.P[3] == 32 &&
               ly.acts[0] == CNF_ACT_TANH && ly.acts[1] == CNF_ACT_TANH && ly.acts[2] == CNF_ACT_TANH;
    if (nd.jvp) {                        // forward-mode sweep: k_step3j for the headline shape, the JVP fragment stream for
        if (matches<LyCfg5J>(ly)) p.variant = 6;     // config 5's, else the run-time-layout kernel
        return;
    }
    if (matches<LyCfg3>(ly)) p.variant = 2;
    else if (matches<LyCfg2>(ly)) p.variant = 3;
    else if (matches<LyCfg1>(ly)) p.variant = 4;
    else if (matches<LyCfg5>(ly)) p.variant = 5;
}

void mfma_plan_free(MfmaPlan& p) {
    if (p.d_img) (void)hipFree(p.d_img);
    if (p.d_img3) (void)hipFree(p.d_img3);
    if (p.d_idle) (void)hipFree(p.d_idle);
    if (p.d_img3b) (void)hipFree(p.d_img3b);
    p.d_idle = nullptr;
    p.d_img3b = nullptr;
    p.d_img = nullptr;
    p.d_img3 = nullptr;
}

template <class LY, int WPT = MF_WPT>
static hipError_t set_attr() {
    hipError_t e = hipFuncSetAttribute((const void*)k_mfma<LY, true, WPT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       MF_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_mfma<LY, false, WPT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               MF_LDS_BYTES);
}

cnf_status mfma_plan_pack(MfmaPlan& p, const NetDesc& nd, const float* d_params, hipStream_t s) {
    if (!p.variant) return CNF_OK;
    if (!p.d_img) {
        if (hipMalloc(&p.d_img, (size_t)p.ly.img_floats * sizeof(float)) != hipSuccess) return CNF_ERR_HIP;
        hipError_t e = set_attr<RtLayout>();
        if (e == hipSuccess) e = set_attr<LyCfg3>();
        if (e == hipSuccess) e = set_attr<LyCfg2>();
        if (e == hipSuccess) e = set_attr<LyCfg1>();
        if (e == hipSuccess) e = set_attr<LyCfg5, MF_WPT_NARROW>();
        if (e == hipSuccess) e = set_attr<LyCfg5J, MF_WPT_NARROW>();
        if (e == hipSuccess) e = set_attr<RtLayout, MF_WPT_NARROW>();
        if (e != hipSuccess) return CNF_ERR_HIP;
    }
    hipLaunchKernelGGL(k_pack_image, dim3((p.ly.img_floats + 255) / 256), dim3(256), 0, s, p.ly, nd,
                       d_params, p.d_img);
    if (p.variant == 2 || p.shape3) {       // headline shape: register-fragment image of k_step3 / k_step3j
        if (!p.d_img3 && hipMalloc(&p.d_img3, step3_img_floats() * sizeof(float)) != hipSuccess) return CNF_ERR_HIP;
        if (!p.d_idle) {
            if (hipMalloc(&p.d_idle, sizeof(StepState)) != hipSuccess) return CNF_ERR_HIP;
            if (hipMemsetAsync(p.d_idle, 0, sizeof(StepState), s) != hipSuccess) return CNF_ERR_HIP;
        }
        step3_pack(nd, d_params, p.d_img3, s);
        if (!p.d_img3b && hipMalloc(&p.d_img3b, step3b_img_bytes()) != hipSuccess) return CNF_ERR_HIP;
        step3b_pack(nd, d_params, p.d_img3b, s);
    }
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

bool mfma_supported(const MfmaPlan& p, const NetDesc&, bool train, int) {
    if (p.variant == 0) return false;
    return train || p.ly.c_off >= 0;     // TestMode: exact trace in closed form for 2-layer nets
}

// CNF_STEP_V1=1: the first-generation step kernel (k_mfma) for the headline shape too -- A/B measurements only
static bool step_v1() {
    static const bool v = [] { const char* e = getenv("CNF_STEP_V1"); return e && e[0] == '1'; }();
    return v;
}

// Networks whose weights stream from L2 run 16-sample workgroups (one team of 8 waves): the two teams of a
// 32-sample tile share nothing there (each fetches its own fragments), so the narrow tile costs no extra traffic
// and puts twice as many CUs to work at small batches (BASELINE config 5: 2048 columns = 128 workgroups, not 64).
// Measured on config 5, same box: 32-sample tiles 41.8 us per RHS, 16-sample tiles 25.9, with the fragment stream 21.7.
static bool narrow_tiles(const MfmaPlan& p) { return p.variant != 0 && !p.ly.wlds; }

static int step3_grid_for(int B) {
    const int nt = (B + 31) / 32;
    return nt < 512 ? (nt < 1 ? 1 : nt) : 512;
}

// does a step attempt of this solve run on k_step3j?  (same test on the host side -- number of error partials -- and in launch())
static bool step3j_route(const MfmaPlan& p, bool train, bool recording) {
    return p.shape3 && p.ly.jvp && train && !p.cond && !recording && p.d_img3 && !step_v1();
}
static int base_grid_for(const MfmaPlan& p, int B) {
    const int nb = narrow_tiles(p) ? 16 : MF_NB;
    int nt = (B + nb - 1) / nb;
    return nt < 512 ? (nt < 1 ? 1 : nt) : 512;
}
int mfma_grid_for(const MfmaPlan& p, int B, bool recording, bool train) {
    return step3j_route(p, train, recording) ? step3_grid_for(B) : base_grid_for(p, B);
}

template <class LY, int WPT = MF_WPT>
static void launch_static(const MfmaPlan& p, const MfmaArgs& a, dim3 grid, hipStream_t s) {
    LY ly;
    ly.n_in_ = p.ly.n_in; ly.norm_z_ = p.ly.norm_z; ly.norm_j_ = p.ly.norm_j;
    const size_t shm = (size_t)LY::total_floats() * sizeof(float);
    if (a.mode == 2) hipLaunchKernelGGL((k_mfma<LY, true, WPT>), grid, dim3(MF_KTHREADS), shm, s, ly, a);
    else hipLaunchKernelGGL((k_mfma<LY, false, WPT>), grid, dim3(MF_KTHREADS), shm, s, ly, a);
}

static cnf_status launch(const MfmaPlan& p, const MfmaArgs& a0, hipStream_t s) {
    MfmaArgs a = a0;
    if (a.test) { a.cimg = p.d_img + p.ly.c_off; a.SWC = p.ly.SWC; }
    a.cond = p.cond; a.cbs = p.cbs;
    // Resident-fragment kernels of the headline shape (cnf_step3.hip).  k_step3j: JVP handles -- and VJP handles WITHOUT the
    // |eps^T J| row (FFJORD): zdot and ldot = -eps.(J eps) = -(eps^T J).eps do not depend on the mode, and one forward sweep
    // of two column tiles is the shorter schedule.  k_step3: VJP handles with that row.  They take the step attempts
    // (mode 2) and the two single evaluations of the automatic initial dt (modes 0 / 1 with an init phase).
    const bool s3ok = !a.test && !a.cond && !a.dump && p.d_img3 && !step_v1();
    const bool use_j = s3ok && ((p.shape3 && p.ly.jvp) || (p.variant == 2 && !p.ly.norm_j));
    const bool use_v = s3ok && !use_j && p.variant == 2;
    // single evaluations: the two launches of the automatic initial dt (modes 0 / 1 with an init phase) and, for JVP
    // handles, the plain evaluation of cnf_rhs (mode 0 without an integrator state: it reads a zeroed one).  The plain
    // VJP evaluation stays on k_mfma: measured 14.9 us there against 16.5 us here (the heavier prologue), JVP 50 against 14.6.
    int single = a.mode == 2 ? 0 : ((a.st && a.init_phase >= 0) ? a.mode + 1 : -1);            // -1: not theirs
    if (a.mode == 0 && !a.st && a.init_phase < 0 && p.d_idle && p.ly.jvp) { single = 1; a.st = p.d_idle; }
    const bool s3 = (use_j || use_v) && single >= 0;
    const dim3 grid(s3 ? step3_grid_for(a.B) : base_grid_for(p, a.B)), block(MF_KTHREADS);
    const size_t shm = (size_t)p.ly.total_floats * sizeof(float);
    const bool narrow = narrow_tiles(p);
    if (s3) {
        if (a.mode == 0) {      // f(u): the kernel reads its state through the buffer-set pointers (k1 is masked out)
            a.U[0] = a.U[1] = const_cast<float*>(a.u);
            a.K1[0] = a.K1[1] = a.du;
        }
        // k_step3jb: the same schedule with every fp32 product formed from six bf16 MFMA terms on exactly split operands
        // (CNF_STEP_FP32=1 keeps the fp32 MFMA kernel k_step3j: A/B measurements)
        static const bool fp32_only = [] { const char* e = getenv("CNF_STEP_FP32"); return e && e[0] == '1'; }();
        if (use_j && p.d_img3b && !fp32_only) step3jb_launch(a, p.d_img3b, p.ly.n_in, p.ly.norm_z, p.ly.norm_j, grid, s, single);
        else if (use_j) step3j_launch(a, p.d_img3, p.ly.n_in, p.ly.norm_z, p.ly.norm_j, grid, s, single);
        else if (p.d_img3b && !fp32_only && !a.dump) step3b_launch(a, p.d_img3b, p.ly.n_in, p.ly.norm_z, p.ly.norm_j, grid, s, single);
        else step3_launch(a, p.d_img3, p.ly.n_in, p.ly.norm_z, p.ly.norm_j, grid, s, single);
    }
    else if (narrow && p.variant == 6 && !a.test) launch_static<LyCfg5J, MF_WPT_NARROW>(p, a, grid, s);
    else if (narrow && p.variant == 6) launch_static<LyCfg5, MF_WPT_NARROW>(p, a, grid, s);       // TestMode: no tangent images
    else if (narrow && p.variant != 5) {
        RtLayout ly{p.ly};
        if (a.mode == 2) hipLaunchKernelGGL((k_mfma<RtLayout, true, MF_WPT_NARROW>), grid, block, shm, s, ly, a);
        else hipLaunchKernelGGL((k_mfma<RtLayout, false, MF_WPT_NARROW>), grid, block, shm, s, ly, a);
    }
    else if (narrow) launch_static<LyCfg5, MF_WPT_NARROW>(p, a, grid, s);
    else if (a.test && p.variant != 5) {
        // exact trace: the run-time-layout kernel (the static BASELINE shapes 1-3 are TrainMode kernels)
        RtLayout ly{p.ly};
        if (a.mode == 2) hipLaunchKernelGGL((k_mfma<RtLayout, true>), grid, block, shm, s, ly, a);
        else hipLaunchKernelGGL((k_mfma<RtLayout, false>), grid, block, shm, s, ly, a);
    }
    else if (p.variant == 2) launch_static<LyCfg3>(p, a, grid, s);
    else if (p.variant == 3) launch_static<LyCfg2>(p, a, grid, s);
    else if (p.variant == 4) launch_static<LyCfg1>(p, a, grid, s);
    else {
        RtLayout ly{p.ly};
        if (a.mode == 2) hipLaunchKernelGGL((k_mfma<RtLayout, true>), grid, block, shm, s, ly, a);
        else hipLaunchKernelGGL((k_mfma<RtLayout, false>), grid, block, shm, s, ly, a);
    }
    return hipGetLastError() == hipSuccess ? CNF_OK : CNF_ERR_HIP;
}

cnf_status mfma_rhs(const MfmaPlan& p, const NetDesc& nd_, bool train, const float* u,
                    const float* eps, float* du, int B, hipStream_t s) {
    if (!mfma_supported(p, nd_, train, B)) return CNF_ERR_UNSUPPORTED;
    MfmaArgs a{};
    a.init_phase = -1;
    a.test = train ? 0 : 1;
    a.mode = 0; a.B = B; a.img = p.d_img; a.eps = eps; a.u = u; a.du = du;
    return launch(p, a, s);
}

// k1 = f(u0) at the start of a solve, with the first norm of the automatic initial dt (d0, d1) and its
// controller phase folded into the same launch
cnf_status mfma_rhs_init0(const MfmaPlan& p, const NetDesc& nd_, bool train, StepState* st, const float* u,
                          const float* eps, float* du, float* partials, unsigned* ticket, int B, hipStream_t s) {
    if (!mfma_supported(p, nd_, train, B)) return CNF_ERR_UNSUPPORTED;
    MfmaArgs a{};
    a.test = train ? 0 : 1;
    a.mode = 0; a.B = B; a.img = p.d_img; a.eps = eps; a.u = u; a.du = du;
    a.st = st; a.st_out = st; a.init_phase = 0; a.partials = partials; a.ticket = ticket;
    a.n_total = (float)((size_t)(nd_.n_in + (train ? 3 : 1)) * B);
    return launch(p, a, s);
}

cnf_status mfma_rhs_stage(const MfmaPlan& p, const NetDesc& nd_, bool train, const StepState* st,
                          float* const U[2], float* const K1[2], float* const Ks[5],
                          const float* eps, int nk, int B, hipStream_t s, StepState* st_init, float* partials,
                          unsigned* ticket) {
    if (!mfma_supported(p, nd_, train, B) || nk != 1) return CNF_ERR_UNSUPPORTED;
    MfmaArgs a{};
    a.init_phase = -1;
    a.test = train ? 0 : 1;
    a.mode = 1; a.B = B; a.img = p.d_img; a.eps = eps; a.st = st;
    a.U[0] = U[0]; a.U[1] = U[1]; a.K1[0] = K1[0]; a.K1[1] = K1[1]; a.Ks0 = Ks[0];
    if (st_init) {       // second norm of the automatic initial dt + its controller phase in this launch
        a.st_out = st_init; a.init_phase = 1; a.partials = partials; a.ticket = ticket;
        a.n_total = (float)((size_t)(nd_.n_in + (train ? 3 : 1)) * B);
    }
    return launch(p, a, s);
}

// The whole solve of the headline shape in one launch (k_solve3b / k_solve3jb, cnf_step3.hip): at most one 32-column tile
// per workgroup the device holds at once.  CNF_ERR_UNSUPPORTED: not this handle / batch -- the caller streams step
// launches instead.  CNF_PERSISTENT=0 switches it off.
cnf_status mfma_solve_persistent(const MfmaPlan& p, const NetDesc& nd, bool train, StepState* st_out, float* const U[2],
                                 const float* eps, int B, hipStream_t s, void* mirror, unsigned seq, Solve3Args& sv, int device,
                                 float* dump, size_t dump_stride, size_t dump_step_stride, int dump_cap, float* hs_out,
                                 float* const* K1) {
    static const bool off = [] { const char* e = getenv("CNF_PERSISTENT"); return e && e[0] == '0'; }();
    static const bool fp32_only = [] { const char* e = getenv("CNF_STEP_FP32"); return e && e[0] == '1'; }();
    // k_solve3jb (one forward sweep of state and tangent columns): JVP handles, and VJP handles without the |eps^T J| row
    // (FFJORD: zdot and ldot do not depend on the mode); k_solve3b: VJP handles with that row
    const bool vjp_ok = p.variant == 2 && !p.ly.jvp;
    const bool jvp = (p.shape3 && p.ly.jvp) || (vjp_ok && !p.ly.norm_j && !dump);
    // recording (gradient path): k_solve3b<RECORD> for VJP handles, k_solve3jb<.., RECORD> for JVP handles of the shape
    if (dump && !(vjp_ok || (p.shape3 && p.ly.jvp))) return CNF_ERR_UNSUPPORTED;
    if (off || fp32_only || step_v1() || !train || !p.d_img3b || !(jvp || vjp_ok))
        return CNF_ERR_UNSUPPORTED;
    if (p.cond && dump) return CNF_ERR_UNSUPPORTED;         // (conditional recording solves: k_mfma's step launches)
    const int ntile = (B + 31) / 32;
    const int resident = step3b_solve_resident(jvp, dump != nullptr, device);
    if (ntile < 1 || resident < 1) return CNF_ERR_UNSUPPORTED;
    // One 32-column tile per workgroup when the device holds them all at once; larger batches run several tiles per
    // workgroup with the state in the integrator's buffers (the tiles dealt evenly: ceil(ntile / rounds) workgroups)
    int grid = ntile;
    if (ntile > resident) {
        if (dump || !K1) return CNF_ERR_UNSUPPORTED;
        const int rounds = (ntile + resident - 1) / resident;
        grid = (ntile + rounds - 1) / rounds;
        const unsigned long long wt = (unsigned long long)sv.wait_ticks * (unsigned)rounds;      // the bound is per tile carried
        sv.wait_ticks = wt < 4000000000ull ? (unsigned)wt : 4000000000u;
    }
    if (grid > 512) return CNF_ERR_UNSUPPORTED;
    MfmaArgs a{};
    a.init_phase = -1;
    a.mode = 2; a.B = B; a.eps = eps; a.st = st_out; a.st_out = st_out;
    a.n_total = (float)((size_t)(nd.n_in + 3) * B);
    a.U[0] = U[0]; a.U[1] = U[1];
    // Caller-owned columns.  One tile per workgroup: the launch reads u0 where the caller keeps it and writes the final
    // columns where the caller wants them -- no copy launches around the solve.  Otherwise the state lives in the
    // integrator's buffers between attempts: u0 is copied in, the caller copies the result out (sv.u_out = null says so).
    const bool direct = grid == ntile && !dump && sv.u_out != nullptr && !sv.xs;
    if (grid != ntile) sv.u_out = nullptr;                 // (one tile per workgroup: the final columns go where the caller wants them)
    if (sv.u0 && sv.u0 != U[0]) {
        if (direct) a.U[0] = const_cast<float*>(sv.u0);        // (read in the prologue only: the final store goes to sv.u_out)
        else if (hipMemcpyAsync(U[0], sv.u0, (size_t)B * (nd.n_in + 3) * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
            return CNF_ERR_HIP;
    }
    if (K1) { a.K1[0] = K1[0]; a.K1[1] = K1[1]; }
    a.mirror = mirror; a.seq = seq;
    a.cond = p.cond; a.cbs = p.cbs;                          // conditional models: the per-sample first-layer bias rows
    a.dump = dump; a.dump_stride = dump_stride; a.dump_step_stride = dump_step_stride; a.dump_cap = dump_cap; a.hs_out = hs_out;
    sv.nvars = nd.nvars; sv.naugs = nd.naugs; sv.norm_z_aug = nd.norm_z_aug;
    return step3b_solve_launch(a, p.d_img3b, p.ly.n_in, p.ly.norm_z, p.ly.norm_j, grid, s, sv, jvp, device);
}

cnf_status mfma_step(const MfmaPlan& p, const NetDesc& nd, bool train, const StepState* st_in,
                     StepState* st_out, float* const U[2], float* const K1[2], float* const Ks[5],
                     const float* eps, const float* partials_in, float* partials_out, bool apply_ctrl,
                     bool finalize, int B, hipStream_t s, float* dump, size_t dump_stride, void* mirror,
                     unsigned seq, size_t dump_step_stride, int dump_cap, float* hs_out) {
    if (!mfma_supported(p, nd, train, B)) return CNF_ERR_UNSUPPORTED;
    MfmaArgs a{};
    a.init_phase = -1;
    a.test = train ? 0 : 1;
    a.mode = 2; a.B = B; a.img = p.d_img; a.eps = eps; a.st = st_in; a.st_out = st_out;
    a.partials_in = partials_in; a.apply_ctrl = apply_ctrl ? 1 : 0;
    a.n_total = (float)((size_t)(nd.n_in + (train ? 3 : 1)) * B);
    a.U[0] = U[0]; a.U[1] = U[1]; a.K1[0] = K1[0]; a.K1[1] = K1[1]; a.Ks0 = Ks[0];
    a.partials = partials_out;
    a.dump = dump; a.dump_stride = dump_stride;
    a.dump_step_stride = dump_step_stride; a.dump_cap = dump_cap; a.hs_out = hs_out;
    a.mirror = mirror; a.seq = seq;
    cnf_status r = launch(p, a, s);
    if (r != CNF_OK) return r;
    if (finalize) {
        launch_controller(apply_ctrl ? st_out : const_cast<StepState*>(st_in), partials_out, 2, a.n_total, s);
        if (hipGetLastError() != hipSuccess) return CNF_ERR_HIP;
    }
    return CNF_OK;
}
