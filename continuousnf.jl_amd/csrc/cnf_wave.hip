// k_solve_wave -- the WHOLE solve of a small two-layer network, one WAVE per 16-sample tile, nothing but registers.
//
// What the reference runs here: `augmented_f` (src/icnf.jl:318-350 Train/VJP, :384-420 Train/JVP, :148-164 Test with
// src/utils.jl:1-36) six times per Tsit5 step inside `base_sol` (src/base_icnf.jl:137-143), wrapped by `inference_prob`
// (:266-286) and `inference_sol` (:167-189) -- for the networks of its own README and regression test
// (`Dense(n_in => 3 n_in, tanh), Dense(3 n_in => n_in, tanh)`, README.md:47, test/regression_tests.jl:7): BASELINE
// configs 1 and 2.
//
// Why a kernel of its own: on k_mfma these solves are a LATENCY chain -- 6 evaluations x 6-7 barrier-delimited phases
// per attempt at ~0.4 us each, 23-34 launches (profiles/round3_generic_one_launch.txt) -- with the chip empty
// (4096 samples are 128 workgroups).  Here a network whose padded widths fit one wave's registers is evaluated by ONE
// wave with no LDS and no barrier at all:
//   * v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain).  The accumulator tile of a layer -- lane (q, c) holds
//     rows 4q..4q+3 of sample c -- IS the B operand of the next layer's product: k-step j of input tile kt contracts
//     feature 16 kt + 4q + j, taken from register j; the A operand (weights) is loaded in that k order once per solve.
//     Activations never leave the registers that produced them.
//   * row sums (eps.J eps, |zdot|^2, |eps^T J|^2) are two cross-row lane exchanges; afterwards every lane of a sample holds
//     the totals, and lane group q keeps scalar row q (dlogp, E, n) of the Runge-Kutta state.
//   * all six stages, the embedded error estimate, the controller (every lane runs it on the same numbers) and the
//     automatic initial dt are inside; the waves of a launch MEET once per attempt through the tagged words of k_solve3b
//     (cnf_step3.hip): one store and one poll round trip.  u0 assembly, post-processing and the five loss sums ride along:
//     ONE launch per inference.
// TestMode (exact trace): the closed form of two-layer networks, tr J = sigma'_1^T (W_1 .* W_2^T) sigma'_2: one more product
// against C = W_1 .* W_2^T instead of the reverse sweep.  Conditional models: the per-sample first-layer bias rows stay in
// registers for the whole solve.
//
// Forms of the one kernel template (the comments at the template say what each does and what it measured):
//   GRAD  loss_and_grad in the same launch: the discrete adjoint of the accepted steps behind the solve -- TrainMode in both
//         compute modes (src/exts/mlj_ext/core_icnf.jl:59-73 at its batch_size of 32) and TestMode (the exact-trace adjoint:
//         test/call_tests.jl, benchmark/benchmarks.jl) --, the weight-gradient tiles contracted by a HELPER wave (HELP) from
//         factor tiles filed in LDS; RICH: the forward pass also files its evaluations' intermediates for the backward pass;
//   ID2   the second activation is the identity (a PlanarLayer, or the identity layer cnf_create appends to a ONE-layer
//         network: the network of the reference's benchmark suite);
//   WGW   the tiles of a batch of at most 64 samples as the waves of one workgroup (LDS meeting), plain solves.
#include <cstdlib>
#include <map>
#include <mutex>

#include "cnf_wave.h"
#include "cnf_mfma_dev.h"

namespace {

constexpr int WV_VJP = 0, WV_JVP = 1, WV_TEST = 2;
#ifndef WV_GRAD_TANH_ACCURATE
#define WV_GRAD_TANH_ACCURATE true     // the pullback's own forward half: cnf_tanh (2 ulp everywhere) instead of the exp2 / rcp form
#endif

struct WaveArgs {
    NetDesc nd;
    const float* P;        // flat parameters (Lux order: per layer weight out x in column-major, then bias)
    const float* eps;      // n_in x B
    const float* cond;     // conditional models: per-sample first-layer bias [B][cbs] (W1[:, n_in:] ys + b1), else null
    int cbs;
    int B;
    float n_total;         // D * B
    float* U0;             // the integrator's buffer set 0 ([B][D])
    StepState* st_out;
    void* mirror;
    unsigned seq;
    WaveGradArgs g;        // gradient path (GRAD instantiations): the discrete adjoint of the accepted steps in the same launch
};

__device__ __forceinline__ f32x4 mm4(const float (&A)[4], const f32x4& b, f32x4 acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], b[j], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ float wv_wave_sum(float v) {            // every lane ends with the same total (fixed tree)
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}
// tanh for the gradient path: the exp2 / rcp form loses RELATIVE accuracy near 0 (1 - 2/(t + 1) with t ~ 1: an absolute
// 6e-8), which a long span turns into 1e-4 of the gradient's scale on the 2-6-2 network; below 0.25 the odd Taylor polynomial
// up to x^9 (truncation 2e-9 relative) takes over -- both forms on the signed argument, one select
__device__ __forceinline__ float tanh_grad(float a) {
    const float x2 = a * a;
    float p = 0.021869488f;                       // 62/2835
    p = fmaf(p, x2, -0.053968254f);               // -17/315
    p = fmaf(p, x2, 0.13333333f);                 // 2/15
    p = fmaf(p, x2, -0.33333334f);                // -1/3
    const float small = fmaf(a * x2, p, a);
    return fabsf(a) < 0.25f ? small : tanh_fast(a);
}
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Tsit5 rows a_{s+1, 1..6} by value in the kernel arguments (scalar loads, indexed by the stage of a ROLLED stage loop)
struct WvTab { float a[7][8]; };
static const WvTab kWvTab = {{
    {0, 0, 0, 0, 0, 0, 0, 0},
    {TS_A21, 0, 0, 0, 0, 0, 0, 0},
    {TS_A31, TS_A32, 0, 0, 0, 0, 0, 0},
    {TS_A41, TS_A42, TS_A43, 0, 0, 0, 0, 0},
    {TS_A51, TS_A52, TS_A53, TS_A54, 0, 0, 0, 0},
    {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65, 0, 0, 0},
    {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76, 0, 0}}};

// Sum over the four lanes of a sample (c, c + 16, c + 32, c + 48) on the VALU: v_permlane16_swap / v_permlane32_swap exchange
// rows of 16 / halves of 32 lanes between two registers; with both operands the same value, the two results are the value
// of this lane's row (half) partner pair, so their sum is the pair sum in every lane.  (The ds_bpermute form of
// __shfl_xor is an LDS-crossbar round trip per exchange, six of them per evaluation.)
__device__ __forceinline__ float wv_quad_sum(float v) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// NI / NH: 16-row tiles of n_in / of the hidden layer;  TANH: tanh on both layers (compile time: the evaluation is then one
// straight-line block the scheduler can interleave -- a run-time switch per activation call cut it into 2000 blocks)
//
// GRAD (loss_and_grad of small batches: the reference's training step, src/exts/mlj_ext/core_icnf.jl:59-73, on its README /
// regression networks at batch_size 32): the same launch goes on with the DISCRETE ADJOINT of the steps it accepted.  The
// forward pass files the z rows of u_n per accepted step (each lane its own registers: read back by the same lane) and
// the step sizes in LDS; the backward pass recomputes the stage states of a step from u_n (five forward evaluations), then
// pulls the cotangent back through the six stages -- the four sweeps of k_adj (cnf_grad.hip; algebra in
// oracle/cnf_grad_oracle.py) as MFMA products on the register-resident weights -- and contracts the weight-gradient
// factors over the wave's 16 samples with MFMAs whose k index is the SAMPLE (the factor tiles transposed through a
// wave-private LDS buffer: one 16-byte write, four 4-byte reads per tile).  The weight-gradient tiles stay in registers for
// the whole backward pass; each wave writes one partial, k_grad_reduce adds them in wave order (bit-reproducible).
//
// ID2: the second layer's activation is the identity -- a PlanarLayer's, or the identity map appended by the caller to a ONE-layer network (`Dense(n => n, tanh)`, the
// network of the reference's benchmark suite, benchmark/benchmarks.jl:29: W_2 = I, b_2 = 0, identity activation -- exact in
// fp32: products with 1 and 0, sums with 0; nd.id2: its gradient is not written); the activation is compiled out.
//
// WGW = 4: the tiles of a batch of at most 64 samples (the reference's training batch of 32, its benchmark's 64 samples) as the
// WAVES OF ONE WORKGROUP, one per SIMD: they meet through LDS and a workgroup barrier (~0.1 k cycles) instead of the tagged
// words in memory (2.4-2.9 k cycles per attempt: 10-18 % of an attempt of these networks).
//
// RICH (TrainMode / VJP gradients while the store is affordable: small batches): the forward pass also files what its evaluation
// at every stage point already holds -- h_1, sigma'_1, tbar_1 = W_2' (eps sigma'_2), zdot, sigma'_2, eps'J -- and the backward pass
// reads them back (a stage ahead) instead of forming them again: 4 of its 8 products and all of its tanh go (the very same
// numbers: the gradient is bit-identical to the recomputing form's).
//
// HELP (every GRAD instantiation with one tile per workgroup): a SECOND wave of the workgroup, on another SIMD of the CU, does
// the weight-gradient contraction -- it keeps the Wbar tiles, reads the factor tiles the main wave files in LDS (two sets, one
// workgroup barrier per stage evaluation) and issues the 48 MFMAs of a stage beside the main wave's next stage: the
// contraction (a quarter of the backward pass: LDS round trip + MFMAs) leaves the main wave's dependent chain.
template <int NI, int NH, int MODE, bool TANH, bool GRAD = false, bool ID2 = false, int WGW = 1, bool RICH = false>
__global__ void __launch_bounds__(GRAD && WGW == 1 ? 128 : 64 * WGW) k_solve_wave(WaveArgs a, Solve3Args sv, const WvTab tab) {
    constexpr bool HELP = GRAD && WGW == 1;
    static_assert(!RICH || (GRAD && MODE == WV_VJP), "RICH is a form of the TrainMode / VJP gradient");
    constexpr int RR = 3 * NH + 3 * NI;                     // f32x4 per lane and stage evaluation in the rich store
    f32x4* rich_ptr = nullptr;                             // where the next evaluation files them (null: nowhere)
    static_assert(!ID2 || TANH, "ID2 is instantiated for tanh first layers");
    static_assert(!GRAD || TANH, "the in-launch adjoint is written for tanh networks");
    constexpr bool GTEST = GRAD && MODE == WV_TEST;         // the adjoint of the exact-trace solve
    constexpr bool TRAIN = MODE != WV_TEST;
    constexpr int NS = TRAIN ? 3 : 1;                      // scalar rows of the state
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    // WGW > 1: the waves of a workgroup are consecutive tiles.  One workgroup (B <= 64): they meet through LDS alone; SEVERAL
    // workgroups (large batches: up to 512 workgroups x WGW tiles): LDS inside the workgroup, the tagged words between them
    const int wloc = WGW > 1 ? (int)(threadIdx.x >> 6) : 0, nwg = WGW > 1 ? (int)(blockDim.x >> 6) : 1;
    const bool xwg = WGW > 1 && gridDim.x > 1;
    const int wid = WGW > 1 ? (int)blockIdx.x * nwg + wloc : (int)blockIdx.x;      // this wave's tile
    const NetDesc& nd = a.nd;
    const int n_in = nd.n_in, nh = nd.dims[1], D = n_in + NS;
    const int act1 = nd.acts[0], act2 = nd.acts[1];
    constexpr bool fast = TANH;                            // exp2 / rcp form, sigma' = 1 - h^2
    const float* P = a.P;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (sv.t_out && wid == 0 && lane == 0) sv.t_out[0] = __builtin_amdgcn_s_memrealtime();

    // ---- weights as A operands, in the k order the accumulator tiles present: lane (q, i), k-step j of input tile kt ->
    // M[16 m + i][16 kt + 4 q + j].  Straight from the flat vector (L2 hits after the first wave), once per solve. ----
    float fW1[NH][NI][4], fW2[NI][NH][4];
    float fW2T[(MODE == WV_VJP || GRAD) ? NH : 1][NI][4], fW1T[(MODE == WV_VJP || GRAD) ? NI : 1][NH][4], fC[MODE == WV_TEST ? NH : 1][NI][4];
    float fCT[GTEST ? NI : 1][NH][4];
    auto w1 = [&](int o, int k) { return (o < nh && k < n_in) ? P[nd.w_off[0] + o + (size_t)k * nh] : 0.f; };       // W1[o][k]
    auto w2 = [&](int o, int k) { return (o < n_in && k < nh) ? P[nd.w_off[1] + o + (size_t)k * n_in] : 0.f; };     // W2[o][k]
#pragma unroll
    for (int m = 0; m < NH; ++m)
#pragma unroll
        for (int kt = 0; kt < NI; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 16 * m + c, k = 16 * kt + 4 * q + j;
                fW1[m][kt][j] = w1(o, k);
                if (MODE == WV_VJP || GRAD) fW2T[m][kt][j] = w2(k, o);             // W2^T[o][k] = W2[k][o]
                if (MODE == WV_TEST) fC[m][kt][j] = w1(o, k) * w2(k, o);           // C[o][k] = W1[o][k] W2[k][o]
            }
#pragma unroll
    for (int m = 0; m < NI; ++m)
#pragma unroll
        for (int kt = 0; kt < NH; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 16 * m + c, k = 16 * kt + 4 * q + j;
                fW2[m][kt][j] = w2(o, k);
                if (MODE == WV_VJP || GRAD) fW1T[m][kt][j] = w1(k, o);             // W1^T[o][k] = W1[k][o]
                if (GTEST) fCT[m][kt][j] = w1(k, o) * w2(o, k);                    // C^T[o][k] = C[k][o] = W1[k][o] W2[o][k]
            }
    // biases in the accumulator layout (rows 16 m + 4 q + j); conditional models: a row per sample instead of b1
    const int smp = wid * 16 + c;
    const bool live = smp < a.B;
    const size_t sb = live ? (size_t)smp : 0;
    f32x4 b1v[NH], b2v[NI], rmask[NI];
#pragma unroll
    for (int m = 0; m < NH; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 16 * m + 4 * q + j;
            b1v[m][j] = r < nh ? (a.cond ? a.cond[sb * a.cbs + r] : P[nd.b_off[0] + r]) : 0.f;
        }
#pragma unroll
    for (int m = 0; m < NI; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 16 * m + 4 * q + j;
            b2v[m][j] = r < n_in ? P[nd.b_off[1] + r] : 0.f;
            rmask[m][j] = r < n_in ? 1.f : 0.f;            // (rows beyond n_in carry act(0), which is not 0 for every activation)
        }

    // ---- state: z rows (u, k1..k7), the probe rows, and ONE scalar row per lane group (q = 0: dlogp, 1: E, 2: n) ----
    f32x4 uz[NI], kz[7][NI], ep[NI];
    float us, ks[7];
    const bool sown = q < NS;                              // this lane group owns scalar row q
#pragma unroll
    for (int m = 0; m < NI; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 16 * m + 4 * q + j;
            const bool ok = live && r < n_in;
            float v = 0.f;
            if (sv.xs) { if (ok && r < sv.nvars) v = sv.xs[sb * sv.nvars + r]; }     // u0 = (xs; 0)   src/base_icnf.jl:275-282
            else if (ok) v = sv.u0[sb * D + r];
            uz[m][j] = v;
            ep[m][j] = (TRAIN && ok) ? a.eps[sb * n_in + r] : 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) kz[s][m][j] = 0.f;
        }
    us = (!sv.xs && live && sown) ? sv.u0[sb * D + n_in + q] : 0.f;
#pragma unroll
    for (int s = 0; s < 7; ++s) ks[s] = 0.f;

    // ---- one evaluation of augmented_f at z -> (zdot, this lane group's scalar row) ----
    auto act4 = [&](int kind, const f32x4& pre, f32x4& h, f32x4& d, bool accurate = false, bool second = false) __attribute__((always_inline)) {
        if (ID2 && second) {
            h = pre;
            d = f32x4{1.f, 1.f, 1.f, 1.f};
        } else if (fast) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[j] = (GRAD || accurate) ? tanh_grad(pre[j]) : tanh_fast(pre[j]); d[j] = fmaf(-h[j], h[j], 1.f); }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float hh, dd; cnf_act(kind, pre[j], hh, dd); h[j] = hh; d[j] = dd; }
        }
    };
    auto rhs = [&](const f32x4 (&z)[NI], f32x4 (&zd)[NI], float& sd) {
        f32x4 h1[NH], d1[NH];
#pragma unroll
        for (int m = 0; m < NH; ++m) {
            f32x4 acc = b1v[m];
#pragma unroll
            for (int kt = 0; kt < NI; ++kt) acc = mm4(fW1[m][kt], z[kt], acc);
            act4(act1, acc, h1[m], d1[m]);
        }
        f32x4 d2[NI];
        float e2 = 0.f;
#pragma unroll
        for (int m = 0; m < NI; ++m) {
            // one chain per input tile (they run side by side in the matrix pipe), added in tile order
            f32x4 part[NH];
#pragma unroll
            for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW2[m][kt], h1[kt], kt == 0 ? b2v[m] : zero4);
            f32x4 acc = part[0];
#pragma unroll
            for (int kt = 1; kt < NH; ++kt) acc += part[kt];
            f32x4 h2;
            act4(act2, acc, h2, d2[m], false, true);
            zd[m] = h2 * rmask[m];
            d2[m] *= rmask[m];
#pragma unroll
            for (int j = 0; j < 4; ++j) e2 = fmaf(zd[m][j], zd[m][j], e2);
        }
        float ld = 0.f, n2 = 0.f;
        if (MODE == WV_VJP) {
            f32x4 g2[NI], g1[NH];
#pragma unroll
            for (int m = 0; m < NI; ++m) g2[m] = ep[m] * d2[m];
#pragma unroll
            for (int m = 0; m < NH; ++m) {
                f32x4 acc = zero4;
#pragma unroll
                for (int kt = 0; kt < NI; ++kt) acc = mm4(fW2T[m][kt], g2[kt], acc);
                g1[m] = acc * d1[m];
                if (RICH && rich_ptr) { rich_ptr[m] = h1[m]; rich_ptr[NH + m] = d1[m]; rich_ptr[2 * NH + m] = acc; }
            }
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                f32x4 part[NH];
#pragma unroll
                for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW1T[m][kt], g1[kt], zero4);
                f32x4 eJ = part[0];
#pragma unroll
                for (int kt = 1; kt < NH; ++kt) eJ += part[kt];
#pragma unroll
                for (int j = 0; j < 4; ++j) { ld = fmaf(eJ[j], ep[m][j], ld); n2 = fmaf(eJ[j], eJ[j], n2); }
                if (RICH && rich_ptr) { rich_ptr[3 * NH + m] = zd[m]; rich_ptr[3 * NH + NI + m] = d2[m]; rich_ptr[3 * NH + 2 * NI + m] = eJ; }
            }
        } else if (MODE == WV_JVP) {
            f32x4 t1[NH];
#pragma unroll
            for (int m = 0; m < NH; ++m) {
                f32x4 acc = zero4;
#pragma unroll
                for (int kt = 0; kt < NI; ++kt) acc = mm4(fW1[m][kt], ep[kt], acc);
                t1[m] = acc * d1[m];
            }
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                f32x4 part[NH];
#pragma unroll
                for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW2[m][kt], t1[kt], zero4);
                f32x4 acc = part[0];
#pragma unroll
                for (int kt = 1; kt < NH; ++kt) acc += part[kt];
                const f32x4 t2 = acc * d2[m];
#pragma unroll
                for (int j = 0; j < 4; ++j) { ld = fmaf(t2[j], ep[m][j], ld); n2 = fmaf(t2[j], t2[j], n2); }
            }
        } else {                                           // exact trace: sigma'_1^T (C sigma'_2)
#pragma unroll
            for (int m = 0; m < NH; ++m) {
                f32x4 acc = zero4;
#pragma unroll
                for (int kt = 0; kt < NI; ++kt) acc = mm4(fC[m][kt], d2[kt], acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) ld = fmaf(acc[j], d1[m][j], ld);
            }
        }
        // row sums: the four lanes of a sample (q = 0..3) exchange twice; every one of them ends with the totals
        ld = wv_quad_sum(ld);
        if (TRAIN) {
            e2 = wv_quad_sum(e2); n2 = wv_quad_sum(n2);
            // (v_sqrt_f32, 1 ulp, on the one value this lane group keeps: the IEEE form is ~15 instructions and two
            // exec-masked branches per root)
            const float r2 = q == 1 ? (nd.norm_z ? e2 : 0.f) : (q == 2 ? (nd.norm_j ? n2 : 0.f) : 0.f);
            const float rt = __builtin_amdgcn_sqrtf(r2);
            sd = q == 0 ? -ld : rt;
        } else {
            sd = q == 0 ? -ld : 0.f;
        }
    };

    __shared__ float hsL[GRAD ? WV_GCAP : 1];                                           // step sizes of the accepted steps
    constexpr int TBW = 4 * (NI + NH) * 256;                // floats of one set of factor tiles
    // factor tiles, [sample][row]: one set per wave -- HELP: two sets, filed by the main wave and contracted by the helper in turn
    __shared__ __attribute__((aligned(16))) float tbuf_all[GRAD ? (HELP ? 2 : WGW) * TBW : 4];
    float* tbuf = tbuf_all + (GRAD && WGW > 1 ? (threadIdx.x >> 6) * TBW : 0);
    __shared__ int hcnt;                                    // HELP: stage evaluations the helper has to contract (0: none)
    if constexpr (HELP) {
        if ((threadIdx.x >> 6) == 1) {
            // ================= the helper wave: Wbar_l += abar_l h_{l-1}' + pbar_l t_{l-1}' over the tile's 16 samples =================
            // MFMAs whose k index is the SAMPLE, on the factor tiles the main wave filed [sample][row] (slots: [0, NI) abar_2 |
            // pbar_2 | z | tau | then NH each: h_1, t_1, abar_1, pbar_1;  TestMode: pbar_2 = s'_2, t_1 = c_l s'_1 -> H)
            f32x4 gW2[NI][NH], gW1[NH][NI], gH[GTEST ? NH : 1][NI];
#pragma unroll
            for (int m = 0; m < NI; ++m)
#pragma unroll
                for (int k = 0; k < NH; ++k) { gW2[m][k] = zero4; gW1[k][m] = zero4; if (GTEST) gH[k][m] = zero4; }
            __syncthreads();                               // the main wave has said how many stage evaluations there are
            const int cnt = __builtin_amdgcn_readfirstlane(hcnt);
            const float* set = tbuf_all;
            for (int it = 0; it < cnt; ++it) {
                __syncthreads();                           // ... and has filed the next one's tiles in `set`
                auto hget = [&](int slot, float (&o)[4]) __attribute__((always_inline)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = set[slot * 256 + (4 * j + q) * 16 + c];
                };
                float A2[NI][4], P2[NI][4], Z0[NI][4], T0[NI][4];
#pragma unroll
                for (int m = 0; m < NI; ++m) { hget(m, A2[m]); hget(NI + m, P2[m]); hget(2 * NI + m, Z0[m]); hget(3 * NI + m, T0[m]); }
#pragma unroll
                for (int k = 0; k < NH; ++k) {
                    float H1[4], T1[4], A1[4], P1[4];
                    hget(4 * NI + k, H1); hget(4 * NI + NH + k, T1); hget(4 * NI + 2 * NH + k, A1); hget(4 * NI + 3 * NH + k, P1);
#pragma unroll
                    for (int m = 0; m < NI; ++m) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            gW2[m][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[m][j], H1[j], gW2[m][k], 0, 0, 0);
                            gW1[k][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[j], Z0[m][j], gW1[k][m], 0, 0, 0);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (GTEST) {
                                gH[k][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(T1[j], P2[m][j], gH[k][m], 0, 0, 0);
                            } else {
                                gW2[m][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(P2[m][j], T1[j], gW2[m][k], 0, 0, 0);
                                gW1[k][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(P1[j], T0[m][j], gW1[k][m], 0, 0, 0);
                            }
                        }
                    }
                }
                set = set == tbuf_all ? tbuf_all + TBW : tbuf_all;
            }
            if (cnt > 0) {
                if constexpr (GTEST) {
                    // Wbar_1[k][i] -= H[k][i] W_2[i][k];  Wbar_2[i][k] -= H[k][i] W_1[k][i]  (H's tile transposed through LDS: the
                    // main wave files nothing after its last barrier)
#pragma unroll
                    for (int k = 0; k < NH; ++k)
#pragma unroll
                        for (int m = 0; m < NI; ++m) {
                            reinterpret_cast<f32x4*>(tbuf_all)[c * 4 + q] = gH[k][m];
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
                            float Ht[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) Ht[j] = tbuf_all[(4 * q + j) * 16 + c];     // H[16 k + c][16 m + 4 q + j]
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                gW1[k][m][j] -= gH[k][m][j] * w2(16 * m + c, 16 * k + 4 * q + j);
                                gW2[m][k][j] -= Ht[j] * w1(16 * k + c, 16 * m + 4 * q + j);
                            }
                        }
                }
                float* gp = a.g.gpart + (size_t)wid * a.g.n_params;
#pragma unroll
                for (int m = 0; m < NI; ++m)
#pragma unroll
                    for (int k = 0; k < NH; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (!nd.id2) {   // Wbar_2[o][kk]: row o = 16 m + 4 q + j, column kk = 16 k + c   (an APPENDED identity layer has no parameters)
                                const int o = 16 * m + 4 * q + j, kk = 16 * k + c;
                                if (o < n_in && kk < nh) gp[nd.w_off[1] + o + (size_t)kk * n_in] = gW2[m][k][j];
                            }
                            {   // Wbar_1[o][kk]: row o = 16 k + 4 q + j, column kk = 16 m + c
                                const int o = 16 * k + 4 * q + j, kk = 16 * m + c;
                                if (o < nh && kk < n_in + nd.n_cond) gp[nd.w_off[0] + o + (size_t)kk * nh] = gW1[k][m][j];
                            }
                        }
            }
            return;
        }
    }
    __shared__ float mw[2][WGW][2];                        // WGW > 1: the waves' meeting words
    __shared__ float mx[2][4];                             // ... and what wave 0 brought back from the other workgroups
    __shared__ float mv[WGW][4];                           // ... and the waves' loss-sum partials
    bool gover = false;                                    // more accepted steps than the trajectory store holds
    // ---- integrator state: every lane carries the same copy and runs the same controller on the same sums ----
    StepState ns = sv.init;
    float hstep = ns.h, abstol = ns.abstol, reltol = ns.reltol;
    int nsync = 0;
    const unsigned mbase = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sv.base_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int G = WGW > 1 ? (int)gridDim.x * nwg : (int)gridDim.x;      // tiles
    // the rich store's slot of (step, stage): [step][stage][wave][lane][RR]
    auto rich_slot = [&](int step, int stage) -> f32x4* {
        if (!RICH || !a.g.rich || step >= a.g.traj_cap) return nullptr;
        return reinterpret_cast<f32x4*>(a.g.rich) + (((size_t)step * 6 + stage) * G + wid) * (size_t)(64 * RR) + lane * RR;
    };
    float p0 = 0.f, p1 = 0.f;
    // The waves' partials (e, b) -> the sums over all of them in p0, p1 (the same order in every wave).  false: a wait ran out.
    // The tagged-word exchange: party `me` of `parties` files (e, b) and collects everyone's; the sums land in p0, p1 (the same
    // order in every party).  false: a wait ran out.
    auto exchange = [&](float e, float b, int me, int parties) -> bool {
        unsigned long long* pb = reinterpret_cast<unsigned long long*>(sv.part) + (nsync & 1) * 1024;
        const unsigned tag = mbase + (unsigned)nsync + 1u;
        if (lane == 0) {
            __hip_atomic_store(pb + 2 * me, ((unsigned long long)tag << 32) | __float_as_uint(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pb + 2 * me + 1, ((unsigned long long)tag << 32) | __float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lane i takes the words of parties i, i + 64, ... (at most 8): ALL of them requested at once per poll round -- one round
        // trip per round whatever the grid (taken one after the other they cost a round trip each: 8.5 k cycles per meeting
        // at 256 waves) --, each accepted when both halves carry this meeting's index, in party order
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        const auto prs = __builtin_amdgcn_make_buffer_rsrc(pb, 0, 16 * 512, 0x00020000);
        const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
        float pe[8], pbv[8];
        unsigned need = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { pe[i] = 0.f; pbv[i] = 0.f; if (lane + 64 * i < parties) need |= 1u << i; }
        int ok = 1;
        for (int spin = 0; need != 0; ++spin) {
            // (buffer loads past the caches, aux = sc0 | sc1: loads the compiler counts and waits for itself -- a hand-written
            // request's destination register is not protected until its wait)
            u32x4_ wq[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (64 * i < parties)                      // (wave-uniform: rounds beyond the grid issue nothing)
                    wq[i] = __builtin_bit_cast(u32x4_, __builtin_amdgcn_raw_buffer_load_b128(prs, 16 * min(lane + 64 * i, parties - 1), 0, 0x11));
                else wq[i] = u32x4_{0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (64 * i < parties && (need >> i & 1) && wq[i].y == tag && wq[i].w == tag) {
                    pe[i] = __uint_as_float(wq[i].x); pbv[i] = __uint_as_float(wq[i].z); need &= ~(1u << i);
                }
            if (need == 0) break;
            if (spin + 1 >= sv.spin_limit) { ok = 0; break; }
            if ((spin & 63) == 63 && __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        float c0 = 0.f, c1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { c0 += pe[i]; c1 += pbv[i]; }
        if (!ok) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p0 = wv_wave_sum(c0); p1 = wv_wave_sum(c1);
        const float bad = wv_wave_sum(ok ? 0.f : 1.f);
        return bad == 0.f;
    };
    // The waves' partials (e, b) -> the sums over all of them in p0, p1 (the same order in every wave).  false: a wait ran out.
    auto meet = [&](float e_lane, float b_lane) -> bool {
        const float e = wv_wave_sum(e_lane), b = wv_wave_sum(b_lane);
        if constexpr (WGW > 1) {                           // the waves of one workgroup: LDS words (two sets by parity) and a barrier
            const int par = nsync & 1;
            if (lane == 0) { mw[par][wloc][0] = e; mw[par][wloc][1] = b; }
            __syncthreads();
            float c0 = 0.f, c1 = 0.f;
            for (int w = 0; w < nwg; ++w) { c0 += mw[par][w][0]; c1 += mw[par][w][1]; }
            if (!xwg) { p0 = c0; p1 = c1; ++nsync; return true; }
            // several workgroups: wave 0 carries the workgroup's sums to the others and brings theirs back
            if (wloc == 0) {
                const bool ok = exchange(c0, c1, (int)blockIdx.x, (int)gridDim.x);
                if (lane == 0) { mx[par][0] = p0; mx[par][1] = p1; mx[par][2] = ok ? 0.f : 1.f; }
            }
            __syncthreads();
            p0 = mx[par][0]; p1 = mx[par][1];
            const float bad = mx[par][2];
            ++nsync;
            return uni(bad) == 0.f;
        }
        const bool ok = exchange(e, b, wid, (int)gridDim.x);
        ++nsync;
        return ok;
    };
    auto after_ctrl = [&]() {
        hstep = uni(ns.h); abstol = uni(ns.abstol); reltol = uni(ns.reltol);
    };
    auto add_norm = [&](float& acc, float u, float x) {   // (x / sk)^2, sk = atol + rtol |u|
        const float sk = fmaf(fabsf(u), reltol, abstol);
        const float y = x / sk;
        acc = fmaf(y, y, acc);
    };

    bool alive = true;
    {
        // ---- k1 = f(u0); automatic initial dt (Hairer): its two norms, f(u0 + h0 f0) and that norm ----
        float e = 0.f, b = 0.f;
        rich_ptr = rich_slot(0, 0);                        // k1 = f(u_0): stage 1 of step 0
        rhs(uz, kz[0], ks[0]);
        rich_ptr = nullptr;
        if (live) {
#pragma unroll
            for (int m = 0; m < NI; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (rmask[m][j] != 0.f) { add_norm(e, uz[m][j], uz[m][j]); add_norm(b, uz[m][j], kz[0][m][j]); }
            if (sown) { add_norm(e, us, us); add_norm(b, us, ks[0]); }
        }
        if (sv.hairer) alive = meet(e, b);
        if (sv.hairer && alive) {
            ctrl_phase(&ns, 0, p0, p1, a.n_total);
            after_ctrl();
            e = 0.f;
            f32x4 zt[NI], f1[NI];
            float s1;
#pragma unroll
            for (int m = 0; m < NI; ++m) zt[m] = uz[m] + hstep * kz[0][m];
            rhs(zt, f1, s1);
            if (live) {
#pragma unroll
                for (int m = 0; m < NI; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (rmask[m][j] != 0.f) add_norm(e, uz[m][j], f1[m][j] - kz[0][m][j]);
                if (sown) add_norm(e, us, s1 - ks[0]);
            }
            alive = meet(e, 0.f);
            if (alive) { ctrl_phase(&ns, 1, p0, p1, a.n_total); after_ctrl(); }
        }
    }
    // ---- step attempts ----
    constexpr float BT[7] = {TS_BT1, TS_BT2, TS_BT3, TS_BT4, TS_BT5, TS_BT6, TS_BT7};
    f32x4 zt[NI];
#pragma unroll
    for (int m = 0; m < NI; ++m) zt[m] = zero4;
    for (int it = 0; alive && !__builtin_amdgcn_readfirstlane(ns.done) && it < sv.maxiters; ++it) {
        float errsum = 0.f, badcnt = 0.f, sacc = 0.f;
#ifdef WV_STAMPS
        const unsigned long long wv_t0 = __builtin_amdgcn_s_memtime();
#endif
        // stage state U_{s+1} = u + h sum_j a_{s+1,j} k_j, then k_{s+1} = f(U_{s+1}): ONE copy of the evaluation code, the
        // stage's row of the table by scalar loads, its result filed by selects (an indexed store would go to scratch)
        // gradient path: every attempt files its stage states U_1 = u_n, U_2..U_6 in the slot of step `naccept` (a rejected
        // attempt's are overwritten by the next one), each lane its own registers
        f32x4* tjs = nullptr;
        if (GRAD && ns.naccept < a.g.traj_cap) {
            tjs = reinterpret_cast<f32x4*>(a.g.traj) + ((size_t)ns.naccept * 6 * G + wid) * (64 * NI) + lane * NI;
#pragma unroll
            for (int m = 0; m < NI; ++m) tjs[m] = uz[m];
        }
#pragma unroll 1
        for (int s = 1; s <= 6; ++s) {
            float as[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) as[j] = tab.a[s][j];
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                f32x4 acc = as[0] * kz[0][m];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc += as[j] * kz[j][m];
                zt[m] = uz[m] + hstep * acc;
            }
            if (GRAD && tjs && s < 6) {
#pragma unroll
                for (int m = 0; m < NI; ++m) tjs[(size_t)s * G * (64 * NI) + m] = zt[m];
            }
            if (s == 6) {                                  // a_7j = b_j: zt is the new solution (FSAL); its scalar row likewise
                sacc = as[0] * ks[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) sacc += as[j] * ks[j];
            }
            f32x4 zd[NI];
            float sd;
            // (stages 2..6 of this step; the evaluation at the new solution is stage 1 of the NEXT step -- FSAL --: a rejected
            // attempt's is overwritten by the accepted one's)
            rich_ptr = s < 6 ? rich_slot(ns.naccept, s) : rich_slot(ns.naccept + 1, 0);
            rhs(zt, zd, sd);
            rich_ptr = nullptr;
#pragma unroll
            for (int i = 1; i < 7; ++i) {
#pragma unroll
                for (int m = 0; m < NI; ++m) kz[i][m] = s == i ? zd[m] : kz[i][m];
                ks[i] = s == i ? sd : ks[i];
            }
        }
        const float uns = us + hstep * sacc;
        if (live) {
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                f32x4 ez = BT[0] * kz[0][m];
#pragma unroll
                for (int j = 1; j < 7; ++j) ez += BT[j] * kz[j][m];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (rmask[m][j] != 0.f) {
                        const float scl = fmaf(fmaxf(fabsf(uz[m][j]), fabsf(zt[m][j])), reltol, abstol);
                        const float x = hstep * ez[j] / scl;
                        errsum = fmaf(x, x, errsum);
                        badcnt += !(fabsf(zt[m][j]) <= 3.0e38f) ? 1.f : 0.f;
                    }
            }
            if (sown) {
                float es = BT[0] * ks[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) es += BT[j] * ks[j];
                const float scl = fmaf(fmaxf(fabsf(us), fabsf(uns)), reltol, abstol);
                const float x = hstep * es / scl;
                errsum = fmaf(x, x, errsum);
                badcnt += !(fabsf(uns) <= 3.0e38f) ? 1.f : 0.f;
            }
        }
#ifdef WV_STAMPS
        const unsigned long long wv_t1 = __builtin_amdgcn_s_memtime();
#endif
        alive = meet(errsum, badcnt);
#ifdef WV_STAMPS
        const unsigned long long wv_t2 = __builtin_amdgcn_s_memtime();
#endif
        if (!alive) break;
        const int acc0 = ns.naccept;
        const float t_att = ns.t, h_att = ns.h;
        ctrl_after_step(&ns, p0, p1, a.n_total);
        const bool accepted = __builtin_amdgcn_readfirstlane(ns.naccept != acc0);
        if (sv.trace && wid == 0 && lane == 0 && it < sv.trace_cap) {
            float* tr = sv.trace + 4 * it;
            tr[0] = t_att; tr[1] = h_att; tr[2] = ns.eest; tr[3] = accepted ? 1.f : 0.f;
#ifdef WV_STAMPS
            tr[0] = (float)(wv_t1 - wv_t0); tr[1] = (float)(wv_t2 - wv_t1); tr[2] = (float)(__builtin_amdgcn_s_memtime() - wv_t2);   // stages+error | meeting | controller
#endif
        }
        after_ctrl();
        if (GRAD && accepted) {                            // h_n of this step (its stage states were filed by the attempt)
            if (acc0 < a.g.traj_cap && acc0 < WV_GCAP) {
                hsL[acc0] = h_att;
                if (wid == 0 && lane == 0) a.g.hs_out[acc0] = h_att;
            } else gover = true;
        }
        if (accepted) {                                    // u <- u_new, k1 <- k7
#pragma unroll
            for (int m = 0; m < NI; ++m) { uz[m] = zt[m]; kz[0][m] = kz[6][m]; }
            us = uns; ks[0] = ks[6];
        }
    }
    // ---- the final state: to the caller's columns (a bare solve) or to the integrator's buffer set 0; never to the
    // caller's columns after an abort (the caller may be solving in place; the streamed driver starts again from u0) ----
    float* out = sv.u_out ? sv.u_out : a.U0;
    if ((alive || !sv.u_out) && live) {
        float* o = out + (size_t)smp * D;
#pragma unroll
        for (int m = 0; m < NI; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = 16 * m + 4 * q + j; if (r < n_in) o[r] = uz[m][j]; }
        if (sown) o[n_in + q] = us;
    }
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
    if (sv.logpx && alive) {
        // ---- inference_sol (src/base_icnf.jl:167-189): logp(z) - dlogp, the regulariser rows, |z_aug| ----
        float ss = 0.f, sa = 0.f;
#pragma unroll
        for (int m = 0; m < NI; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * m + 4 * q + j;
                const float v = uz[m][j];
                ss = fmaf(v, v, ss);
                if (r >= sv.nvars) sa = fmaf(v, v, sa);
            }
        ss = wv_quad_sum(ss); sa = wv_quad_sum(sa);
        // scalar rows of this sample: lane group q holds row q; bring E and n to group 0
        const float s1 = __shfl(us, c + 16, 64), s2 = __shfl(us, c + 32, 64);
        if (live && q == 0) {
            const float log2pi = 1.8378770664093453f;
            const float lp = -0.5f * fmaf((float)n_in, log2pi, ss) - us;
            const float aa = (sv.norm_z_aug && sv.naugs > 0) ? sqrtf(sa) : 0.f;
            const float Ev = TRAIN ? s1 : 0.f, Nv = TRAIN ? s2 : 0.f;
            const size_t bb = (size_t)smp, Bz = (size_t)a.B;
            sv.logpx[bb] = lp; sv.regs[bb] = Ev; sv.regs[Bz + bb] = Nv; sv.regs[2 * Bz + bb] = aa;
            v4[0] = lp; v4[1] = Ev; v4[2] = Nv; v4[3] = aa;
        }
        if (sv.sums5) {
            unsigned long long* qb = reinterpret_cast<unsigned long long*>(sv.part) + 2048;
            const unsigned tag = mbase + (unsigned)nsync + 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = wv_wave_sum(v4[j]);
            int slot = wid, parties = G;                   // one word quadruple per wave ...
            bool files = true;
            if constexpr (WGW > 1) {
                if (xwg) {                                 // ... or per workgroup: its waves' partials added in wave order first
                    if (lane == 0) { mv[wloc][0] = v4[0]; mv[wloc][1] = v4[1]; mv[wloc][2] = v4[2]; mv[wloc][3] = v4[3]; }
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < 4; ++j) { float t = 0.f; for (int w = 0; w < nwg; ++w) t += mv[w][j]; v4[j] = t; }
                    slot = (int)blockIdx.x; parties = (int)gridDim.x; files = wloc == 0;
                }
            }
            if (files && lane < 4)
                __hip_atomic_store(qb + 4 * slot + lane, ((unsigned long long)tag << 32) | __float_as_uint(lane == 0 ? v4[0] : lane == 1 ? v4[1] : lane == 2 ? v4[2] : v4[3]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wid == 0) {                         // wave 0 adds the partials in wave (workgroup) order
                float c4[4] = {0.f, 0.f, 0.f, 0.f};
                float late = 0.f;
                const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
                for (int w = lane; w < parties; w += 64) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int got = 0;
                        for (int spin = 0; spin < sv.spin_limit; ++spin) {
                            const unsigned long long ww = __hip_atomic_load(qb + 4 * w + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(ww >> 32) == tag) { c4[j] += __uint_as_float((unsigned)ww); got = 1; break; }
                            if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (!got) late = 1.f;
                    }
                }
                if (late != 0.f) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[j] = wv_wave_sum(c4[j]);
                late = wv_wave_sum(late);
                if (lane == 0) {
                    sv.sums5[0] = c4[0]; sv.sums5[1] = c4[1]; sv.sums5[2] = c4[2]; sv.sums5[3] = c4[3]; sv.sums5[4] = (float)a.B;
                }
                if (late != 0.f) alive = false;
            }
        }
    }
    if constexpr (GRAD) {
        const bool run_bwd = alive && !gover && __builtin_amdgcn_readfirstlane(ns.done) && !__builtin_amdgcn_readfirstlane(ns.nonfinite);
        if constexpr (HELP) {                              // the helper learns how many stage evaluations it will be handed
            if (lane == 0) hcnt = run_bwd ? 6 * __builtin_amdgcn_readfirstlane(ns.naccept) : 0;
            __syncthreads();
        }
        if (run_bwd) {
            // ================= backward: discrete adjoint of the accepted steps (oracle/cnf_grad_oracle.py) =================
            const float invB = 1.0f / (float)a.B;
            const float cl0 = invB, cE0 = a.g.lam1 * invB, cn0 = a.g.lam2 * invB;   // cotangents of the scalar rows: constants
            // d loss / d z(t1) = (z + lam3 unit(z_aug) on the augmented rows) / B      (k_final_cotangent)
            f32x4 lam[NI];
            {
                float sa = 0.f;
                const bool aug = TRAIN && nd.norm_z_aug && nd.naugs > 0;      // (TestMode: loss = -mean(logpx), src/base_icnf.jl:489-497)
#pragma unroll
                for (int m = 0; m < NI; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int r = 16 * m + 4 * q + j; if (aug && r >= nd.nvars) sa = fmaf(uz[m][j], uz[m][j], sa); }
                sa = wv_quad_sum(sa);
                const float nrm = sqrtf(sa);
#pragma unroll
                for (int m = 0; m < NI; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 16 * m + 4 * q + j;
                        float v = uz[m][j];
                        if (aug && r >= nd.nvars && nrm > 0.f) v = fmaf(a.g.lam3, uz[m][j] / nrm, v);
                        lam[m][j] = live ? v * invB : 0.f;
                    }
            }
            f32x4 gW2[NI][NH], gW1[NH][NI], gb1[NH], gb2[NI];
            f32x4 gH[GTEST ? NH : 1][NI];                   // TestMode: sum over stages and samples of c_l s'_1 s'_2' (see below)
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                gb2[m] = zero4;
#pragma unroll
                for (int k = 0; k < NH; ++k) { gW2[m][k] = zero4; gW1[k][m] = zero4; if (GTEST) gH[k][m] = zero4; }
            }
#pragma unroll
            for (int k = 0; k < NH; ++k) gb1[k] = zero4;
            // the forward half of an evaluation: nn(z) and the activations' derivatives
            auto fwd2 = [&](const f32x4 (&z)[NI], f32x4 (&h1)[NH], f32x4 (&d1)[NH], f32x4 (&zd)[NI], f32x4 (&d2)[NI]) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < NH; ++m) {
                    f32x4 acc = b1v[m];
#pragma unroll
                    for (int kt = 0; kt < NI; ++kt) acc = mm4(fW1[m][kt], z[kt], acc);
                    act4(act1, acc, h1[m], d1[m], WV_GRAD_TANH_ACCURATE);
                }
#pragma unroll
                for (int m = 0; m < NI; ++m) {
                    f32x4 part[NH];
#pragma unroll
                    for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW2[m][kt], h1[kt], kt == 0 ? b2v[m] : zero4);
                    f32x4 acc = part[0];
#pragma unroll
                    for (int kt = 1; kt < NH; ++kt) acc += part[kt];
                    f32x4 h2;
                    act4(act2, acc, h2, d2[m], WV_GRAD_TANH_ACCURATE, true);
                    zd[m] = h2 * rmask[m];
                    d2[m] *= rmask[m];
                }
            };
            // sigma'' from what the forward half kept (tanh: -2 h sigma'); other activations: from the pre-activation again
            auto dd_of = [&](bool second, const f32x4& h, const f32x4& d) __attribute__((always_inline)) {
                f32x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = (ID2 && second) ? 0.f : -2.0f * h[j] * d[j];
                return r;
            };
            // conditional models: the conditioning inputs ys sit behind z in the first layer's input (h_0 = [z; ys], src/base_icnf.jl:
            // 288-309); they enter the backward pass only as rows n_in .. n_in + n_cond - 1 of the operand Wbar_1 is contracted with
            f32x4 ysv[NI];
#pragma unroll
            for (int m = 0; m < NI; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 16 * m + 4 * q + j;
                    ysv[m][j] = (a.g.ys && live && r >= n_in && r < n_in + nd.n_cond) ? a.g.ys[sb * nd.n_cond + (r - n_in)] : 0.f;
                }
            f32x4 W1eps[MODE == WV_JVP ? NH : 1];           // JVP mode: p_1 = W_1 eps, the same at every stage
            if constexpr (MODE == WV_JVP) {
#pragma unroll
                for (int m = 0; m < NH; ++m) {
                    f32x4 acc = zero4;
#pragma unroll
                    for (int kt = 0; kt < NI; ++kt) acc = mm4(fW1[m][kt], ep[kt], acc);
                    W1eps[m] = acc;
                }
            }
            const int nacc = __builtin_amdgcn_readfirstlane(ns.naccept);
            const f32x4* tj0 = reinterpret_cast<const f32x4*>(a.g.traj) + (size_t)wid * (64 * NI) + lane * NI;
            const size_t tstage = (size_t)G * (64 * NI), tstep = 6 * tstage;
            f32x4 U_next[6][NI];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int m = 0; m < NI; ++m) U_next[i][m] = nacc > 0 ? tj0[(size_t)(nacc - 1) * tstep + i * tstage + m] : zero4;
            // between the writes and the reads of the wave's OWN transposition buffer: a one-wave workgroup's barrier; with
            // several waves per workgroup nothing but this wave's LDS traffic in order (its DS instructions execute in issue
            // order) -- a workgroup barrier here would tie the waves' backward passes together for nothing
            auto tsync = [&]() __attribute__((always_inline)) {
                if constexpr (WGW == 1) __syncthreads();
                else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); }
            };
            auto tput = [&](int slot, const f32x4& v) __attribute__((always_inline)) {
                reinterpret_cast<f32x4*>(tbuf + slot * 256)[c * 4 + q] = v;
            };
            auto tget = [&](int slot, float (&o)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = tbuf[slot * 256 + (4 * j + q) * 16 + c];
            };
            f32x4 Rnx[RICH ? RR : 1];
            if constexpr (RICH) {
                const f32x4* nx = nacc > 0 ? rich_slot(nacc - 1, 5) : nullptr;
#pragma unroll
                for (int r = 0; r < RR; ++r) Rnx[r] = nx ? nx[r] : zero4;
            }
            for (int n = nacc - 1; n >= 0; --n) {
                const float hn = hsL[n];
                f32x4 U[6][NI];                            // the stage states the forward pass filed; the next step's are requested a step ahead
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int m = 0; m < NI; ++m) U[i][m] = U_next[i][m];
                if (n > 0) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int m = 0; m < NI; ++m) U_next[i][m] = tj0[(size_t)(n - 1) * tstep + i * tstage + m];
                }
                f32x4 ws[6][NI];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int m = 0; m < NI; ++m) ws[i][m] = zero4;
#pragma unroll 1
                for (int i = 5; i >= 0; --i) {
                    // ---- this stage's point and the cotangent of its k:  kbar = h (b_i lam + sum_{m > i} a_{m,i} w_m) ----
                    f32x4 z[NI], kb[NI];
                    const float bi = tab.a[6][i];
#pragma unroll
                    for (int m = 0; m < NI; ++m) {
                        z[m] = U[0][m];
#pragma unroll
                        for (int k = 1; k < 6; ++k) z[m] = i == k ? U[k][m] : z[m];
                        f32x4 acc = bi * lam[m];
#pragma unroll
                        for (int k = 1; k < 6; ++k) acc += tab.a[k][i] * ws[k][m];     // (a_{k,i} = 0 for k <= i)
                        kb[m] = hn * acc;
                    }
                    const float c_l = hn * bi * cl0, c_E = hn * bi * cE0, c_n = hn * bi * cn0;
                    (void)c_E; (void)c_n;
                    // ---- forward: h_1, sigma', sigma'' ; zdot ; ahat = kbar + c_E zdot / |zdot| ----
                    f32x4 h1[NH], d1[NH], zd[NI], d2[NI];
                    f32x4 Rc[RICH ? RR : 1];                // RICH: what the forward pass filed for this stage point, read a stage ahead
                    if constexpr (RICH) {
#pragma unroll
                        for (int r = 0; r < RR; ++r) Rc[r] = Rnx[r];
                        // the next stage point backwards: (n, i - 1), or the last one of the step before
                        const f32x4* nx = i > 0 ? rich_slot(n, i - 1) : (n > 0 ? rich_slot(n - 1, 5) : nullptr);
                        if (nx) {
#pragma unroll
                            for (int r = 0; r < RR; ++r) Rnx[r] = nx[r];
                        }
#pragma unroll
                        for (int m = 0; m < NH; ++m) { h1[m] = Rc[m]; d1[m] = Rc[NH + m]; }
#pragma unroll
                        for (int m = 0; m < NI; ++m) { zd[m] = Rc[3 * NH + m]; d2[m] = Rc[3 * NH + NI + m]; }
                    } else {
                        fwd2(z, h1, d1, zd, d2);
                    }
                    // what the weight-gradient contraction takes: abar_2, abar_1 and a second factor pair per layer
                    f32x4 ab2[NI], ab1[NH], w[NI], pb2[NI], t1[NH], pb1[NH], tau[NI];
                    if constexpr (GTEST) {
                        // ---- TestMode: Phi = kbar' nn(z) - c_l tr J,  tr J = s'_1' C s'_2 with C = W_1 .* W_2' (closed form).
                        // r = C s'_2, s = C' s'_1;  abar_2 = kbar s'_2 - c_l s''_2 s,  abar_1 = (W_2' abar_2) s'_1 - c_l s''_1 r;
                        // d tr / d W_1 = (s'_1 s'_2') .* W_2', d tr / d W_2 = (s'_2 s'_1') .* W_1': the sum H of c_l s'_1 s'_2' over
                        // stages and samples is contracted like a factor pair and multiplied by the weights once, at the end ----
                        const float c_lv = live ? c_l : 0.f;    // (columns past the batch carry no cotangent)
                        f32x4 r[NH], sv2[NI];
#pragma unroll
                        for (int m = 0; m < NH; ++m) {
                            f32x4 acc = zero4;
#pragma unroll
                            for (int kt = 0; kt < NI; ++kt) acc = mm4(fC[m][kt], d2[kt], acc);
                            r[m] = acc;
                        }
#pragma unroll
                        for (int m = 0; m < NI; ++m) {
                            f32x4 part[NH];
#pragma unroll
                            for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fCT[m][kt], d1[kt], zero4);
                            sv2[m] = part[0];
#pragma unroll
                            for (int kt = 1; kt < NH; ++kt) sv2[m] += part[kt];
                            ab2[m] = kb[m] * d2[m] - c_lv * dd_of(true, zd[m], d2[m]) * sv2[m];
                        }
#pragma unroll
                        for (int m = 0; m < NH; ++m) {
                            f32x4 acc = zero4;
#pragma unroll
                            for (int kt = 0; kt < NI; ++kt) acc = mm4(fW2T[m][kt], ab2[kt], acc);
                            ab1[m] = acc * d1[m] - c_lv * dd_of(false, h1[m], d1[m]) * r[m];
                            t1[m] = c_lv * d1[m];           // second pair of layer 1's slot: (c_l s'_1) x s'_2 -> H
                            pb1[m] = zero4;
                        }
#pragma unroll
                        for (int m = 0; m < NI; ++m) {
                            f32x4 part[NH];
#pragma unroll
                            for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW1T[m][kt], ab1[kt], zero4);
                            w[m] = part[0];
#pragma unroll
                            for (int kt = 1; kt < NH; ++kt) w[m] += part[kt];
                            pb2[m] = d2[m];
                            tau[m] = zero4;
                        }
                    } else {
                    float e2 = 0.f;
#pragma unroll
                    for (int m = 0; m < NI; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) e2 = fmaf(zd[m][j], zd[m][j], e2);
                    f32x4 ahat[NI];
                    {
                        const float nz = nd.norm_z ? sqrtf(wv_quad_sum(e2)) : 0.f;
                        const float sc = (nd.norm_z && nz > 0.f && live) ? c_E / nz : 0.f;
#pragma unroll
                        for (int m = 0; m < NI; ++m) ahat[m] = kb[m] + sc * zd[m];
                    }
                    f32x4 tb1[NH], p1[NH], p2[NI], tb2[NI];
                    if constexpr (MODE == WV_JVP) {
                    // ---- JVP compute mode (src/icnf.jl:384-420): tau = eps, omega = -c_l eps + c_n Je / |Je|.  Tangent sweep first
                    // (p_1 = W_1 eps is the same for every stage: kept from the start of the backward pass), then the tbar chain ----
                    float n2 = 0.f;
#pragma unroll
                    for (int m = 0; m < NH; ++m) { p1[m] = W1eps[m]; t1[m] = p1[m] * d1[m]; }
#pragma unroll
                    for (int m = 0; m < NI; ++m) {
                        f32x4 part[NH];
#pragma unroll
                        for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW2[m][kt], t1[kt], zero4);
                        p2[m] = part[0];
#pragma unroll
                        for (int kt = 1; kt < NH; ++kt) p2[m] += part[kt];
                        tau[m] = ep[m];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float je = p2[m][j] * d2[m][j]; n2 = fmaf(je, je, n2); }
                    }
                    {
                        const float nj = nd.norm_j ? sqrtf(wv_quad_sum(n2)) : 0.f;
                        const float sc = (nd.norm_j && nj > 0.f) ? c_n / nj : 0.f;
#pragma unroll
                        for (int m = 0; m < NI; ++m) { tb2[m] = sc * (p2[m] * d2[m]) - c_l * ep[m]; pb2[m] = tb2[m] * d2[m]; }
                    }
#pragma unroll
                    for (int m = 0; m < NH; ++m) {
                        f32x4 acc = zero4;
#pragma unroll
                        for (int kt = 0; kt < NI; ++kt) acc = mm4(fW2T[m][kt], pb2[kt], acc);
                        tb1[m] = acc;
                        pb1[m] = acc * d1[m];
                    }
                    } else {
                    // ---- reverse sweep of eps (tbar chain: omega = eps): pbar_2 = eps s'_2, tbar_1 = W_2' pbar_2, pbar_1 = tbar_1 s'_1, eJ ----
                    f32x4 eJ[NI];
#pragma unroll
                    for (int m = 0; m < NI; ++m) { tb2[m] = ep[m]; pb2[m] = ep[m] * d2[m]; }
                    float n2 = 0.f;
                    if constexpr (RICH) {                  // tbar_1 and eps'J as the forward pass formed them
#pragma unroll
                        for (int m = 0; m < NH; ++m) { tb1[m] = Rc[2 * NH + m]; pb1[m] = tb1[m] * d1[m]; }
#pragma unroll
                        for (int m = 0; m < NI; ++m) {
                            eJ[m] = Rc[3 * NH + 2 * NI + m];
#pragma unroll
                            for (int j = 0; j < 4; ++j) n2 = fmaf(eJ[m][j], eJ[m][j], n2);
                        }
                    } else {
#pragma unroll
                    for (int m = 0; m < NH; ++m) {
                        f32x4 acc = zero4;
#pragma unroll
                        for (int kt = 0; kt < NI; ++kt) acc = mm4(fW2T[m][kt], pb2[kt], acc);
                        tb1[m] = acc;
                        pb1[m] = acc * d1[m];
                    }
#pragma unroll
                    for (int m = 0; m < NI; ++m) {
                        f32x4 part[NH];
#pragma unroll
                        for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW1T[m][kt], pb1[kt], zero4);
                        eJ[m] = part[0];
#pragma unroll
                        for (int kt = 1; kt < NH; ++kt) eJ[m] += part[kt];
#pragma unroll
                        for (int j = 0; j < 4; ++j) n2 = fmaf(eJ[m][j], eJ[m][j], n2);
                    }
                    }
                    // ---- tau = -c_l eps + c_n eJ / |eJ| ; tangent sweep: p_1 = W_1 tau, t_1 = s'_1 p_1, p_2 = W_2 t_1 ----
                    {
                        const float nj = nd.norm_j ? sqrtf(wv_quad_sum(n2)) : 0.f;
                        const float sc = (nd.norm_j && nj > 0.f) ? c_n / nj : 0.f;
#pragma unroll
                        for (int m = 0; m < NI; ++m) tau[m] = sc * eJ[m] - c_l * ep[m];
                    }
#pragma unroll
                    for (int m = 0; m < NH; ++m) {
                        f32x4 acc = zero4;
#pragma unroll
                        for (int kt = 0; kt < NI; ++kt) acc = mm4(fW1[m][kt], tau[kt], acc);
                        p1[m] = acc;
                        t1[m] = acc * d1[m];
                    }
#pragma unroll
                    for (int m = 0; m < NI; ++m) {
                        f32x4 part[NH];
#pragma unroll
                        for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW2[m][kt], t1[kt], zero4);
                        p2[m] = part[0];
#pragma unroll
                        for (int kt = 1; kt < NH; ++kt) p2[m] += part[kt];
                    }
                    }   // (VJP: tbar chain, tau, tangent sweep)
                    // ---- reverse sweep of the cotangent: abar_l = hbar_l s'_l + tbar_l s''_l p_l ; hbar_{l-1} = W_l' abar_l ----
#pragma unroll
                    for (int m = 0; m < NI; ++m) ab2[m] = ahat[m] * d2[m] + tb2[m] * dd_of(true, zd[m], d2[m]) * p2[m];
#pragma unroll
                    for (int m = 0; m < NH; ++m) {
                        f32x4 acc = zero4;
#pragma unroll
                        for (int kt = 0; kt < NI; ++kt) acc = mm4(fW2T[m][kt], ab2[kt], acc);
                        ab1[m] = acc * d1[m] + tb1[m] * dd_of(false, h1[m], d1[m]) * p1[m];
                    }
#pragma unroll
                    for (int m = 0; m < NI; ++m) {
                        f32x4 part[NH];
#pragma unroll
                        for (int kt = 0; kt < NH; ++kt) part[kt] = mm4(fW1T[m][kt], ab1[kt], zero4);
                        w[m] = part[0];
#pragma unroll
                        for (int kt = 1; kt < NH; ++kt) w[m] += part[kt];
                    }
                    }   // (TrainMode)
#pragma unroll
                    for (int k = 0; k < 6; ++k)
#pragma unroll
                        for (int m = 0; m < NI; ++m) ws[k][m] = i == k ? w[m] : ws[k][m];
                    // ---- weight gradient: Wbar_l += abar_l h_{l-1}' + pbar_l t_{l-1}' over the wave's 16 samples; bbar_l += abar_l ----
                    // slots: [0, NI) abar_2 | [NI, 2NI) pbar_2 | [2NI, 3NI) z | [3NI, 4NI) tau | then NH each: h_1, t_1, abar_1, pbar_1
#pragma unroll
                    for (int m = 0; m < NI; ++m) { tput(m, ab2[m]); tput(NI + m, pb2[m]); tput(2 * NI + m, z[m] + ysv[m]); tput(3 * NI + m, tau[m]); gb2[m] += ab2[m]; }
#pragma unroll
                    for (int m = 0; m < NH; ++m) {
                        tput(4 * NI + m, h1[m]); tput(4 * NI + NH + m, t1[m]); tput(4 * NI + 2 * NH + m, ab1[m]); tput(4 * NI + 3 * NH + m, pb1[m]);
                        gb1[m] += ab1[m];
                    }
                    if constexpr (HELP) {
                        __syncthreads();                   // handed to the helper; the next stage files the other set
                        tbuf = tbuf == tbuf_all ? tbuf_all + TBW : tbuf_all;
                    } else {
                    tsync();
                    {
                        float A2[NI][4], P2[NI][4], Z0[NI][4], T0[NI][4];
#pragma unroll
                        for (int m = 0; m < NI; ++m) { tget(m, A2[m]); tget(NI + m, P2[m]); tget(2 * NI + m, Z0[m]); tget(3 * NI + m, T0[m]); }
#pragma unroll
                        for (int k = 0; k < NH; ++k) {
                            float H1[4], T1[4], A1[4], P1[4];
                            tget(4 * NI + k, H1); tget(4 * NI + NH + k, T1); tget(4 * NI + 2 * NH + k, A1); tget(4 * NI + 3 * NH + k, P1);
#pragma unroll
                            for (int m = 0; m < NI; ++m) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    gW2[m][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[m][j], H1[j], gW2[m][k], 0, 0, 0);
                                    gW1[k][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[j], Z0[m][j], gW1[k][m], 0, 0, 0);
                                }
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    if (GTEST) {               // H[hidden][input] += (c_l s'_1) s'_2'
                                        gH[k][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(T1[j], P2[m][j], gH[k][m], 0, 0, 0);
                                    } else {
                                        gW2[m][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(P2[m][j], T1[j], gW2[m][k], 0, 0, 0);
                                        gW1[k][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(P1[j], T0[m][j], gW1[k][m], 0, 0, 0);
                                    }
                                }
                            }
                        }
                    }
                    tsync();
                    }   // (!HELP)
                }
                // lambda <- lambda + sum_i w_i
#pragma unroll
                for (int m = 0; m < NI; ++m) lam[m] += ((ws[0][m] + ws[1][m]) + (ws[2][m] + ws[3][m])) + (ws[4][m] + ws[5][m]);
            }
            // ---- outputs: d loss / d z(t0), this wave's partial of the flat gradient (Lux order: W out x in column-major, then b) ----
            if (live) {
#pragma unroll
                for (int m = 0; m < NI; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int r = 16 * m + 4 * q + j; if (r < n_in) a.g.lam_out[(size_t)smp * n_in + r] = lam[m][j]; }
            }
            if constexpr (GTEST && !HELP) {
                // Wbar_1[k][i] -= H[k][i] W_2[i][k];  Wbar_2[i][k] -= H[k][i] W_1[k][i]  (H's tile transposed through the LDS buffer)
#pragma unroll
                for (int k = 0; k < NH; ++k)
#pragma unroll
                    for (int m = 0; m < NI; ++m) {
                        tput(0, gH[k][m]);
                        tsync();
                        float Ht[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) Ht[j] = tbuf[(4 * q + j) * 16 + c];     // H[16 k + c][16 m + 4 q + j]
                        tsync();
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            gW1[k][m][j] -= gH[k][m][j] * w2(16 * m + c, 16 * k + 4 * q + j);
                            gW2[m][k][j] -= Ht[j] * w1(16 * k + c, 16 * m + 4 * q + j);
                        }
                    }
            }
            float* gp = a.g.gpart + (size_t)wid * a.g.n_params;
            if constexpr (!HELP) {                         // (HELP: the weight tiles are the helper wave's)
#pragma unroll
            for (int m = 0; m < NI; ++m)
#pragma unroll
                for (int k = 0; k < NH; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!nd.id2) {   // Wbar_2[o][kk]: row o = 16 m + 4 q + j, column kk = 16 k + c   (an APPENDED identity layer has no parameters)
                            const int o = 16 * m + 4 * q + j, kk = 16 * k + c;
                            if (o < n_in && kk < nh) gp[nd.w_off[1] + o + (size_t)kk * n_in] = gW2[m][k][j];
                        }
                        {   // Wbar_1[o][kk]: row o = 16 k + 4 q + j, column kk = 16 m + c
                            const int o = 16 * k + 4 * q + j, kk = 16 * m + c;
                            if (o < nh && kk < n_in + nd.n_cond) gp[nd.w_off[0] + o + (size_t)kk * nh] = gW1[k][m][j];
                        }
                    }
            }
            // bias gradients: the sum over the 16 samples of a lane group's rows (DPP row reduction, fixed order)
            auto row_sum = [&](float v) __attribute__((always_inline)) {
                v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
                v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
                v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
                v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
                return v;
            };
#pragma unroll
            for (int k = 0; k < NH; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = row_sum(gb1[k][j]);
                    const int r = 16 * k + 4 * q + j;
                    if (c == 0 && r < nh) gp[nd.b_off[0] + r] = v;
                }
#pragma unroll
            for (int m = 0; m < NI; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = row_sum(gb2[m][j]);
                    const int r = 16 * m + 4 * q + j;
                    if (!nd.id2 && c == 0 && r < n_in) gp[nd.b_off[1] + r] = v;
                }
        }
    }
    if (wid == 0 && lane == 0) {
        if (GRAD && gover) { ns.done = 0; ns.n_partials = -1; }       // (the caller runs the streamed gradient path)
        if (!alive || __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ns.done = 0; ns.n_partials = -1; }
        __hip_atomic_store(sv.base_dev, mbase + (unsigned)nsync + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ns.cur = 0;
        *a.st_out = ns;
        if (sv.t_out) { sv.t_out[1] += __builtin_amdgcn_s_memrealtime() - sv.t_out[0]; sv.t_out[2] += 1; }
        mirror_store(a.mirror, a.seq, ns);
    }
}

typedef void (*wave_fn)(WaveArgs, Solve3Args, const WvTab);
// one workgroup of up to four waves (B <= 64): the README / regression networks and the one-layer benchmark network, TrainMode
// (VJP) and TestMode, plain solve and gradient
wave_fn pick_wg(int ni, int nh, int mode, bool grad, bool id2, bool tanh2) {
    if (grad) return nullptr;            // (gradients: one tile per workgroup + its helper wave, whatever the batch)
    if (ni != 1 || !(tanh2 || id2)) return nullptr;
    const bool t = mode == WV_TEST;
    if (id2)
        return (nh != 1 || mode == WV_JVP) ? nullptr
             : (t ? (wave_fn)k_solve_wave<1, 1, WV_TEST, true, false, true, 4> : (wave_fn)k_solve_wave<1, 1, WV_VJP, true, false, true, 4>);
#define WV_WG(NHV) (mode == WV_VJP ? (wave_fn)k_solve_wave<1, NHV, WV_VJP, true, false, false, 4> \
                  : mode == WV_JVP ? (wave_fn)k_solve_wave<1, NHV, WV_JVP, true, false, false, 4> : (wave_fn)k_solve_wave<1, NHV, WV_TEST, true, false, false, 4>)
    switch (nh) {
        case 1: return WV_WG(1);
        case 2: return WV_WG(2);
        case 3: return WV_WG(3);
        case 4: return WV_WG(4);
    }
#undef WV_WG
    return nullptr;
}
// the RICH form of the TrainMode / VJP gradient (the forward pass files its intermediates)
wave_fn pick_grad_rich(int nh, bool id2, bool wg) {
    if (wg) return nullptr;
    if (id2) return nh == 1 ? (wave_fn)k_solve_wave<1, 1, WV_VJP, true, true, true, 1, true> : nullptr;
    switch (nh) {
        case 1: return (wave_fn)k_solve_wave<1, 1, WV_VJP, true, true, false, 1, true>;
        case 2: return (wave_fn)k_solve_wave<1, 2, WV_VJP, true, true, false, 1, true>;
        case 3: return (wave_fn)k_solve_wave<1, 3, WV_VJP, true, true, false, 1, true>;
        case 4: return (wave_fn)k_solve_wave<1, 4, WV_VJP, true, true, false, 1, true>;
    }
    return nullptr;
}
wave_fn pick_grad(int ni, int nh, bool id2, bool test = false, bool jvp = false) {
    if (ni != 1) return nullptr;
    if (jvp && !test) {
        if (id2) return nh == 1 ? (wave_fn)k_solve_wave<1, 1, WV_JVP, true, true, true> : nullptr;
        switch (nh) {
            case 1: return (wave_fn)k_solve_wave<1, 1, WV_JVP, true, true>;
            case 2: return (wave_fn)k_solve_wave<1, 2, WV_JVP, true, true>;
            case 3: return (wave_fn)k_solve_wave<1, 3, WV_JVP, true, true>;
            case 4: return (wave_fn)k_solve_wave<1, 4, WV_JVP, true, true>;
            default: return nullptr;
        }
    }
    if (id2) return nh != 1 ? nullptr : test ? (wave_fn)k_solve_wave<1, 1, WV_TEST, true, true, true> : (wave_fn)k_solve_wave<1, 1, WV_VJP, true, true, true>;
    if (test)
        switch (nh) {
            case 1: return (wave_fn)k_solve_wave<1, 1, WV_TEST, true, true>;
            case 2: return (wave_fn)k_solve_wave<1, 2, WV_TEST, true, true>;
            case 3: return (wave_fn)k_solve_wave<1, 3, WV_TEST, true, true>;
            case 4: return (wave_fn)k_solve_wave<1, 4, WV_TEST, true, true>;
            default: return nullptr;
        }
    switch (nh) {
        case 1: return (wave_fn)k_solve_wave<1, 1, WV_VJP, true, true>;
        case 2: return (wave_fn)k_solve_wave<1, 2, WV_VJP, true, true>;
        case 3: return (wave_fn)k_solve_wave<1, 3, WV_VJP, true, true>;
        case 4: return (wave_fn)k_solve_wave<1, 4, WV_VJP, true, true>;
    }
    return nullptr;
}
template <int NI, int NH, bool TANH>
wave_fn pick_mode(int mode) {
    return mode == WV_VJP ? (wave_fn)k_solve_wave<NI, NH, WV_VJP, TANH>
         : mode == WV_JVP ? (wave_fn)k_solve_wave<NI, NH, WV_JVP, TANH> : (wave_fn)k_solve_wave<NI, NH, WV_TEST, TANH>;
}
template <bool TANH>
wave_fn pick_shape_t(int ni, int nh, int mode) {
    if (ni == 1 && nh == 1) return pick_mode<1, 1, TANH>(mode);
    if (ni == 1 && nh == 2) return pick_mode<1, 2, TANH>(mode);
    if (ni == 1 && nh == 3) return pick_mode<1, 3, TANH>(mode);
    if (ni == 1 && nh == 4) return pick_mode<1, 4, TANH>(mode);
    if (ni == 2 && nh == 6) return pick_mode<2, 6, TANH>(mode);
    return nullptr;
}
wave_fn pick_shape(int ni, int nh, int mode, bool tanh2 = true, bool id2 = false) {
    if (id2 && ni == 1 && nh == 1)       // (tanh, identity) with one tile each: the activation compiled out
        return mode == WV_VJP ? (wave_fn)k_solve_wave<1, 1, WV_VJP, true, false, true>
             : mode == WV_JVP ? (wave_fn)k_solve_wave<1, 1, WV_JVP, true, false, true> : (wave_fn)k_solve_wave<1, 1, WV_TEST, true, false, true>;
    return tanh2 ? pick_shape_t<true>(ni, nh, mode) : pick_shape_t<false>(ni, nh, mode);
}
// a one-layer tanh network seen as (tanh layer, identity layer): cnf_abi.hip appends W_2 = I, b_2 = 0 behind the parameters
// ... or a real second layer with the identity activation (PlanarLayer: tanh(w'z + b) u): the same instantiations, its
// gradient is written (nd.id2 = 0)
bool is_id2(const NetDesc& nd) { return nd.n_layers == 2 && nd.acts[0] == 1 && nd.acts[1] == 0; }

}  // namespace

// Two-layer networks whose tile counts have an instantiation (n_in <= 16 with up to 64 hidden units; 32 -> 96 -> 32); every
// activation of the library runs (cnf_act returns sigma' from the forward pass).  One 16-sample tile per wave, one meeting
// word pair per wave: up to 512 x 16 columns.
// Beyond 512 tiles (B > 8192): four tiles per workgroup where that form is instantiated (n_in <= 16, tanh networks, no
// conditioning) -- up to 512 workgroups = 32768 columns, provided the device holds them all at once.
static int wave_xwg_capacity(wave_fn fn) {                 // workgroups of four waves the current device holds at once
    int dev = 0, n_cu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fn, 256, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n_cu * per_cu;
}
static bool wave_xwg_ok(const NetDesc& nd, bool train, int B) {
    if (B > 16 * 4 * 512 || nd.n_cond > 0) return false;
    const int ni = (nd.n_in + 15) / 16, nh = (nd.dims[1] + 15) / 16;
    const int mode = !train ? WV_TEST : (nd.jvp ? WV_JVP : WV_VJP);
    wave_fn fn = pick_wg(ni, nh, mode, false, is_id2(nd), nd.acts[0] == 1 && nd.acts[1] == 1);
    if (!fn) return false;
    static std::mutex mu;
    static std::map<const void*, int> cap;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cap.find((const void*)fn);
    if (it == cap.end()) it = cap.emplace((const void*)fn, wave_xwg_capacity(fn)).first;
    return (B + 63) / 64 <= it->second;
}
bool wave_solve_supported(const NetDesc& nd, bool train, int B) {
    if (nd.n_layers != 2 || B < 1) return false;
    const int ni = (nd.n_in + 15) / 16, nh = (nd.dims[1] + 15) / 16;
    if (pick_shape(ni, nh, 0) == nullptr) return false;
    return B <= 16 * 512 || wave_xwg_ok(nd, train, B);
}

int wave_grad_waves(int B) { return (B + 15) / 16; }
// floats per accepted step of the rich store (0: this network / mode has no RICH form)
size_t wave_grad_rich_floats(const NetDesc& nd, int B, bool train) {
    static const bool off = [] { const char* e = getenv("CNF_WAVE_RICH"); return e && e[0] == '0'; }();
    const int ni = (nd.n_in + 15) / 16, nh = (nd.dims[1] + 15) / 16;
    if (off || !train || nd.jvp || ni != 1 || nh > 4 || (nd.acts[1] != 1 && nh != 1)) return 0;
    return (size_t)6 * wave_grad_waves(B) * 64 * (3 * nh + 3 * ni) * 4;
}
size_t wave_grad_traj_floats(const NetDesc& nd, int B) { return (size_t)6 * wave_grad_waves(B) * 64 * ((nd.n_in + 15) / 16) * 4; }
bool wave_grad_supported(const NetDesc& nd, int B, bool train) {
    static const bool off = [] { const char* e = getenv("CNF_WAVE_GRAD"); return e && e[0] == '0'; }();
    if (off || !wave_solve_supported(nd, train, B)) return false;
    if (nd.acts[0] != 1 || !(nd.acts[1] == 1 || is_id2(nd))) return false;
    if (nd.n_cond > 0 && nd.n_in + nd.n_cond > 16 * ((nd.n_in + 15) / 16)) return false;       // [z; ys] within the input tiles
    return pick_grad((nd.n_in + 15) / 16, (nd.dims[1] + 15) / 16, is_id2(nd)) != nullptr && wave_grad_waves(B) <= 512;
}

cnf_status wave_solve_launch(const NetDesc& nd, bool train, const float* d_params, const float* cond, int cbs, StepState* st_out,
                             float* U0, const float* eps, int B, hipStream_t s, void* mirror, unsigned seq, const Solve3Args& sv_,
                             const WaveGradArgs* grad) {
    if (!wave_solve_supported(nd, train, B)) return CNF_ERR_UNSUPPORTED;
    if (grad && nd.n_cond > 0 && !grad->ys) return CNF_ERR_BAD_ARG;
    if (grad && (!wave_grad_supported(nd, B, train) || !grad->traj || !grad->gpart || !grad->lam_out || !grad->hs_out || grad->traj_cap < 1))
        return CNF_ERR_UNSUPPORTED;
    static const bool off = [] { const char* e = getenv("CNF_PERSISTENT"); const char* w = getenv("CNF_WAVE"); return (e && e[0] == '0') || (w && w[0] == '0'); }();
    if (off) return CNF_ERR_UNSUPPORTED;
    const int ni = (nd.n_in + 15) / 16, nh = (nd.dims[1] + 15) / 16;
    int grid = (B + 15) / 16;
    const int mode = !train ? WV_TEST : (nd.jvp ? WV_JVP : WV_VJP);
    wave_fn fn = grad ? pick_grad(ni, nh, is_id2(nd), !train, nd.jvp != 0) : pick_shape(ni, nh, mode, nd.acts[0] == 1 && nd.acts[1] == 1, is_id2(nd));
    if (!fn || (grid > 512 && grad)) return CNF_ERR_UNSUPPORTED;
    // at most four tiles: the waves of ONE workgroup (they meet through LDS), where that form is instantiated; more than 512
    // tiles: workgroups of four tiles (LDS inside, the tagged words between them: wave_solve_supported has checked that they fit)
    static const bool wg_off = [] { const char* e = getenv("CNF_WAVE_WG"); return e && e[0] == '0'; }();
    // (CNF_WAVE_XWG_MIN=<tiles>: take the four-tile workgroups from that many tiles on -- measurements; the default is what 512
    // meeting words force)
    static const int xwg_min = [] { const char* e = getenv("CNF_WAVE_XWG_MIN"); const int v = e ? atoi(e) : 0; return v > 4 ? v : 513; }();
    const bool xwg = grid > 512 || (grid >= xwg_min && !cond && !grad && pick_wg(ni, nh, mode, false, is_id2(nd), nd.acts[0] == 1 && nd.acts[1] == 1));
    wave_fn wfn = ((grid <= 4 && !wg_off && !cond) || xwg) ? pick_wg(ni, nh, mode, grad != nullptr, is_id2(nd), nd.acts[0] == 1 && nd.acts[1] == 1) : nullptr;
    if (xwg && (!wfn || cond)) return CNF_ERR_UNSUPPORTED;
    const int waves = xwg ? 4 : grid;
    if (grad && grad->rich && mode == WV_VJP && ni == 1) {   // the forward pass files its intermediates: the RICH instantiations
        if (wave_fn r = pick_grad_rich(nh, is_id2(nd), wfn != nullptr)) { if (wfn) wfn = r; else fn = r; }
        else return CNF_ERR_BAD_ARG;                       // (wave_grad_rich_floats said it exists)
    }
    if (wfn) { fn = wfn; grid = xwg ? (grid + 3) / 4 : 1; }
    WaveArgs a{};
    a.nd = nd; a.P = d_params; a.eps = eps; a.cond = cond; a.cbs = cbs; a.B = B;
    a.n_total = (float)((size_t)(nd.n_in + (train ? 3 : 1)) * B);
    a.U0 = U0; a.st_out = st_out; a.mirror = mirror; a.seq = seq;
    if (grad) a.g = *grad;
    Solve3Args sv = sv_;
    sv.nvars = nd.nvars; sv.naugs = nd.naugs; sv.norm_z_aug = nd.norm_z_aug;
    // (a bare solve reads u0 and writes u_out where the caller keeps them; an inference assembles u0 from sv.xs)
    if (!sv.xs && !sv.u0) return CNF_ERR_BAD_ARG;
    WvTab tab = kWvTab;
    void* args[] = {&a, &sv, &tab};
    // (a gradient: the tile's main wave and its helper)
    if (hipLaunchKernel((const void*)fn, dim3(grid), dim3(wfn ? 64 * waves : (grad ? 128 : 64)), args, 0, s) != hipSuccess) {
        (void)hipGetLastError();
        return CNF_ERR_UNSUPPORTED;
    }
    return CNF_OK;
}
