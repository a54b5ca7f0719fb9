// k_adj3b -- the pullback of one Runge-Kutta step of the headline shape 32 -> 128 -> 128 -> 32 (tanh) on SPLIT-bf16 products:
// what k_adj3 (cnf_grad.hip) computes on v_mfma_f32_16x16x4_f32, with every fp32 product formed from six
// v_mfma_f32_16x16x32_bf16 terms on operands split exactly into three bf16 pieces (cnf_split.h) -- the arithmetic of the
// forward kernels k_step3b / k_solve3b (cnf_step3.hip), at 2.7x the matrix rate of the fp32 MFMA.
//
// What it is the derivative of: `augmented_f` in TrainMode / VJP (src/icnf.jl:318-350) inside the Tsit5 step the forward solve
// accepted -- the reference gets it from Enzyme + SciMLSensitivity inside MLJModelInterface.fit
// (src/exts/mlj_ext/core_icnf.jl:59-73).  Algebra: oracle/cnf_grad_oracle.py (rhs_vjp) and the header of k_adj3: per stage, with
// ahat = kbar_z + c_E zdot/|zdot|, omega = eps, tau = -c_l eps + c_n eJ/|eJ|, four sweeps over the three layers --
// forward (h, s', s''), reverse of eps (tbar, pbar = tbar s'), forward tangent of tau (t, q = s'' W t), reverse cotangent
// (abar = hbar s' + tbar q) -- emitting the four factor arrays of the weight gradient (abar_l, pbar_l | h_{l-1}, t_{l-1}) and
// zbar; the six stages of the step run last to first inside the launch, zbar and the running sums stay on the CU.
//
// Layout = k_step3b's: one workgroup of 8 waves per 32 samples; wave w owns the 16-row tile w of both wide layers in both
// directions (A fragments of W1, W2, W3^T, W2^T resident in registers as split pieces for the whole launch: 120 VGPRs), the
// K = 128 operands of the 32-row products (rows of W3 on waves 0-3, rows of W1^T on waves 4-7) are split LDS images; the
// activations travel between layers as XOR-swizzled split images (conflict-free ds_read_b128 operands).  What is new against
// k_adj3: the elementwise state of the WIDE layers (s', s'', tbar: the producing lane is the consuming lane in every sweep)
// lives in REGISTERS (32 VGPRs: h and tbar, s' and s'' formed again from h) instead of 119 KB of LDS -- which is what makes room for the split images -- and the per-row
// state of the 32-row arrays (lambda, the zbar shift register, eps, ahat, s'_3, s''_3) sits in swizzled fp32 LDS rows owned by
// the lanes that produce zdot / zbar (waves w and w + 4 share the addresses: 13 barrier intervals per stage).
#include <type_traits>

#include "cnf_adj3b.h"
#include "cnf_step3_dev.h"
#include "cnf_mfma_dev.h"

#include <cstdlib>

namespace a3b {
constexpr int WS = 256, WP = 32 * WS, WI = 3 * WP;        // K = 128 split image: [piece][32 rows][128 bf16], chunks XOR-swizzled by the row
constexpr int NS = 64, NP = 32 * NS, NI = 3 * NP;         // K = 32
// fp32 area (float offsets).  Owner rows: [32 samples][32 rows], 16-byte chunk c of sample r at chunk position c ^ (r & 7)
constexpr int OWN = 32 * 32;
constexpr int KS = 0;                                      // zbar shift register: 5 owner arrays
constexpr int LAM = KS + 5 * OWN, LSUM = LAM + OWN, EPSA = LSUM + OWN, AHAT = EPSA + OWN, D13 = AHAT + OWN, D23 = D13 + OWN;
constexpr int EJ = D23 + OWN;                              // eJ, then zbar: from waves 4-7 to the owner lanes
constexpr int RED = EJ + OWN;                              // partials [|zdot|^2 | |eJ|^2][32 samples][8]
constexpr int BIAS = RED + 2 * 32 * 8;                     // b1 (128), b2 (128), b3 (32)
constexpr int FP_END = BIAS + 2 * 128 + 32;
constexpr int H1G = FP_END * 4, H2G = H1G + WI, W3I = H2G + WI, W1TI = W3I + WI, X0S = W1TI + WI, G3S = X0S + NI;
constexpr int TOTAL_BYTES = G3S + NI;
static_assert(H1G % 16 == 0 && TOTAL_BYTES <= 160 * 1024, "LDS plan");
static_assert(WI == s3g::WI, "the global image of k_step3b");
}  // namespace a3b

namespace {

#ifdef A3B_STAMPS
#define A3T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); a3acc[i] += t_ - a3last; a3last = t_; } while (0)
#else
#define A3T(i) do {} while (0)
#endif

__device__ __forceinline__ f32x4 a3b_d2tanh4(const f32x4& h, const f32x4& d1) {      // s'' = -2 h s'
    return f32x4{-2.f * h.x * d1.x, -2.f * h.y * d1.y, -2.f * h.z * d1.z, -2.f * h.w * d1.w};
}

// PHASE 0: everything, stage after stage.  The first three sweeps of a stage do not depend on the adjoint state (the tbar chain
// starts from eps, tau is made of eps, eJ and the constant cotangents of the scalar rows): only ahat and the hbar chain behind it
// carry lambda and the zbar of the later stages.  Where the batch leaves CUs idle (launch_adj3b decides) a run of steps is TWO
// launches:
//   PHASE 1, grid (tiles, stages of the run): sweeps 1-3 of ALL stages side by side; HS, PB, TS filed as always; what sweep 4
//            needs is parked in M.park, per stage and workgroup ADJ3B_PARK_FLOATS: the lanes' own registers h_1, h_2,
//            tbar_1 q_1, tbar_2 q_2 (lane order: coalesced both ways) and the owner rows s'_3, eps q_3, c_E zdot / |zdot|;
//   PHASE 2, grid (tiles): per stage the parked state back, abar_3 = ahat s'_3 + eps q_3, the hbar chain (AB), zbar and the
//            bookkeeping of lambda -- four barrier intervals instead of thirteen, and only the transposed weights resident.
template <int PHASE>
__global__ void __launch_bounds__(512, 2)
k_adj3b(NetDesc nd, GradLayout gl, const char* __restrict__ imgb, Adj3bSteps M) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_in = nd.n_in, D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const int t = wave & 1, hf = (wave >> 1) & 1;         // 32-row products: row tile and sample half of this wave
    const bool zown = wave < 4;                           // waves 0-3: the W3 products and the owner rows; waves 4-7: the W1^T products
    const int smp = 16 * hf + s;                          // sample of this lane in the 32-row products
    const int r0 = 16 * t + 4 * q;                        // first of its 4 rows there
    const int nv = n_in - r0;                             // valid rows among them (<= 0 .. >= 4)
    const int b0 = blockIdx.x * 32;
    const bool olive = b0 + smp < M.B;                    // the owner lane's sample exists
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- weights: resident split fragments (as k_step3b), the two K = 128 LDS images by LDS-DMA, biases ----
    constexpr int NCI = 2 * a3b::WI / 16, NCB = (2 * 128 + 32) / 4;
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef const __attribute__((address_space(1))) char* glb_c;
    const f32x4 sgb = reinterpret_cast<const f32x4*>(imgb + s3g::BIASB)[min(tid, NCB - 1)];
    S3bOp wF1, wF2[4], wB3, wB2[4];
    {
        const char* fw = imgb + s3g::F32 + (size_t)wave * 10 * 2048 + 16 * lane;
        constexpr int AH = 5, F0 = PHASE == 2 ? 5 : 0;     // (PHASE 2: the transposed fragments only)
        f32x4 raw[10][2];
#pragma unroll
        for (int f = F0; f < F0 + AH; ++f) { raw[f][0] = *(const f32x4*)(fw + f * 2048); raw[f][1] = *(const f32x4*)(fw + f * 2048 + 1024); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = F0; f < 10; ++f) {
            if (f + AH < 10) {
                raw[f + AH][0] = *(const f32x4*)(fw + (f + AH) * 2048);
                raw[f + AH][1] = *(const f32x4*)(fw + (f + AH) * 2048 + 1024);
            }
            S3bOp o = s3b_split8(raw[f][0], raw[f][1]);
            s3b_pin(o);
            if (f == 0) wF1 = o; else if (f < 5) wF2[f - 1] = o; else if (f == 5) wB3 = o; else wB2[f - 6] = o;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < (NCI + 511) / 512; ++i) {
        const int c = 512 * i + 64 * wave;
        if (c < NCI)
            __builtin_amdgcn_global_load_lds((glb_c)(imgb + s3g::W3I + 16 * (c + lane)), (lds_c)(ldsb + a3b::W3I + 16 * c), 16, 0, 0);
    }
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + a3b::BIAS)[tid] = sgb;

    // ---- addresses (k_step3b's) ----
    const int wb_rd = s * a3b::WS + 16 * (q ^ s);
    const int wb_wr = s * a3b::WS + 16 * ((2 * wave + (q >> 1)) ^ s) + 8 * (q & 1);
    constexpr int HBW = 16 * a3b::WS;
    const char* nrA = ldsb + (zown ? a3b::W3I : a3b::W1TI) + 16 * t * a3b::WS;
    const char* nrB = ldsb + (zown ? a3b::H2G : a3b::H1G) + 16 * hf * a3b::WS;
    const int nsw = (-(s >> 2)) & 3;
    const int nw = smp * a3b::NS + 16 * ((2 * t + (q >> 1)) ^ nsw) + 8 * (q & 1);
    char* x0w = ldsb + a3b::X0S + nw;
    char* g3w = ldsb + a3b::G3S + nw;
    const int nb_rd = s * a3b::NS + 16 * (q ^ nsw);
    constexpr int HBN = 16 * a3b::NS;
    const float* bias = lds + a3b::BIAS;
    // owner rows: this lane's 4 rows of sample smp (waves w and w + 4 share the address: the hand-over of eJ / zbar)
    const int own = smp * 32 + 4 * (((r0 >> 2)) ^ (smp & 7));
    float* redw = lds + a3b::RED + smp * 8 + 4 * t + q;
    auto red8 = [&](int kind) {
        const float* r = lds + a3b::RED + (kind * 32 + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    auto ownp = [&](int arr) -> f32x4* { return reinterpret_cast<f32x4*>(lds + arr + own); };
    // The factor rows of the WIDE layers leave for global memory one interval after they were formed, read back from the split
    // images (the three pieces sum to the fp32 value exactly) by the four waves that IDLE in the next 32-row interval: a wide
    // interval then issues no global store at all.  (On this ISA stores and loads share one in-order counter: a register the
    // allocator spilled is reloaded behind every store in front of it, and with the stores in the wide epilogues each such
    // reload -- the kernel sits at 256 registers -- waited for a write acknowledgement: ~1 k cycles per interval.)
    // 4 waves = 256 lanes cover 32 samples x 128 features: 4 x (4 features) per lane, rows coalesced
    const int fl_li = (wave & 3) * 64 + lane;
    auto flush = [&](int img, float* arr, int row_len, int off) __attribute__((always_inline)) {
        // (the lane's position behind an opaque zero: otherwise every address below is loop invariant, hoisted and spilled)
        int li = fl_li;
        asm volatile("" : "+v"(li));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = li + 256 * j, r = idx >> 5, k = 4 * (idx & 31);
            const f32x4 v = s3b_load4(ldsb + img + r * a3b::WS + 16 * ((k >> 3) ^ (r & 15)) + 2 * (k & 7), a3b::WP);
            if (b0 + r < M.B) st4_wide(arr + (size_t)(b0 + r) * row_len + off + k, v);
        }
    };
    // ... and the rows of the 32-row arrays (x, tau; pbar_3, abar_3) from the K = 32 images likewise: one 4-feature group per lane
    auto flush32 = [&](int img, float* arr, int row_len, int off) __attribute__((always_inline)) {
        int li = fl_li;
        asm volatile("" : "+v"(li));
        const int r = li >> 3, k = 4 * (li & 7);
        const f32x4 v = s3b_load4(ldsb + img + r * a3b::NS + 16 * ((k >> 3) ^ ((-(r >> 2)) & 3)) + 2 * (k & 7), a3b::NP);
        const int cnt = b0 + r < M.B ? n_in - k : 0;
        if (cnt >= 4) st4_wide(arr + (size_t)(b0 + r) * row_len + off + k, v);
        else if (cnt > 0) st4(arr + (size_t)(b0 + r) * row_len + off + k, v, cnt);
    };
    // One 32-row product (K = 128) of this wave: rows 16 t .. of W3 / W1^T against the wave's sample half, both from split images.
    // -DA3B_NARROW_PREFETCH (A/B, round 5): the operands of k-block kb + 1 requested before the MFMAs of k-block kb issue (double
    // buffer) and two PAIRS of accumulators in turn.  Measured: 2.52 against 2.20 ms per gradient at B = 32, 5.4 against 4.2 at
    // 8192 -- the extra live registers push three more weight pieces into scratch (320 against 168 bytes per lane), and every
    // reload waits behind the stores in front of it.  At 256 registers this kernel pays for each one.
    auto narrow = [&]() __attribute__((always_inline)) -> f32x4 {
#ifndef A3B_NARROW_PREFETCH
        f32x4 z0 = zero4, z1 = zero4;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const S3bOp av = s3b_load(nrA + (wb_rd ^ (64 * kb)), a3b::WP), bvv = s3b_load(nrB + (wb_rd ^ (64 * kb)), a3b::WP);
            S3_SB();
            z0 = s3b_term<0>(av, bvv, z0); z1 = s3b_term<3>(av, bvv, z1);
            z0 = s3b_term<1>(av, bvv, z0); z1 = s3b_term<4>(av, bvv, z1);
            z0 = s3b_term<2>(av, bvv, z0); z1 = s3b_term<5>(av, bvv, z1);
            S3_SB();
        }
        return z0 + z1;
#else
        f32x4 z[4] = {zero4, zero4, zero4, zero4};
        S3bOp av = s3b_load(nrA + wb_rd, a3b::WP), bvv = s3b_load(nrB + wb_rd, a3b::WP);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            S3bOp an = av, bn = bvv;
            if (kb + 1 < 4) { an = s3b_load(nrA + (wb_rd ^ (64 * (kb + 1))), a3b::WP); bn = s3b_load(nrB + (wb_rd ^ (64 * (kb + 1))), a3b::WP); }
            S3_SB();
            f32x4& za = z[2 * (kb & 1)]; f32x4& zb_ = z[2 * (kb & 1) + 1];
            za = s3b_term<0>(av, bvv, za); zb_ = s3b_term<3>(av, bvv, zb_);
            za = s3b_term<1>(av, bvv, za); zb_ = s3b_term<4>(av, bvv, zb_);
            za = s3b_term<2>(av, bvv, za); zb_ = s3b_term<5>(av, bvv, zb_);
            S3_SB();
            av = an; bvv = bn;
        }
        return (z[0] + z[1]) + (z[2] + z[3]);
#endif
    };
    // (measured, round 5: waiting here for the stores' acknowledgements -- the flushing waves idle until the barrier anyway -- costs
    // nothing at B <= 2048 and 18 % at B = 8192, where the 226 MB of factor rows per step keep the write path busy: no wait)
#define A3B_FLUSHED() do {} while (0)

    // ---- per launch: eps into the owner rows (waves 0-3 use it); lambda, the running sum, the shift register and the state of
    // the first stage are the business of waves 4-7, which produce zbar: they do the bookkeeping of a finished stage themselves,
    // in the interval that produced its zbar (no interval of its own) ----
    // One stage of one step: where its state and its factor rows are, its step size and the cotangents of its scalar rows
    struct Stage { size_t qi, qo; float cb, hstep; };
    auto stage_of = [&](int step, int stg) __attribute__((always_inline)) -> Stage {
        Stage a;
        const float hs = M.hs[M.step_hi - step];
        const size_t q = (size_t)(M.step_hi - step) * 6 + stg;              // this stage's slot in the factor arrays
        a.qi = q * M.B * gl.sum_in; a.qo = q * M.B * gl.sum_out;           // (floats in front of its rows in HS/TS and in AB/PB)
        a.cb = M.bw[stg]; a.hstep = hs;
        return a;
    };
    // Entering stage `stg` (waves 4-7, behind the zbar `w` of the stage evaluated before it).  Within a step: w joins the running
    // sum and the shift register of kbar_z.  A new step (stg == 5 behind stage 0 of the step after it): lambda <- lambda + the sum
    // over that step's stages of zbar; the shift register is all zero by then (kc[m][d] = 0 past stage 0).
    auto stage_entry = [&](const Stage& a, int stg, const f32x4& xz, const f32x4& w, bool have_w) __attribute__((always_inline)) {
        const int ocnt = olive ? nv : 0;
        f32x4 k0 = zero4, lam = *ownp(a3b::LAM);
        if (have_w && stg < 5) {
            k0 = *ownp(a3b::KS) + M.kc[stg + 1][0] * w;
            *ownp(a3b::LSUM) = *ownp(a3b::LSUM) + w;
#pragma unroll
            for (int d = 1; d < 5; ++d) *ownp(a3b::KS + (d - 1) * a3b::OWN) = *ownp(a3b::KS + d * a3b::OWN) + M.kc[stg + 1][d] * w;
            *ownp(a3b::KS + 4 * a3b::OWN) = zero4;
        } else if (have_w) {
            lam = lam + (*ownp(a3b::LSUM) + w);
            *ownp(a3b::LAM) = lam;
            *ownp(a3b::LSUM) = zero4;
        }
        *ownp(a3b::AHAT) = ld4_mask((a.cb * lam + k0) * a.hstep, ocnt);      // (completed to ahat behind the forward sweep)
        s3b_store4(x0w, a3b::NP, xz);
    };
    // PHASE 1: the one stage of this workgroup
    const int step1 = M.step_hi - (int)blockIdx.y / 6, stg1 = 5 - (int)blockIdx.y % 6;
    {
        const int cnt = olive ? nv : 0;
        if (zown) {
            if (PHASE != 2) *ownp(a3b::EPSA) = ld4(M.eps + (size_t)(b0 + smp) * n_in + r0, cnt);
        } else if (PHASE == 1) {
            s3b_store4(x0w, a3b::NP, ld4(M.traj + (size_t)step1 * M.slot_stride + (size_t)stg1 * M.n + (size_t)(b0 + smp) * D + r0, cnt));
        } else {
            *ownp(a3b::LAM) = ld4(M.lam + (size_t)(b0 + smp) * n_in + r0, cnt);
            *ownp(a3b::LSUM) = zero4;
#pragma unroll
            for (int d = 0; d < 5; ++d) *ownp(a3b::KS + d * a3b::OWN) = zero4;
            const Stage a0 = stage_of(M.step_hi, 5);
            f32x4 x5 = zero4;                              // (PHASE 2 has no forward sweep: the state image is not read)
            if (PHASE == 0) x5 = ld4(M.traj + (size_t)M.step_hi * M.slot_stride + 5 * M.n + (size_t)(b0 + smp) * D + r0, cnt);
            stage_entry(a0, 5, x5, zero4, false);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA pieces have landed
    s3_bar();

    // elementwise state of the wide layers, tile `wave`, halves A / B: lane (q, s) = rows 16 wave + 4q .. of samples s, 16 + s
    // (h of the two wide layers and tbar -- from the tangent sweep on tbar .* q -- : s' = 1 - h^2 and s'' = -2 h s' are formed again where
    // they are used, which is cheaper than the 16 registers they would occupy)
    f32x4 H1r[2], TB1[2], H2r[2], TB2[2];
#ifdef A3B_STAMPS
    unsigned long long a3acc[32] = {0};
    unsigned long long a3last = __builtin_amdgcn_s_memtime();
#endif
    f32x4 xpf = zero4;
    const int wrow = 16 * wave + 4 * q;                    // first of this lane's 4 rows in a wide tile

    // the stages of the steps of this launch, last to first
    // this lane's parked state of (stage slot q, this workgroup): 8 register quads of all lanes, then 3 owner quads of waves 0-3
    auto park_at = [&](int step, int stg) __attribute__((always_inline)) -> float* {
        return M.park + ((size_t)((M.step_hi - step) * 6 + stg) * gridDim.x + blockIdx.x) * ADJ3B_PARK_FLOATS + 4 * tid;
    };
    f32x4 pfr[8], pfo[3];                                  // PHASE 2: the parked state of the stage after this one, in flight
    auto park_load = [&](int step, int stg) __attribute__((always_inline)) {
        const float* pk = park_at(step, stg);
#pragma unroll
        for (int j = 0; j < 8; ++j) pfr[j] = *reinterpret_cast<const f32x4*>(pk + j * 2048);
        if (zown) {
#pragma unroll
            for (int j = 0; j < 3; ++j) pfo[j] = *reinterpret_cast<const f32x4*>(pk + 8 * 2048 + j * 1024);
        }
    };
    if (PHASE == 2) park_load(M.step_hi, 5);
    for (int step = PHASE == 1 ? step1 : M.step_hi, stg = PHASE == 1 ? stg1 : 5;;) {
        const Stage a = stage_of(step, stg);
        const bool last = PHASE == 1 || (stg == 0 && step == M.step_lo);
        const int nstep = stg > 0 ? step : step - 1, nstg = stg > 0 ? stg - 1 : 5;       // the stage evaluated next
        // (an opaque zero per stage keeps the compiler from hoisting the 64-bit addresses of all array families out of the loop)
        int zopq = 0;
        asm volatile("" : "+v"(zopq));
        const int orow = b0 + smp + zopq;
        const int ocnt = olive ? nv : 0;
      if (PHASE != 2) {
        // ---- sweep 1: forward.  I0: layer 1, tile `wave`, both halves (K = 32) ----
        {
            const f32x4 bv = *(const f32x4*)(bias + wrow);
            S3bOp b[2];
            b[0] = s3b_load(ldsb + a3b::X0S + nb_rd, a3b::NP);
            b[1] = s3b_load(ldsb + a3b::X0S + nb_rd + HBN, a3b::NP);
            S3_SB();
            f32x4 acc[2] = {zero4, zero4};
            s3b_mm<2>(acc, wF1, b);
            const f32x4 ha = s3_tanh4(acc[0] + bv), hb = s3_tanh4(acc[1] + bv);
            H1r[0] = ha; H1r[1] = hb;
            s3b_store4(ldsb + a3b::H1G + wb_wr, a3b::WP, ha);
            s3b_store4(ldsb + a3b::H1G + wb_wr + HBW, a3b::WP, hb);
        }
        A3T(0);
        s3_bar();
        A3T(1);
        // I1: layer 2
        {
            const f32x4 bv = *(const f32x4*)(bias + 128 + wrow);
            f32x4 acc[2] = {zero4, zero4};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                S3bOp b[2];
                b[0] = s3b_load(ldsb + a3b::H1G + (wb_rd ^ (64 * kb)), a3b::WP);
                b[1] = s3b_load(ldsb + a3b::H1G + HBW + (wb_rd ^ (64 * kb)), a3b::WP);
                S3_SB();
                s3b_mm<2>(acc, wF2[kb], b);
                S3_SB();
            }
            const f32x4 ha = s3_tanh4(acc[0] + bv), hb = s3_tanh4(acc[1] + bv);
            H2r[0] = ha; H2r[1] = hb;
            s3b_store4(ldsb + a3b::H2G + wb_wr, a3b::WP, ha);
            s3b_store4(ldsb + a3b::H2G + wb_wr + HBW, a3b::WP, hb);
        }
        A3T(2);
        s3_bar();
        A3T(3);
        // I2 (waves 0-3): layer 3: zdot rows r0 .. r0+3 of sample smp; pbar_3 = eps s'_3 -> G3S, PB
        f32x4 zdv = zero4;
        if (zown) {
            const f32x4 bv3 = *(const f32x4*)(bias + 256 + r0);
            const f32x4 zsum = narrow();
            zdv = s3_tanh4(zsum + bv3);                 // padded rows: zero weights and bias -> 0
            const f32x4 d13 = s3_dtanh4(zdv);
            *ownp(a3b::D13) = d13;
            *ownp(a3b::D23) = a3b_d2tanh4(zdv, d13);
            const f32x4 pb = *ownp(a3b::EPSA) * d13;       // (eps is zero in padded rows and beyond the batch)
            s3b_store4(g3w, a3b::NP, pb);
            redw[0] = s3_dot4(zdv, zdv);
        } else {                                           // (waves 4-7: h1, h2 -> HS)
            flush(a3b::H1G, M.HS + a.qi, gl.sum_in, gl.in_off[1]);
            flush(a3b::H2G, M.HS + a.qi, gl.sum_in, gl.in_off[2]);
            flush32(a3b::X0S, M.HS + a.qi, gl.sum_in, 0);         // (the stage state: h_0)
            A3B_FLUSHED();
        }
        A3T(4);
        s3_bar();
        A3T(5);
        // ---- sweep 2: the tbar chain (omega = eps).  I3: W3^T pbar_3 = tbar_2; pbar_2 = tbar_2 s'_2 over h2 in place ----
        {
            S3bOp b[2];
            b[0] = s3b_load(ldsb + a3b::G3S + nb_rd, a3b::NP);
            b[1] = s3b_load(ldsb + a3b::G3S + nb_rd + HBN, a3b::NP);
            S3_SB();
            f32x4 acc[2] = {zero4, zero4};
            s3b_mm<2>(acc, wB3, b);
            TB2[0] = acc[0]; TB2[1] = acc[1];
            const f32x4 pa = acc[0] * s3_dtanh4(H2r[0]), pb = acc[1] * s3_dtanh4(H2r[1]);
            s3b_store4(ldsb + a3b::H2G + wb_wr, a3b::WP, pa);
            s3b_store4(ldsb + a3b::H2G + wb_wr + HBW, a3b::WP, pb);
            if (zown) {                                    // ahat = kbar_z + c_E zdot / |zdot|   (|zdot|^2: complete since the barrier)
                const float nz = red8(0);
                const float inv = (nd.norm_z && nz > 0.f) ? (a.hstep * a.cb * M.lam_E) * __builtin_amdgcn_rsqf(nz) : 0.f;
                if (PHASE == 1) *ownp(a3b::AHAT) = ld4_mask(inv * zdv, ocnt);      // (parked; PHASE 2 adds kbar_z)
                else *ownp(a3b::AHAT) = ld4_mask(*ownp(a3b::AHAT) + inv * zdv, ocnt);
            }
        }
        A3T(6);
        s3_bar();
        A3T(7);
        // I4: W2^T pbar_2 = tbar_1; pbar_1 = tbar_1 s'_1 over h1 in place
        {
            f32x4 acc[2] = {zero4, zero4};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                S3bOp b[2];
                b[0] = s3b_load(ldsb + a3b::H2G + (wb_rd ^ (64 * kb)), a3b::WP);
                b[1] = s3b_load(ldsb + a3b::H2G + HBW + (wb_rd ^ (64 * kb)), a3b::WP);
                S3_SB();
                s3b_mm<2>(acc, wB2[kb], b);
                S3_SB();
            }
            TB1[0] = acc[0]; TB1[1] = acc[1];
            const f32x4 pa = acc[0] * s3_dtanh4(H1r[0]), pb = acc[1] * s3_dtanh4(H1r[1]);
            s3b_store4(ldsb + a3b::H1G + wb_wr, a3b::WP, pa);
            s3b_store4(ldsb + a3b::H1G + wb_wr + HBW, a3b::WP, pb);
        }
        A3T(8);
        s3_bar();
        A3T(9);
        // I5 (waves 4-7): eJ = W1^T pbar_1 -> the owner rows; |eJ|^2 partials
        if (!zown) {
            const f32x4 jsum = narrow();
            const f32x4 ej = ld4_mask(jsum, nv);        // (rows of z only)
            *ownp(a3b::EJ) = ej;
            redw[32 * 8] = s3_dot4(ej, ej);
            // the next stage's state (of this step, or the last stage of the step before it): in flight during sweeps 3 and 4
            if (!last) xpf = ld4(M.traj + (size_t)nstep * M.slot_stride + (size_t)nstg * M.n + (size_t)orow * D + r0, ocnt);
        } else {                                           // (waves 0-3: pbar_2, pbar_1 -> PB)
            flush(a3b::H2G, M.PB + a.qo, gl.sum_out, gl.out_off[1]);
            flush(a3b::H1G, M.PB + a.qo, gl.sum_out, gl.out_off[0]);
            flush32(a3b::G3S, M.PB + a.qo, gl.sum_out, gl.out_off[2]);
            A3B_FLUSHED();
        }
        A3T(10);
        s3_bar();
        A3T(11);
        // E1 (owner lanes): tau = -c_l eps + c_n eJ / |eJ| -> X0S as t_0, TS; the next stage's state is requested
        if (zown) {
            const float nj = red8(1);
            const float inv = (nd.norm_j && nj > 0.f) ? (a.hstep * a.cb * M.lam_n) * __builtin_amdgcn_rsqf(nj) : 0.f;
            const f32x4 tau = ld4_mask(inv * *ownp(a3b::EJ) - (a.hstep * a.cb * M.lam_l) * *ownp(a3b::EPSA), nv);
            s3b_store4(x0w, a3b::NP, tau);
        }
        A3T(12);
        s3_bar();
        A3T(13);
        // ---- sweep 3: the tangent chain.  I0': t_1 = s'_1 (W1 t_0), q_1 = s''_1 (W1 t_0) ----
        {
            S3bOp b[2];
            b[0] = s3b_load(ldsb + a3b::X0S + nb_rd, a3b::NP);
            b[1] = s3b_load(ldsb + a3b::X0S + nb_rd + HBN, a3b::NP);
            S3_SB();
            f32x4 acc[2] = {zero4, zero4};
            s3b_mm<2>(acc, wF1, b);
            const f32x4 da = s3_dtanh4(H1r[0]), db = s3_dtanh4(H1r[1]);
            const f32x4 ta = da * acc[0], tb = db * acc[1];
            TB1[0] = TB1[0] * (a3b_d2tanh4(H1r[0], da) * acc[0]);                    // tbar_1 q_1, q_1 = s''_1 (W1 t_0): all sweep 4 needs of them
            TB1[1] = TB1[1] * (a3b_d2tanh4(H1r[1], db) * acc[1]);
            s3b_store4(ldsb + a3b::H1G + wb_wr, a3b::WP, ta);
            s3b_store4(ldsb + a3b::H1G + wb_wr + HBW, a3b::WP, tb);
        }
        A3T(14);
        s3_bar();
        A3T(15);
        // I1': t_2, q_2
        {
            f32x4 acc[2] = {zero4, zero4};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                S3bOp b[2];
                b[0] = s3b_load(ldsb + a3b::H1G + (wb_rd ^ (64 * kb)), a3b::WP);
                b[1] = s3b_load(ldsb + a3b::H1G + HBW + (wb_rd ^ (64 * kb)), a3b::WP);
                S3_SB();
                s3b_mm<2>(acc, wF2[kb], b);
                S3_SB();
            }
            const f32x4 da = s3_dtanh4(H2r[0]), db = s3_dtanh4(H2r[1]);
            const f32x4 ta = da * acc[0], tb = db * acc[1];
            TB2[0] = TB2[0] * (a3b_d2tanh4(H2r[0], da) * acc[0]);
            TB2[1] = TB2[1] * (a3b_d2tanh4(H2r[1], db) * acc[1]);
            s3b_store4(ldsb + a3b::H2G + wb_wr, a3b::WP, ta);
            s3b_store4(ldsb + a3b::H2G + wb_wr + HBW, a3b::WP, tb);
        }
        A3T(16);
        s3_bar();
        A3T(17);
        // I2' (waves 0-3): abar_3 = ahat s'_3 + eps (s''_3 W3 t_2) -> G3S, AB
        if (zown) {
            const f32x4 zsum = narrow();
            const f32x4 q3 = *ownp(a3b::D23) * zsum;
            if (PHASE == 1) {                              // the owner rows sweep 4 needs -> park
                float* pk = park_at(step, stg) + 8 * 2048;
                *reinterpret_cast<f32x4*>(pk) = *ownp(a3b::D13);
                *reinterpret_cast<f32x4*>(pk + 1024) = *ownp(a3b::EPSA) * q3;
                *reinterpret_cast<f32x4*>(pk + 2048) = *ownp(a3b::AHAT);
            } else {
                const f32x4 ab = *ownp(a3b::AHAT) * *ownp(a3b::D13) + *ownp(a3b::EPSA) * q3;
                s3b_store4(g3w, a3b::NP, ab);
            }
        } else {                                           // (waves 4-7: t_1, t_2 -> TS)
            flush(a3b::H1G, M.TS + a.qi, gl.sum_in, gl.in_off[1]);
            flush(a3b::H2G, M.TS + a.qi, gl.sum_in, gl.in_off[2]);
            flush32(a3b::X0S, M.TS + a.qi, gl.sum_in, 0);         // (tau: t_0)
            A3B_FLUSHED();
        }
        if (PHASE == 1) {                                  // the lanes' registers -> park; the stage is PHASE 2's from here
            float* pk = park_at(step, stg);
            *reinterpret_cast<f32x4*>(pk) = H1r[0]; *reinterpret_cast<f32x4*>(pk + 2048) = H1r[1];
            *reinterpret_cast<f32x4*>(pk + 2 * 2048) = H2r[0]; *reinterpret_cast<f32x4*>(pk + 3 * 2048) = H2r[1];
            *reinterpret_cast<f32x4*>(pk + 4 * 2048) = TB1[0]; *reinterpret_cast<f32x4*>(pk + 5 * 2048) = TB1[1];
            *reinterpret_cast<f32x4*>(pk + 6 * 2048) = TB2[0]; *reinterpret_cast<f32x4*>(pk + 7 * 2048) = TB2[1];
            break;
        }
        A3T(18);
        s3_bar();
        A3T(19);
      } else {
        // ---- PHASE 2: the parked state of this stage (requested a stage ahead: `pfr`, `pfo`) takes its place, the next stage's
        //      is requested and travels under this stage's four intervals; abar_3 = (kbar_z h + c_E zdot / |zdot|) s'_3 + eps q_3
        //      -> G3S.  (Measured: abar_3 formed by waves 4-7 in the I5' that produced the zbar in front of it, into alternating
        //      images -- three intervals per stage instead of four -- is SLOWER, 1.16 against 1.12 ms at B = 32: that interval's
        //      chain, product -> zbar -> bookkeeping -> abar_3 -> split store, grows by more than the interval saved.  Also slower,
        //      1.19 ms: every wave forming the B operand of the W3^T product itself in registers -- ahat's owner rows from LDS, its
        //      own copy of the parked rows, eight splits per lane -- instead of this interval.) ----
        H1r[0] = pfr[0]; H1r[1] = pfr[1]; H2r[0] = pfr[2]; H2r[1] = pfr[3];
        TB1[0] = pfr[4]; TB1[1] = pfr[5]; TB2[0] = pfr[6]; TB2[1] = pfr[7];
        const f32x4 d13 = pfo[0], eq3 = pfo[1], zt = pfo[2];
        if (!last) park_load(nstep, nstg);
        if (zown) s3b_store4(g3w, a3b::NP, (*ownp(a3b::AHAT) + zt) * d13 + eq3);
        s3_bar();
      }
        // ---- sweep 4: the hbar chain.  I3': abar_2 = (W3^T abar_3) s'_2 + tbar_2 q_2 ----
        {
            S3bOp b[2];
            b[0] = s3b_load(ldsb + a3b::G3S + nb_rd, a3b::NP);
            b[1] = s3b_load(ldsb + a3b::G3S + nb_rd + HBN, a3b::NP);
            S3_SB();
            f32x4 acc[2] = {zero4, zero4};
            s3b_mm<2>(acc, wB3, b);
            const f32x4 aa = acc[0] * s3_dtanh4(H2r[0]) + TB2[0], ab = acc[1] * s3_dtanh4(H2r[1]) + TB2[1];
            s3b_store4(ldsb + a3b::H2G + wb_wr, a3b::WP, aa);
            s3b_store4(ldsb + a3b::H2G + wb_wr + HBW, a3b::WP, ab);
        }
        A3T(20);
        s3_bar();
        A3T(21);
        // I4': abar_1 = (W2^T abar_2) s'_1 + tbar_1 q_1
        {
            f32x4 acc[2] = {zero4, zero4};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                S3bOp b[2];
                b[0] = s3b_load(ldsb + a3b::H2G + (wb_rd ^ (64 * kb)), a3b::WP);
                b[1] = s3b_load(ldsb + a3b::H2G + HBW + (wb_rd ^ (64 * kb)), a3b::WP);
                S3_SB();
                s3b_mm<2>(acc, wB2[kb], b);
                S3_SB();
            }
            const f32x4 aa = acc[0] * s3_dtanh4(H1r[0]) + TB1[0], ab = acc[1] * s3_dtanh4(H1r[1]) + TB1[1];
            s3b_store4(ldsb + a3b::H1G + wb_wr, a3b::WP, aa);
            s3b_store4(ldsb + a3b::H1G + wb_wr + HBW, a3b::WP, ab);
        }
        A3T(22);
        s3_bar();
        A3T(23);
        // I5' (waves 4-7): zbar = W1^T abar_1 -> the owner rows (zero beyond the batch and in padded rows)
        if (!zown) {
            const f32x4 jsum = narrow();
            const f32x4 zb = ld4_mask(jsum, ocnt);
            if (!last) stage_entry(stage_of(nstep, nstg), nstg, xpf, zb, true);
            else if (ocnt > 0)                             // the cotangent in front of the first step of the run
                st4(M.lam_out + (size_t)orow * n_in + r0, *ownp(a3b::LAM) + (*ownp(a3b::LSUM) + zb), ocnt);
        } else {                                           // (waves 0-3: abar_2, abar_1 -> AB)
            flush(a3b::H2G, M.AB + a.qo, gl.sum_out, gl.out_off[1]);
            flush(a3b::H1G, M.AB + a.qo, gl.sum_out, gl.out_off[0]);
            flush32(a3b::G3S, M.AB + a.qo, gl.sum_out, gl.out_off[2]);
            A3B_FLUSHED();
        }
        A3T(24);
        s3_bar();
        A3T(25);
        if (last) break;
        step = nstep; stg = nstg;
    }
#ifdef A3B_STAMPS
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 5))
        printf("k_adj3b wave %d, cycles per step (work | barrier wait) per interval: I0 %llu|%llu I1 %llu|%llu I2 %llu|%llu I3 %llu|%llu I4 %llu|%llu I5 %llu|%llu E1 %llu|%llu "
               "I0' %llu|%llu I1' %llu|%llu I2' %llu|%llu I3' %llu|%llu I4' %llu|%llu I5' %llu|%llu\n", wave,
               a3acc[0], a3acc[1], a3acc[2], a3acc[3], a3acc[4], a3acc[5], a3acc[6], a3acc[7], a3acc[8], a3acc[9], a3acc[10], a3acc[11], a3acc[12], a3acc[13],
               a3acc[14], a3acc[15], a3acc[16], a3acc[17], a3acc[18], a3acc[19], a3acc[20], a3acc[21], a3acc[22], a3acc[23], a3acc[24], a3acc[25]);
#endif
}

}  // namespace

bool adj3b_supported(const NetDesc& nd) {
    static const bool off = [] { const char* e = getenv("CNF_ADJ3B"); return e && e[0] == '0'; }();
    if (off || nd.n_layers != 3 || nd.n_cond > 0 || nd.jvp) return false;
    if (nd.dims[0] > 32 || nd.dims[1] != 128 || nd.dims[2] != 128 || nd.dims[3] > 32 || nd.dims[0] != nd.dims[3]) return false;
    if (nd.n_in != nd.dims[0]) return false;
    for (int l = 0; l < 3; ++l) if (nd.acts[l] != 1) return false;
    return true;
}

bool adj3b_split(int B, int steps) {
    const int forced = adj_split_mode();                  // A/B and tests: 0 never, 1 always
    if (forced == 0 || B < 1 || steps < 1) return false;
    if (forced == 1) return true;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        if (cus < 1) cus = 1;
    }
    // in units of one stage of the one-launch form: the parallel launch costs ~0.75 per round of `cus` workgroups, the sequential
    // one ~0.3 per stage (4 of 13 barrier intervals)
    const long nst = 6L * steps, rounds = (nst * ((B + 31) / 32) + cus - 1) / cus;
    return 0.75 * rounds + 0.3 * nst < 0.95 * nst;
}

size_t adj3b_park_floats(int B, int steps) { return (size_t)6 * steps * ((B + 31) / 32) * ADJ3B_PARK_FLOATS; }

hipError_t launch_adj3b(const NetDesc& nd, const GradLayout& g, const void* d_img3b, const Adj3bSteps& M, hipStream_t s) {
    if (!adj3b_supported(nd) || !d_img3b || !M.lam_out || M.step_lo < 0 || M.step_hi < M.step_lo ||
        M.step_hi - M.step_lo >= ADJ3B_MAX_STEPS || M.B < 1) return hipErrorInvalidValue;
    const int tiles = (M.B + 31) / 32, steps = M.step_hi - M.step_lo + 1;
    // (per launch: the attribute belongs to the current device, and a process may drive several)
    auto go = [&](auto phase_c, dim3 grid) -> hipError_t {
        constexpr int PH = decltype(phase_c)::value;
        hipError_t e = hipFuncSetAttribute((const void*)k_adj3b<PH>, hipFuncAttributeMaxDynamicSharedMemorySize, a3b::TOTAL_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_adj3b<PH>), grid, dim3(512), a3b::TOTAL_BYTES, s, nd, g, (const char*)d_img3b, M);
        return hipGetLastError();
    };
    if (!M.park || !adj3b_split(M.B, steps)) return go(std::integral_constant<int, 0>{}, dim3(tiles));
    hipError_t e = go(std::integral_constant<int, 1>{}, dim3(tiles, 6 * steps));
    if (e != hipSuccess) return e;
    return go(std::integral_constant<int, 2>{}, dim3(tiles));
}
