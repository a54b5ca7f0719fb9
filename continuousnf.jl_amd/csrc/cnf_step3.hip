// k_step3 -- the Tsit5 step kernel of the headline shape 32 -> 128 -> 128 -> 32, tanh (BASELINE configs 3 and
// 4; any net that pads to it): one launch = one step attempt = 6 evaluations of augmented_f
// (src/icnf.jl:318-350: Dense+tanh forward, reverse sweep of eps, trace and norm rows) + the stage
// combinations + the embedded error estimate + the controller of the previous attempt.  Same arithmetic and
// interface as k_mfma<LyCfg3, true> (cnf_mfma.hip), different schedule:
//
//  * WEIGHTS LIVE IN REGISTERS for the whole launch.  Wave w owns output rows 16w..16w+15 of both wide layers in
//    both sweeps; its A operands (16 rows of W2, of W3^T, of W2^T: 32 + 8 + 32 VGPRs, plus W1 rows on waves 0-3)
//    never change, so they are loaded once from a fragment-ordered image (one coalesced 1 KiB load per
//    fragment) and no weight passes through LDS: no column walks (4 x ds_read_b32) in the reverse sweep, no
//    bank conflicts, and LDS holds activations only.
//  * THE TWO 16-SAMPLE HALVES OF THE TILE ARE PIPELINED AGAINST EACH OTHER.  Every wave computes half A, then
//    half B while A's epilogue (bias, tanh, store) retires under B's MFMAs; the barrier that publishes A's
//    activations sits between "MFMAs of B" and "epilogue of B", and the next layer's half A starts behind it.
//    Each wide product keeps its last two k-blocks for AFTER the next barrier: the wave first requests the
//    operands that barrier has just published, then issues those 8 MFMAs while the requests are in flight, so
//    the LDS round trip behind a barrier is covered by MFMA work that did not depend on it.
//  * The narrow products run one wave per SIMD: W3 h2 (zdot) on waves 0-3; W1^T g1 (eps^T J) on waves 4-7,
//    concurrently with the first layer of the NEXT evaluation on waves 0-3 (it only needs the new stage state).
//  * Runge-Kutta state of the z rows: u, k1 and the running stage sum in the lanes that produce zdot (waves
//    0-3), k2..k7 in LDS (written and read back by the same lane); the three scalar rows in LDS.
// LDS images of k_step3 (fp32) are [sample][feature] with row strides == 8 (mod 16) floats: conflict-free ds_read_b128 B
// operands; an accumulator tile is stored with one ds_write_b128 per lane (lane = sample, 4 rows).
// The split-bf16 kernels (k_step3b / k_solve3b, below: namespace s3v) keep XOR-swizzled unpadded bf16 rows instead: their
// operand READS are conflict-free as well, their 8-byte epilogue STORES are two-way by construction (the 16 lanes of an MFMA
// accumulator row group all write the same half of their 16-byte chunks: positions p and p + 8 share banks).  Measured by
// ablation under rocprofv3 (tools/lds_conflict_ablation.sh, profiles/round5_headline_lds_conflicts.md): 69 % of the launch's
// SQ_LDS_BANK_CONFLICT cycles are those stores (exactly half of their LDS cycles), none are operand reads, the rest are the
// fp32 Runge-Kutta rows; the LDS pipe is 35 % busy, so the conflicts cost 5.6 % of ITS time and nothing measurable of the launch's.
#include "cnf_step3_dev.h"
#include <mutex>

namespace s3 {
constexpr int NB = 32, P0 = 32, PH = 128;
constexpr int SX0 = 40, SXH = 136, SKZ = 264, SW2 = 132;
// LDS plan (floats).  Permanent: the two K = 128 weight images of the narrow products; then the activation area,
// which the W2 staging image aliases during the prologue.
constexpr int W1T = 0;                         // rows k of W1^T (k < 32), 128 outs            [32][136]
constexpr int W3R = W1T + P0 * SXH;            // rows o of W3 (o < 32), 128 ins               [32][136]
constexpr int X0 = W3R + P0 * SXH;             // stage state z                                [32][40]
constexpr int H1 = X0 + NB * SX0;              // h1                                           [32][136]
constexpr int H2 = H1 + NB * SXH;              // h2, then g2 in place                         [32][136]
constexpr int G1 = H2 + NB * SXH;              // g1                                           [32][136]
constexpr int G3 = G1 + NB * SXH;              // g3 = eps .* sigma'(h3)                       [32][40]
constexpr int KZ = G3 + NB * SX0;              // z rows: u, k1, k2..k7                        [32][8*32 (+8)]
constexpr int RED = KZ + NB * SKZ;             // partials [e2 | ld | n2][32 samples][8 = 2 row tiles x 4 q]
constexpr int SC = RED + 3 * NB * 8;           // scalar-row Runge-Kutta state [32][8][3]
constexpr int EPS = SC + NB * 24;              // the probe rows eps                           [32][40]
constexpr int BIAS = EPS + NB * SX0;           // b1 (128), b2 (128), b3 (32)
constexpr int MISC = BIAS + 2 * PH + P0;       // controller scratch, block reductions (64 words)
constexpr int TOTAL = MISC + 64;
constexpr int STG = X0;                        // prologue only: W2 row-major, stride 132     [128][132]
static_assert(STG + PH * SW2 <= RED, "the W2 staging image must not reach the scratch that the prologue uses");
// global image (floats): the LDS images as they are stored, then the register fragments read straight from memory
constexpr int IMG_W2 = 0;                      // [128][132]
constexpr int IMG_W1T = IMG_W2 + PH * SW2;     // [32][136]   (W1T and W3R are contiguous: one copy loop)
constexpr int IMG_W3R = IMG_W1T + P0 * SXH;    // [32][136]
constexpr int IMG_B1 = IMG_W3R + P0 * SXH, IMG_B2 = IMG_B1 + PH, IMG_B3 = IMG_B2 + PH;
constexpr int IMG_FR1 = IMG_B3 + P0;           // W1 row fragments, waves 0-3: 4 per wave (tile w: k-blocks 0,1; tile w+4: 0,1)
constexpr int IMG_FR3 = IMG_FR1 + 4 * 4 * 256; // W3^T row fragments, waves 0-7: 2 per wave
constexpr int IMG_FLOATS = IMG_FR3 + 8 * 2 * 256;
}  // namespace s3

template <int NU, int NB_>
__device__ __forceinline__ void s3_load(f32x4 (&b)[NB_], const float* xb, int boff = 0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) b[boff + u] = *(const f32x4*)(xb + 16 * u);
}
// k-blocks [U0, U1) of one 16x16 output tile: A from registers, B from registers; two accumulation chains (even /
// odd k-steps) so that a wave alone on its SIMD can issue back to back (dependent latency 40 cycles > issue 32)
template <int U0, int U1, int NW, int NB_>
__device__ __forceinline__ void s3_mm(f32x4& a0, f32x4& a1, const f32x4 (&wf)[NW], const f32x4 (&b)[NB_], int woff = 0,
                                      int boff = 0) {
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[woff + u].x, b[boff + u].x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[woff + u].y, b[boff + u].y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[woff + u].z, b[boff + u].z, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[woff + u].w, b[boff + u].w, a1, 0, 0, 0);
    }
}
// Behind a barrier: the 8 operand requests of the next product, issued in the shadow of 4 MFMAs that are ready to go
// (operands in registers): MFMA, two requests, MFMA, two requests, ...  An in-order wave that issued the requests
// first would leave the matrix pipe idle for their issue time on both waves of the SIMD at once.
template <int UW, int UB, int NW, int NB_>
__device__ __forceinline__ void s3_mm_load(f32x4& a0, f32x4& a1, const f32x4 (&wf)[NW], const f32x4 (&b)[NB_], f32x4 (&nb)[8],
                                           const float* xb) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[UW].x, b[UB].x, a0, 0, 0, 0);
    nb[0] = *(const f32x4*)(xb); nb[1] = *(const f32x4*)(xb + 16);
    S3_SB();
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[UW].y, b[UB].y, a1, 0, 0, 0);
    nb[2] = *(const f32x4*)(xb + 32); nb[3] = *(const f32x4*)(xb + 48);
    S3_SB();
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[UW].z, b[UB].z, a0, 0, 0, 0);
    nb[4] = *(const f32x4*)(xb + 64); nb[5] = *(const f32x4*)(xb + 80);
    S3_SB();
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[UW].w, b[UB].w, a1, 0, 0, 0);
    nb[6] = *(const f32x4*)(xb + 96); nb[7] = *(const f32x4*)(xb + 112);
    S3_SB();
}
#ifdef S3_STAMPS
#define S3T(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                    s3acc[i] += t_ - s3last; s3last = t_; } while (0)
#else
#define S3T(i) do {} while (0)
#endif

// Kernel entry, shared by the four step kernels.  Every memory round trip in front of the weight stream is exposed (the
// caches start cold), and the compiler fetches kernel arguments lazily, each batch behind its own wait: all arguments the
// prologue uses are demanded at once (one batch, one wait), and the integrator state words are read with VECTOR loads (an
// opaque zero in the address), which queue with the other requests instead of stalling the wave as scalar loads do.
#define S3_ARGS_UP_FRONT(a, img)                                                                                        \
    asm volatile("" ::"s"(a.st), "s"(a.partials_in), "s"(a.eps), "s"(a.U[0]), "s"(a.U[1]), "s"(a.K1[0]), "s"(a.K1[1]),   \
                 "s"(a.B), "s"(a.apply_ctrl), "s"(img), "s"((int)gridDim.x))
__device__ __forceinline__ const StepState* s3_state_words(const StepState* st) {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return st + z;
}

__global__ void __launch_bounds__(512, 2) k_step3(MfmaArgs a, const float* __restrict__ img3, int n_in, int norm_z,
                                                  int norm_j, const S3Tab tab, int single) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    S3_ARGS_UP_FRONT(a, img3);
    const StepState* st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const int t = wave & 1, hf = (wave >> 1) & 1;         // narrow phases: row tile and sample half of this wave
    const bool zown = wave < 4;                           // waves 0-3 produce zdot: they hold the z rows of the state
    const bool sown = !zown && t == 0 && q == 0;          // waves 4, 6: lane s holds the scalar rows of sample 16 hf + s
    const int smp = 16 * hf + s;                          // sample of this lane in the narrow phases
    const int r0 = 16 * t + 4 * q;                        // first of its 4 rows there
    const int nv = n_in - r0;                             // valid rows among them (may be <= 0 or > 4)
#ifdef S3_STAMPS
    unsigned long long s3acc[36] = {0};
    unsigned long long s3last = __builtin_amdgcn_s_memtime();
    const unsigned long long s3start = s3last;
    const unsigned long long s3rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- everything the launch needs from memory is requested up front, in ONE round trip, and nothing is consumed
    // before all of it is in flight.  Order of issue = order of return: the integrator state words first (the `done`
    // test and the controller need them soonest), then the error partials, the weight stream (135 KB per workgroup:
    // it bounds the prologue, so it must not queue behind anything that waits), then this workgroup's first tile from
    // BOTH buffer sets (which one is current is the controller's decision).
    const StepState* stw = s3_state_words(st);
    const int v_done = stw->done, v_cur = stw->cur;
    const float v_h = stw->h, v_abstol = stw->abstol, v_reltol = stw->reltol;
    // (one unconditional load per thread: a loop here would wait for its data before anything below is even requested)
    const float* ppin = a.apply_ctrl ? a.partials_in : img3;
    const float2 pp = *reinterpret_cast<const float2*>(ppin + 2 * min(tid, (int)gridDim.x - 1));
    // Weights.  W2 is needed twice per wave (16 of its rows forward, 16 of its columns in reverse): it is read from
    // memory ONCE per workgroup into an LDS staging image (which aliases the activation area, idle until the first
    // stage) and the 8 waves cut their two register fragments out of it.  The K = 128 images of the narrow products
    // (W3 rows, W1^T rows) stay in LDS for the whole launch; the K = 32 fragments (W1 rows, W3^T rows: 6 per wave)
    // come straight from memory.  135 KB per workgroup from L2 instead of 240 KB for all-register fragments.
    // The images travel by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, LDS destination = wave-uniform
    // base + 16 lane, no registers in between); the copies are contiguous because memory holds the LDS layout.
    constexpr int NC2 = s3::PH * s3::SW2 / 4, NCN = 2 * s3::P0 * s3::SXH / 4, NCB = (2 * s3::PH + s3::P0) / 4;
    static_assert(NC2 % 64 == 0 && NCN % 64 == 0, "whole wave instructions");
    {
        typedef __attribute__((address_space(3))) float* lds_f;
        typedef const __attribute__((address_space(1))) float* glb_f;
#pragma unroll
        for (int i = 0; i < (NC2 + 511) / 512; ++i) {
            const int c = 512 * i + 64 * wave;                 // wave-uniform chunk (16 B) index
            if (c < NC2)
                __builtin_amdgcn_global_load_lds((glb_f)(img3 + s3::IMG_W2 + 4 * (c + lane)), (lds_f)(lds + s3::STG + 4 * c), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < (NCN + 511) / 512; ++i) {
            const int c = 512 * i + 64 * wave;
            if (c < NCN)
                __builtin_amdgcn_global_load_lds((glb_f)(img3 + s3::IMG_W1T + 4 * (c + lane)), (lds_f)(lds + s3::W1T + 4 * c), 16, 0, 0);
        }
    }
    const f32x4 sgb = reinterpret_cast<const f32x4*>(img3 + s3::IMG_B1)[min(tid, NCB - 1)];
    f32x4 wF1[4], wF2[8], wB3[2], wB2[8];
    // the K = 32 fragments (6 per wave) straight from memory: on their way with the images
    if (zown) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(img3 + s3::IMG_FR1) + (size_t)wave * 4 * 64 + lane;
#pragma unroll
        for (int u = 0; u < 4; ++u) wF1[u] = wp[u * 64];
    }
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(img3 + s3::IMG_FR3) + (size_t)wave * 2 * 64 + lane;
#pragma unroll
        for (int u = 0; u < 2; ++u) wB3[u] = wp[u * 64];
    }
    const int ntile = (a.B + s3::NB - 1) / s3::NB;
    const bool wide = (n_in & 3) == 0;                     // every lane's four rows are all valid or all padding
    f32x4 ru[2], rk[2], re, rs[2][2];
    int ce = 0, cu = 0, cs = 0;
    float* sc = lds + s3::SC + smp * 24;
    auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
    auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
    {
        const int b0 = blockIdx.x * s3::NB + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
        re = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ru[c] = ld4_issue_w(a.U[c] + gcol + r0, cu, img3, wide);
            rk[c] = ld4_issue_w(a.K1[c] + gcol + r0, cu, img3, wide);
            rs[c][0] = ld3_issue(a.U[c] + gcol + n_in, cs, img3);
            rs[c][1] = ld3_issue(a.K1[c] + gcol + n_in, cs, img3);
        }
    }
    __builtin_amdgcn_sched_barrier(0);                     // nothing above is consumed before all of it is requested
    // (wave-uniform values into scalar registers: the buffer pointers selected from `cur` stay out of the vector file)
    const int st_done = __builtin_amdgcn_readfirstlane(v_done), st_cur = __builtin_amdgcn_readfirstlane(v_cur);
    const float st_h = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_h)));
    const float st_abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_abstol)));
    const float st_reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_reltol)));
    S3T(32);
    if (st_done) {       // launches queued past the end of the solve: keep the state chain intact and leave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA pieces must have landed before the wave ends)
        if (a.apply_ctrl && blockIdx.x == 0 && tid == 0) { *a.st_out = *st; publish_mirror(a, *st); }
        return;
    }
    // scalar rows of both candidates wait in LDS (slots 2..5 of the scalar-row state) for the controller's choice
    if (sown) {
        sc_set(2, ld4_mask(rs[0][0], cs)); sc_set(3, ld4_mask(rs[0][1], cs));
        sc_set(4, ld4_mask(rs[1][0], cs)); sc_set(5, ld4_mask(rs[1][1], cs));
    }
    // the probe rows go straight to their LDS image (waves 0-3 need them for g3, waves 4-7 for the trace row)
    *(f32x4*)(lds + s3::EPS + smp * s3::SX0 + r0) = ld4_mask(re, ce);
    float* msc = lds + s3::MISC;
    S3T(33);
    if (a.apply_ctrl) {
        float cp0 = tid < (int)gridDim.x ? pp.x : 0.f, cp1 = tid < (int)gridDim.x ? pp.y : 0.f;
        for (int i = tid + 512; i < (int)gridDim.x; i += 512) { cp0 += a.partials_in[2 * i]; cp1 += a.partials_in[2 * i + 1]; }
        cp0 = s3_wave_sum(cp0); cp1 = s3_wave_sum(cp1);
        if (lane == 0) { msc[wave] = cp0; msc[16 + wave] = cp1; }
    }
    S3T(34);
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + s3::BIAS)[tid] = sgb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA pieces have landed (then the barrier)
    S3T(24);
    s3_bar();                                              // staging image and partial sums complete
    S3T(25);
    int cur = st_cur;
    float hstep = st_h, abstol = st_abstol, reltol = st_reltol;
    if (a.apply_ctrl && tid == 0) {
        // In-kernel step controller (as in k_mfma): every workgroup reduces the same partials in the same order
        // and takes the same decision; block 0 publishes the new state for the next launch and the host mirror.
        float p0 = 0.f, p1 = 0.f;
        for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
        StepState ns = *st;                                // (cached by now: the state words above came from these lines)
        ctrl_after_step(&ns, p0, p1, a.n_total);
        if (blockIdx.x == 0) { *a.st_out = ns; publish_mirror(a, ns); }
        msc[32] = __int_as_float(ns.cur); msc[33] = ns.h; msc[34] = ns.abstol; msc[35] = ns.reltol;
        msc[36] = __int_as_float(ns.done);
    }
    {   // this wave's fragments of W2: rows 16w + s (b128 along the row), columns 16w + s (4 x b32 down the column)
        const float* rw = lds + s3::STG + (16 * wave + s) * s3::SW2 + 4 * q;
        const float* cw = lds + s3::STG + (4 * q) * s3::SW2 + 16 * wave + s;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            wF2[u] = *(const f32x4*)(rw + 16 * u);
            const float* c_ = cw + 16 * u * s3::SW2;
            wB2[u] = f32x4{c_[0], c_[s3::SW2], c_[2 * s3::SW2], c_[3 * s3::SW2]};
        }
    }
    S3T(26);
    s3_bar();                                              // controller done; the staging area is free
    S3T(27);
    if (a.apply_ctrl) {
        cur = __builtin_amdgcn_readfirstlane(__float_as_int(msc[32]));
        hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[33])));
        abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[34])));
        reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[35])));
        if (__float_as_int(msc[36])) return;        // the controller just finished the solve
    }
    // (explicit selects: indexing the kernel-argument arrays with a run-time value makes the compiler fetch the
    // pointer from the argument segment through a vector load -- a memory round trip in front of the state loads)
    // single != 0: ONE evaluation instead of a step attempt -- the two launches of the automatic initial dt
    // (1: f(u) -> a.du with the norms of phase 0; 2: f(u + h k1) -> a.Ks0 with the norm of phase 1), same prologue,
    // same evaluation code; the last workgroup to finish runs the controller phase (as k_mfma does for them)
    const int nstg = single ? 1 : 6;
    const float c21 = single == 1 ? 0.f : (single == 2 ? 1.f : TS_A21);
    const float* Uin = cur ? a.U[1] : a.U[0];
    const float* K1in = cur ? a.K1[1] : a.K1[0];
    float* Uout = cur ? a.U[0] : a.U[1];
    float* K1out = cur ? a.K1[0] : a.K1[1];

    float errsum = 0.f, badcnt = 0.f;
    // the 8 partials (2 row tiles x 4 lanes) of one sample and one kind sit side by side: two b128 reads each
    auto red8 = [&](int kind) {
        const float* r = lds + s3::RED + (kind * s3::NB + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    auto read_scalars = [&]() {
        const float e2 = red8(0), ld = red8(1), n2 = red8(2);
        return f32x4{ld, norm_z ? __builtin_sqrtf(e2) : 0.f, norm_j ? __builtin_sqrtf(n2) : 0.f, 0.f};
    };
    float* redw = lds + s3::RED + smp * 8 + 4 * t + q;                 // this lane's slot of kind 0 (+ 256 per kind)
    // operand / result addresses of the wide phases (lane = sample s of half A, 4 rows 16 wave + 4q ..; half B = +16 rows)
    const float* x0r = lds + s3::X0 + s * s3::SX0 + 4 * q;
    const float* g3r = lds + s3::G3 + s * s3::SX0 + 4 * q;
    const float* h1r = lds + s3::H1 + s * s3::SXH + 4 * q;
    const float* h2r = lds + s3::H2 + s * s3::SXH + 4 * q;
    float* h1w = lds + s3::H1 + s * s3::SXH + 16 * wave + 4 * q;
    float* h2w = lds + s3::H2 + s * s3::SXH + 16 * wave + 4 * q;
    float* g1w = lds + s3::G1 + s * s3::SXH + 16 * wave + 4 * q;
    constexpr int HB = 16 * s3::SXH, XB = 16 * s3::SX0;              // half B = 16 samples on
    const float* nrH2 = lds + s3::H2 + smp * s3::SXH + 4 * q;         // narrow phases: B operands of this wave's half
    const float* nrG1 = lds + s3::G1 + smp * s3::SXH + 4 * q;
    // ... and their A operands: row 16t + s of W3 (waves 0-3) / of W1^T (waves 4-7)
    const float* nrW = lds + (zown ? s3::W3R : s3::W1T) + (16 * t + s) * s3::SXH + 4 * q;
    float* x0w = lds + s3::X0 + smp * s3::SX0 + r0;
    float* g3w = lds + s3::G3 + smp * s3::SX0 + r0;
    // Runge-Kutta state of the z rows r0..r0+3 of sample smp, written and read by this lane only:
    float* rkw = lds + s3::KZ + smp * s3::SKZ + r0;                   // u at rkw, k1 at rkw + 32,
    float* kzw = rkw + 64;                                            // k_{j+2} at kzw + 32 j (j = 0..5)
    float* epw = lds + s3::EPS + smp * s3::SX0 + r0;                  // the probe rows eps
    const float* bias = lds + s3::BIAS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b0 = tile * s3::NB + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        f32x4 uz, k1z;
        if (tile == blockIdx.x) {                              // requested at kernel entry
            uz = ld4_mask(cur ? ru[1] : ru[0], cu);
            k1z = ld4_mask(cur ? rk[1] : rk[0], cu);
            if (sown) { sc_set(0, sc_get(cur ? 4 : 2)); sc_set(1, sc_get(cur ? 5 : 3)); }
            if (single == 1) { k1z = f32x4{0.f, 0.f, 0.f, 0.f}; if (sown) sc_set(1, k1z); }     // there is no k1 yet
        } else {
            ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
            const f32x4 e_ = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
            const f32x4 u_ = ld4_issue_w(Uin + gcol + r0, cu, img3, wide), k_ = ld4_issue_w(K1in + gcol + r0, cu, img3, wide);
            const f32x4 s0 = ld3_issue(Uin + gcol + n_in, cs, img3), s1 = ld3_issue(K1in + gcol + n_in, cs, img3);
            *(f32x4*)epw = ld4_mask(e_, ce); uz = ld4_mask(u_, cu); k1z = ld4_mask(k_, cu);
            if (sown) { sc_set(0, ld4_mask(s0, cs)); sc_set(1, ld4_mask(s1, cs)); }
            if (single == 1) { k1z = f32x4{0.f, 0.f, 0.f, 0.f}; if (sown) sc_set(1, k1z); }
        }
        if (zown) {
            *(f32x4*)x0w = uz + (hstep * c21) * k1z;      // state of evaluation 1: U_2 = u + h a21 k1
            *(f32x4*)rkw = uz;
            *(f32x4*)(rkw + 32) = k1z;
#pragma unroll
            for (int j = 0; j < 6; ++j) *(f32x4*)(kzw + 32 * j) = zero4;       // k2..k7: not produced yet
        }
        S3T(28);
        s3_bar();
        S3T(20);

        // eJ = W1^T g1 on waves 4-7 (one per SIMD): trace and norm partials (src/icnf.jl:334, :343)
        f32x4 aN[8];                                       // A operands of the narrow products, fetched one interval ahead
        auto reverse_first = [&]() {
            f32x4 bN[8];
            s3_load<8>(bN, nrG1);
            S3_SB();
            f32x4 j0 = zero4, j1 = zero4;
            s3_mm<0, 8>(j0, j1, aN, bN);
            const f32x4 ej = j0 + j1;
            redw[s3::NB * 8] = -s3_dot4(ej, *(const f32x4*)epw);
            redw[2 * s3::NB * 8] = s3_dot4(ej, ej);
        };

        for (int stg = 1; stg <= nstg; ++stg) {
            f32x4 bA[8], bB[8];
            f32x4 p0, p1, r0_, r1_;
            // ---- interval 0: first layer (waves 0-3: tiles w and w+4, both halves) || eJ of the previous evaluation (4-7)
            if (zown) {
                s3_load<2>(bA, x0r, 0);
                s3_load<2>(bB, x0r + XB, 0);                                   // half B: used behind the barrier
                const f32x4 bv0 = *(const f32x4*)(bias + 16 * wave + 4 * q), bv1 = *(const f32x4*)(bias + 16 * wave + 64 + 4 * q);
                S3_SB();
                f32x4 c0 = zero4, c1 = zero4, d0 = zero4, d1 = zero4;
                s3_mm<0, 2>(c0, c1, wF1, bA, 0, 0);                            // half A, tile w
                s3_mm<0, 2>(d0, d1, wF1, bA, 2, 0);                            // half A, tile w+4
                *(f32x4*)h1w = s3_tanh4(c0 + c1 + bv0);
                *(f32x4*)(h1w + 64) = s3_tanh4(d0 + d1 + bv1);
            } else if (stg > 1) {
                reverse_first();
            }
            S3T(0);
            s3_bar();                                                          // h1(A) visible
            S3T(1);
            // ---- interval 1: half A of the second layer requested; under its flight waves 0-3 do half B of the first
            // layer (operands already in registers); then second layer, half A, k-blocks 0..5
            s3_load<8>(bA, h1r);
            S3_SB();
            if (zown) {
                const f32x4 bv0 = *(const f32x4*)(bias + 16 * wave + 4 * q), bv1 = *(const f32x4*)(bias + 16 * wave + 64 + 4 * q);
                p0 = zero4; p1 = zero4; r0_ = zero4; r1_ = zero4;
                s3_mm<0, 2>(p0, p1, wF1, bB, 0, 0);                            // half B, tile w
                s3_mm<0, 2>(r0_, r1_, wF1, bB, 2, 0);                          // half B, tile w+4
                *(f32x4*)(h1w + HB) = s3_tanh4(p0 + p1 + bv0);
                *(f32x4*)(h1w + HB + 64) = s3_tanh4(r0_ + r1_ + bv1);
            }
            f32x4 a0 = zero4, a1 = zero4;
            s3_mm<0, 6>(a0, a1, wF2, bA);
            // scalar rows of the PREVIOUS evaluation from its RED partials: complete since the barrier above,
            // rewritten from this evaluation's last forward epilogue on
            if (stg > 1 && sown) sc_set(stg, read_scalars());                  // slot j holds k_j
            S3T(2);
            s3_bar();                                                          // h1(B) visible
            S3T(3);
            // ---- interval 2: half B requested, half A finished under its flight, half B k-blocks 0..5 || epilogue A
            s3_mm_load<6, 6>(a0, a1, wF2, bA, bB, h1r + HB);
            s3_mm<7, 8>(a0, a1, wF2, bA);
            f32x4 e0 = zero4, e1 = zero4;
            s3_mm<0, 6>(e0, e1, wF2, bB);
            const f32x4 bv2 = *(const f32x4*)(bias + s3::PH + 16 * wave + 4 * q);
            *(f32x4*)h2w = s3_tanh4(a0 + a1 + bv2);
            S3T(4);
            s3_bar();                                                          // h2(A) visible
            S3T(5);
            // ---- interval 3: rest of half B and its epilogue (waves 0-3: the W3 rows of the next interval on their way)
            if (zown) s3_load<8>(aN, nrW);
            s3_mm<6, 8>(e0, e1, wF2, bB);
            *(f32x4*)(h2w + HB) = s3_tanh4(e0 + e1 + bv2);
            S3T(6);
            s3_bar();                                                          // h2 complete
            S3T(7);
            // ---- interval 4: last layer on waves 0-3 (one per SIMD): zdot rows r0..r0+3 of sample smp
            if (zown) {
                s3_load<8>(bA, nrH2);
                const f32x4 bv3 = *(const f32x4*)(bias + 2 * s3::PH + r0);
                S3_SB();
                f32x4 z0 = zero4, z1 = zero4;
                s3_mm<0, 8>(z0, z1, aN, bA);
#ifdef S3_STAMPS
                S3_SB(); S3T(29); S3_SB();
#endif
                // stage sum without the k this evaluation will produce: pre = u + h sum_{j<stg} a_{stg+1,j} k_j
                // (straight-line: the coefficients of k's not yet produced are zero in the table, their slots hold zeros)
                const float* A = tab.a[stg < 6 ? stg + 1 : 6];
                f32x4 pre = *(const f32x4*)rkw + (hstep * A[0]) * *(const f32x4*)(rkw + 32);
#pragma unroll
                for (int j = 1; j < 5; ++j) pre += (hstep * A[j]) * *(const f32x4*)(kzw + 32 * (j - 1));
                const f32x4 ev = *(const f32x4*)epw;
                const f32x4 zd = s3_tanh4(z0 + z1 + bv3);                      // padded rows: zero weights and bias -> 0
                *(f32x4*)g3w = ev * s3_dtanh4(zd);                             // g3 = eps .* sigma'_3
                if (stg < 6) *(f32x4*)x0w = pre + (hstep * A[stg]) * zd;       // state of the next evaluation
                *(f32x4*)(kzw + 32 * (stg - 1)) = zd;                          // k_{stg+1}
                redw[0] = s3_dot4(zd, zd);
            }
            S3T(8);
            s3_bar();                                                          // g3 (and the next stage state) visible
            S3T(9);
            // ---- interval 5: reverse of the last layer, both halves; g2(A) over h2(A)
            s3_load<2>(bA, g3r, 0);
            s3_load<2>(bB, g3r + XB, 0);                                       // half B: used behind the barrier
            const f32x4 hv2A = *(const f32x4*)h2w;
            S3_SB();
            f32x4 c0 = zero4, c1 = zero4, d0 = zero4, d1 = zero4;
            s3_mm<0, 2>(c0, c1, wB3, bA, 0, 0);
            *(f32x4*)h2w = (c0 + c1) * s3_dtanh4(hv2A);
            S3T(10);
            s3_bar();                                                          // g2(A) visible
            S3T(11);
            // ---- interval 6: half A of the reverse second layer requested; under its flight the reverse last layer of
            // half B; then reverse second layer, half A, k-blocks 0..5
            s3_mm_load<0, 0>(d0, d1, wB3, bB, bA, h2r);
            const f32x4 hv2B = *(const f32x4*)(h2w + HB);
            s3_mm<1, 2>(d0, d1, wB3, bB, 0, 0);
            *(f32x4*)(h2w + HB) = (d0 + d1) * s3_dtanh4(hv2B);
            a0 = zero4; a1 = zero4;
            s3_mm<0, 6>(a0, a1, wB2, bA);
            S3T(12);
            s3_bar();                                                          // g2(B) visible
            S3T(13);
            // ---- interval 7: half B requested, half A finished, half B k-blocks 0..5 || g1(A)
            s3_mm_load<6, 6>(a0, a1, wB2, bA, bB, h2r + HB);
            const f32x4 hv1A = *(const f32x4*)h1w;
            s3_mm<7, 8>(a0, a1, wB2, bA);
            e0 = zero4; e1 = zero4;
            s3_mm<0, 6>(e0, e1, wB2, bB);
            *(f32x4*)g1w = (a0 + a1) * s3_dtanh4(hv1A);
            S3T(14);
            s3_bar();                                                          // g1(A) visible
            S3T(15);
            // ---- interval 8: rest of half B, g1(B) (waves 4-7: the W1^T rows of their next product on their way)
            if (!zown) s3_load<8>(aN, nrW);
            const f32x4 hv1B = *(const f32x4*)(h1w + HB);
            s3_mm<6, 8>(e0, e1, wB2, bB);
            *(f32x4*)(g1w + HB) = (e0 + e1) * s3_dtanh4(hv1B);
            S3T(16);
            s3_bar();                                                          // g1 complete
            S3T(17);
        }
        if (!zown) reverse_first();                        // eJ of the last evaluation
        S3T(18);
        s3_bar();                                          // RED of the last evaluation complete
        S3T(19);
        if (single) {
            // ---- one evaluation: f -> out, and the norms of the initial-dt phase over the rows this lane owns ----
            float* out = (single == 1 ? a.du : a.Ks0) + (size_t)(tile * s3::NB + 16 * hf + s) * D;
            auto norms = [&](const f32x4& u4, const f32x4& f0, const f32x4& f1, int nvalid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nvalid) {
                        const float sk = fmaf(fabsf(u4[c]), reltol, abstol);
                        if (single == 1) {
                            const float x = u4[c] / sk, y = f1[c] / sk;
                            errsum = fmaf(x, x, errsum); badcnt = fmaf(y, y, badcnt);
                        } else {
                            const float x = (f1[c] - f0[c]) / sk;
                            errsum = fmaf(x, x, errsum);
                        }
                    }
                }
            };
            if (live && zown) {
                const f32x4 f1 = *(const f32x4*)kzw;                    // k2 slot = this evaluation's zdot
                st4(out + r0, f1, nv);
                if (a.init_phase >= 0) norms(*(const f32x4*)rkw, *(const f32x4*)(rkw + 32), f1, nv);
            }
            if (live && sown) {
                const f32x4 f1 = read_scalars();
                out[n_in] = f1.x; out[n_in + 1] = f1.y; out[n_in + 2] = f1.z;
                if (a.init_phase >= 0) norms(sc_get(0), sc_get(1), f1, 3);
            }
        }
        // ---- error estimate and outputs: u_new (= the state evaluation 6 ran at) and k7 ----
        if (!single && live && zown) {
            const f32x4 k7z = *(const f32x4*)(kzw + 32 * 5), uz_ = *(const f32x4*)rkw;
            const f32x4 un = *(const f32x4*)x0w;           // U_7 = u_new: the state the last evaluation ran at
            f32x4 ez = TS_BT1 * *(const f32x4*)(rkw + 32) + TS_BT7 * k7z;
            ez += TS_BT2 * *(const f32x4*)(kzw) + TS_BT3 * *(const f32x4*)(kzw + 32) + TS_BT4 * *(const f32x4*)(kzw + 64) +
                  TS_BT5 * *(const f32x4*)(kzw + 96) + TS_BT6 * *(const f32x4*)(kzw + 128);
#pragma unroll
            for (int c = 0; c < 4; ++c) {                  // rows beyond n_in: u = k = 0 -> contribute exactly 0
                const float scl = fmaf(fmaxf(fabsf(uz_[c]), fabsf(un[c])), reltol, abstol);
                const float x = c < nv ? hstep * ez[c] / scl : 0.f;
                errsum = fmaf(x, x, errsum);
                badcnt += (c < nv && !(fabsf(un[c]) <= 3.0e38f)) ? 1.f : 0.f;
            }
            const size_t gc = (size_t)(tile * s3::NB + 16 * hf + s) * D;
            float* Un = Uout + gc + r0;
            float* K7 = K1out + gc + r0;
            if (nv >= 4) { st4_wide(Un, un); st4_wide(K7, k7z); }
            else { st4(Un, un, nv); st4(K7, k7z, nv); }
        }
        if (!single && live && sown) {
            f32x4 ks[7];
#pragma unroll
            for (int j = 0; j < 6; ++j) ks[j] = sc_get(1 + j);
            const f32x4 us = sc_get(0);
            ks[6] = read_scalars();                        // k7 of the scalar rows, straight from the partials
            const f32x4 uns = us + hstep * stage_acc4<6>(ks);
            err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, 3);
            const size_t gc = (size_t)(tile * s3::NB + 16 * hf + s) * D;
            float* Un = Uout + gc + n_in;
            float* K7 = K1out + gc + n_in;
            Un[0] = uns.x; Un[1] = uns.y; Un[2] = uns.z;
            K7[0] = ks[6].x; K7[1] = ks[6].y; K7[2] = ks[6].z;
        }
        S3T(30);
        S3T(31);
        s3_bar();                                          // this tile's RED / SC / KZ reads precede the next tile's writes
    }
    // deterministic block reduction of the error partial (fixed tree, fixed order)
    errsum = s3_wave_sum(errsum);
    badcnt = s3_wave_sum(badcnt);
    if (lane == 0) { msc[wave] = errsum; msc[16 + wave] = badcnt; }
    s3_bar();
    if (tid == 0) {
        float e = 0.f, b = 0.f;
        for (int w = 0; w < 8; ++w) { e += msc[w]; b += msc[16 + w]; }
        if (!single) {
            a.partials[2 * blockIdx.x] = e;
            a.partials[2 * blockIdx.x + 1] = b;
        } else if (a.init_phase >= 0) {
            // initial-dt phase: partials through agent-scope atomics, then a ticket; whoever draws the last one sums all
            // partials (fixed order) and runs the controller phase -- no separate launches
            __hip_atomic_store(a.partials + 2 * blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.partials + 2 * blockIdx.x + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            msc[40] = (tk == gridDim.x - 1) ? 1.f : 0.f;
            if (tk == gridDim.x - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (single && a.init_phase >= 0) {
        s3_bar();
        if (msc[40] != 0.f) {                        // this workgroup drew the last ticket: all its threads reduce
            float q0 = 0.f, q1 = 0.f;
            for (int i = tid; i < (int)gridDim.x; i += 512) {
                q0 += __hip_atomic_load(a.partials + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q1 += __hip_atomic_load(a.partials + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            q0 = s3_wave_sum(q0); q1 = s3_wave_sum(q1);
            if (lane == 0) { msc[wave] = q0; msc[16 + wave] = q1; }
            s3_bar();
            if (tid == 0) {
                float p0 = 0.f, p1 = 0.f;
                for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
                ctrl_phase(a.st_out, single - 1, p0, p1, a.n_total);
            }
        }
    }
#ifdef S3_STAMPS
    S3T(21);
    const unsigned long long s3rt1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 1 || wave == 4 || wave == 5))
        printf("wave %d total %llu pre %llu tail %llu | F1 %llu+%llu F2A %llu+%llu F2B %llu+%llu F2t %llu+%llu F3 %llu+%llu | "
               "B3 %llu+%llu B2A %llu+%llu B2B %llu+%llu B2t %llu+%llu B1 %llu fin %llu\n",
               wave, s3last - s3start, s3acc[20], s3acc[21], s3acc[0], s3acc[1], s3acc[2], s3acc[3], s3acc[4], s3acc[5], s3acc[6],
               s3acc[7], s3acc[8], s3acc[9], s3acc[10], s3acc[11], s3acc[12], s3acc[13], s3acc[14], s3acc[15], s3acc[16],
               s3acc[17], s3acc[18], s3acc[19]);
    if (blockIdx.x == 7 && lane == 0 && wave == 0) {
        const unsigned long long rt = s3rt1 - s3rt0;
        printf("  clock: %llu shader cycles in %llu x 10 ns -> %.0f MHz\n", s3last - s3start, rt, (double)(s3last - s3start) / (double)rt * 100.0);
    }
    if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 4))
        printf("  wave %d prologue: scalars %llu issue %llu partials %llu staged(in ctrl bar) | to ctrl bar %llu wait %llu ctrl %llu wait %llu tileprep %llu wait(in pre) | F3 loads+mfma %llu (rest in F3) | tail: err %llu stores %llu fin %llu\n",
               wave, s3acc[32], s3acc[33], s3acc[34], s3acc[24], s3acc[25], s3acc[26], s3acc[27], s3acc[28], s3acc[29], s3acc[30], s3acc[31], s3acc[21]);
#endif
}

// k_step3j -- the same step for the JVP compute mode (DIJacVecMatrixMode, src/icnf.jl:384-420): per evaluation ONE
// forward sweep of the state columns h_l and the tangent columns tau_l = sigma'_l .* (W_l tau_{l-1}), tau_0 = eps, as two
// column tiles per sample half that share every weight fragment; ldot = -eps.(J eps), Edot = |zdot|, ndot = |J eps|.
// Three dependent products instead of six: 3 barrier intervals per evaluation, no reverse fragments, no W2 staging.
// Prologue, controller, Runge-Kutta bookkeeping and outputs are those of k_step3.
__global__ void __launch_bounds__(512, 2) k_step3j(MfmaArgs a, const float* __restrict__ img3, int n_in, int norm_z,
                                                  int norm_j, const S3Tab tab, int single) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    S3_ARGS_UP_FRONT(a, img3);
    const StepState* st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const int t = wave & 1, hf = (wave >> 1) & 1;         // narrow phases: row tile and sample half of this wave
    const bool zown = wave < 4;                           // waves 0-3 produce zdot: they hold the z rows of the state
    const bool sown = !zown && t == 0 && q == 0;          // waves 4, 6: lane s holds the scalar rows of sample 16 hf + s
    const int smp = 16 * hf + s;                          // sample of this lane in the narrow phases
    const int r0 = 16 * t + 4 * q;                        // first of its 4 rows there
    const int nv = n_in - r0;                             // valid rows among them (may be <= 0 or > 4)
#ifdef S3_STAMPS
    unsigned long long s3acc[36] = {0};
    unsigned long long s3last = __builtin_amdgcn_s_memtime();
    const unsigned long long s3start = s3last;
    const unsigned long long s3rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- everything the launch needs from memory is requested up front, in ONE round trip, and nothing is consumed
    // before all of it is in flight.  Order of issue = order of return: the integrator state words first (the `done`
    // test and the controller need them soonest), then the error partials, the weight stream (135 KB per workgroup:
    // it bounds the prologue, so it must not queue behind anything that waits), then this workgroup's first tile from
    // BOTH buffer sets (which one is current is the controller's decision).
    const StepState* stw = s3_state_words(st);
    const int v_done = stw->done, v_cur = stw->cur;
    const float v_h = stw->h, v_abstol = stw->abstol, v_reltol = stw->reltol;
    // (one unconditional load per thread: a loop here would wait for its data before anything below is even requested)
    const float* ppin = a.apply_ctrl ? a.partials_in : img3;
    const float2 pp = *reinterpret_cast<const float2*>(ppin + 2 * min(tid, (int)gridDim.x - 1));
    // Weights: forward orientation only.  W3 rows (the K = 128 operand of the last layer) into their LDS image by
    // LDS-DMA; this wave's 16-row tiles of W1 (2 fragments) and of W2 (8 fragments, straight from the row-major image).
    constexpr int NCN = s3::P0 * s3::SXH / 4, NCB = (2 * s3::PH + s3::P0) / 4;
    static_assert(NCN % 64 == 0, "whole wave instructions");
    {
        typedef __attribute__((address_space(3))) float* lds_f;
        typedef const __attribute__((address_space(1))) float* glb_f;
#pragma unroll
        for (int i = 0; i < (NCN + 511) / 512; ++i) {
            const int c = 512 * i + 64 * wave;
            if (c < NCN)
                __builtin_amdgcn_global_load_lds((glb_f)(img3 + s3::IMG_W3R + 4 * (c + lane)), (lds_f)(lds + s3::W3R + 4 * c), 16, 0, 0);
        }
    }
    const f32x4 sgb = reinterpret_cast<const f32x4*>(img3 + s3::IMG_B1)[min(tid, NCB - 1)];
    f32x4 wF1[2], wF2[8];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(img3 + s3::IMG_FR1) + (size_t)((wave & 3) * 4 + 2 * (wave >> 2)) * 64 + lane;
        wF1[0] = wp[0]; wF1[1] = wp[64];
        const float* rw = img3 + s3::IMG_W2 + (16 * wave + s) * s3::SW2 + 4 * q;
#pragma unroll
        for (int u = 0; u < 8; ++u) wF2[u] = *(const f32x4*)(rw + 16 * u);
    }
    const int ntile = (a.B + s3::NB - 1) / s3::NB;
    const bool wide = (n_in & 3) == 0;                     // every lane's four rows are all valid or all padding
    f32x4 ru[2], rk[2], re, rs[2][2];
    int ce = 0, cu = 0, cs = 0;
    float* sc = lds + s3::SC + smp * 24;
    auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
    auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
    {
        const int b0 = blockIdx.x * s3::NB + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
        re = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ru[c] = ld4_issue_w(a.U[c] + gcol + r0, cu, img3, wide);
            rk[c] = ld4_issue_w(a.K1[c] + gcol + r0, cu, img3, wide);
            rs[c][0] = ld3_issue(a.U[c] + gcol + n_in, cs, img3);
            rs[c][1] = ld3_issue(a.K1[c] + gcol + n_in, cs, img3);
        }
    }
    __builtin_amdgcn_sched_barrier(0);                     // nothing above is consumed before all of it is requested
    // (wave-uniform values into scalar registers: the buffer pointers selected from `cur` stay out of the vector file)
    const int st_done = __builtin_amdgcn_readfirstlane(v_done), st_cur = __builtin_amdgcn_readfirstlane(v_cur);
    const float st_h = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_h)));
    const float st_abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_abstol)));
    const float st_reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_reltol)));
    S3T(32);
    if (st_done) {       // launches queued past the end of the solve: keep the state chain intact and leave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA pieces must have landed before the wave ends)
        if (a.apply_ctrl && blockIdx.x == 0 && tid == 0) { *a.st_out = *st; publish_mirror(a, *st); }
        return;
    }
    // scalar rows of both candidates wait in LDS (slots 2..5 of the scalar-row state) for the controller's choice
    if (sown) {
        sc_set(2, ld4_mask(rs[0][0], cs)); sc_set(3, ld4_mask(rs[0][1], cs));
        sc_set(4, ld4_mask(rs[1][0], cs)); sc_set(5, ld4_mask(rs[1][1], cs));
    }
    // the probe rows go straight to their LDS image (waves 0-3 need them for g3, waves 4-7 for the trace row)
    *(f32x4*)(lds + s3::EPS + smp * s3::SX0 + r0) = ld4_mask(re, ce);
    float* msc = lds + s3::MISC;
    S3T(33);
    if (a.apply_ctrl) {
        float cp0 = tid < (int)gridDim.x ? pp.x : 0.f, cp1 = tid < (int)gridDim.x ? pp.y : 0.f;
        for (int i = tid + 512; i < (int)gridDim.x; i += 512) { cp0 += a.partials_in[2 * i]; cp1 += a.partials_in[2 * i + 1]; }
        cp0 = s3_wave_sum(cp0); cp1 = s3_wave_sum(cp1);
        if (lane == 0) { msc[wave] = cp0; msc[16 + wave] = cp1; }
    }
    S3T(34);
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + s3::BIAS)[tid] = sgb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's LDS-DMA pieces have landed (then the barrier)
    S3T(24);
    s3_bar();                                              // staging image and partial sums complete
    S3T(25);
    int cur = st_cur;
    float hstep = st_h, abstol = st_abstol, reltol = st_reltol;
    if (a.apply_ctrl && tid == 0) {
        // In-kernel step controller (as in k_mfma): every workgroup reduces the same partials in the same order
        // and takes the same decision; block 0 publishes the new state for the next launch and the host mirror.
        float p0 = 0.f, p1 = 0.f;
        for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
        StepState ns = *st;                                // (cached by now: the state words above came from these lines)
        ctrl_after_step(&ns, p0, p1, a.n_total);
        if (blockIdx.x == 0) { *a.st_out = ns; publish_mirror(a, ns); }
        msc[32] = __int_as_float(ns.cur); msc[33] = ns.h; msc[34] = ns.abstol; msc[35] = ns.reltol;
        msc[36] = __int_as_float(ns.done);
    }
    S3T(26);
    s3_bar();                                              // controller done; the staging area is free
    S3T(27);
    if (a.apply_ctrl) {
        cur = __builtin_amdgcn_readfirstlane(__float_as_int(msc[32]));
        hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[33])));
        abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[34])));
        reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[35])));
        if (__float_as_int(msc[36])) return;        // the controller just finished the solve
    }
    // (explicit selects: indexing the kernel-argument arrays with a run-time value makes the compiler fetch the
    // pointer from the argument segment through a vector load -- a memory round trip in front of the state loads)
    // single != 0: ONE evaluation instead of a step attempt -- the two launches of the automatic initial dt
    // (1: f(u) -> a.du with the norms of phase 0; 2: f(u + h k1) -> a.Ks0 with the norm of phase 1), same prologue,
    // same evaluation code; the last workgroup to finish runs the controller phase (as k_mfma does for them)
    const int nstg = single ? 1 : 6;
    const float c21 = single == 1 ? 0.f : (single == 2 ? 1.f : TS_A21);
    const float* Uin = cur ? a.U[1] : a.U[0];
    const float* K1in = cur ? a.K1[1] : a.K1[0];
    float* Uout = cur ? a.U[0] : a.U[1];
    float* K1out = cur ? a.K1[0] : a.K1[1];

    float errsum = 0.f, badcnt = 0.f;
    // the 8 partials (2 row tiles x 4 lanes) of one sample and one kind sit side by side: two b128 reads each
    auto red8 = [&](int kind) {
        const float* r = lds + s3::RED + (kind * s3::NB + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    auto read_scalars = [&]() {
        const float e2 = red8(0), ld = red8(1), n2 = red8(2);
        return f32x4{ld, norm_z ? __builtin_sqrtf(e2) : 0.f, norm_j ? __builtin_sqrtf(n2) : 0.f, 0.f};
    };
    float* redw = lds + s3::RED + smp * 8 + 4 * t + q;                 // this lane's slot of kind 0 (+ 256 per kind)
    // LDS images of the tangent columns: T1 where k_step3 keeps g1, T2 where it keeps W1^T (not loaded here), sigma'_3 in G3
    constexpr int T1 = s3::G1, T2 = s3::W1T;
    const float* x0r = lds + s3::X0 + s * s3::SX0 + 4 * q;            // lane = sample s of half A; half B = +16 rows
    const float* e0r = lds + s3::EPS + s * s3::SX0 + 4 * q;           // tau_0 = eps
    const float* h1r = lds + s3::H1 + s * s3::SXH + 4 * q;
    const float* t1r = lds + T1 + s * s3::SXH + 4 * q;
    float* h1w = lds + s3::H1 + s * s3::SXH + 16 * wave + 4 * q;
    float* t1w = lds + T1 + s * s3::SXH + 16 * wave + 4 * q;
    float* h2w = lds + s3::H2 + s * s3::SXH + 16 * wave + 4 * q;
    float* t2w = lds + T2 + s * s3::SXH + 16 * wave + 4 * q;
    constexpr int HB = 16 * s3::SXH, XB = 16 * s3::SX0;              // half B = 16 samples on
    // last layer: wave -> (row tile t, sample half hf, kind): waves 0-3 the state columns (zdot), 4-7 the tangent columns
    const float* nr3 = lds + (zown ? s3::H2 : T2) + smp * s3::SXH + 4 * q;
    const float* nrW = lds + s3::W3R + (16 * t + s) * s3::SXH + 4 * q;
    float* x0w = lds + s3::X0 + smp * s3::SX0 + r0;
    float* g3w = lds + s3::G3 + smp * s3::SX0 + r0;
    // Runge-Kutta state of the z rows r0..r0+3 of sample smp, written and read by this lane only:
    float* rkw = lds + s3::KZ + smp * s3::SKZ + r0;                   // u at rkw, k1 at rkw + 32,
    float* kzw = rkw + 64;                                            // k_{j+2} at kzw + 32 j (j = 0..5)
    float* epw = lds + s3::EPS + smp * s3::SX0 + r0;                  // the probe rows eps
    const float* bias = lds + s3::BIAS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b0 = tile * s3::NB + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        f32x4 uz, k1z;
        if (tile == blockIdx.x) {                              // requested at kernel entry
            uz = ld4_mask(cur ? ru[1] : ru[0], cu);
            k1z = ld4_mask(cur ? rk[1] : rk[0], cu);
            if (sown) { sc_set(0, sc_get(cur ? 4 : 2)); sc_set(1, sc_get(cur ? 5 : 3)); }
            if (single == 1) { k1z = f32x4{0.f, 0.f, 0.f, 0.f}; if (sown) sc_set(1, k1z); }     // there is no k1 yet
        } else {
            ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
            const f32x4 e_ = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
            const f32x4 u_ = ld4_issue_w(Uin + gcol + r0, cu, img3, wide), k_ = ld4_issue_w(K1in + gcol + r0, cu, img3, wide);
            const f32x4 s0 = ld3_issue(Uin + gcol + n_in, cs, img3), s1 = ld3_issue(K1in + gcol + n_in, cs, img3);
            *(f32x4*)epw = ld4_mask(e_, ce); uz = ld4_mask(u_, cu); k1z = ld4_mask(k_, cu);
            if (sown) { sc_set(0, ld4_mask(s0, cs)); sc_set(1, ld4_mask(s1, cs)); }
            if (single == 1) { k1z = f32x4{0.f, 0.f, 0.f, 0.f}; if (sown) sc_set(1, k1z); }
        }
        if (zown) {
            *(f32x4*)x0w = uz + (hstep * c21) * k1z;      // state of evaluation 1: U_2 = u + h a21 k1
            *(f32x4*)rkw = uz;
            *(f32x4*)(rkw + 32) = k1z;
#pragma unroll
            for (int j = 0; j < 6; ++j) *(f32x4*)(kzw + 32 * j) = zero4;       // k2..k7: not produced yet
        }
        S3T(28);
        s3_bar();
        S3T(20);

        // tau_3 = sigma'_3 .* (W3 tau_2) on waves 4-7: trace and norm partials (src/icnf.jl:404, :413).  The product is
        // formed in interval 2; sigma'_3 comes from the wave that formed zdot of the same rows, one barrier later.
        f32x4 aN[8], tacc = zero4;
        auto finish_tau = [&]() {
            const f32x4 tj = tacc * *(const f32x4*)g3w;
            redw[s3::NB * 8] = -s3_dot4(tj, *(const f32x4*)epw);
            redw[2 * s3::NB * 8] = s3_dot4(tj, tj);
        };

        for (int stg = 1; stg <= nstg; ++stg) {
            // ---- interval 0: first layer, tile `wave`, both halves, state and tangent columns (K = 32)
            {
                f32x4 bh[4], bt[4];
                s3_load<2>(bh, x0r, 0); s3_load<2>(bh, x0r + XB, 2);
                s3_load<2>(bt, e0r, 0); s3_load<2>(bt, e0r + XB, 2);
                const f32x4 bv1 = *(const f32x4*)(bias + 16 * wave + 4 * q);
                if (!zown && stg > 1) finish_tau();                            // of the previous evaluation
                S3_SB();
                f32x4 c0 = zero4, c1 = zero4, d0 = zero4, d1 = zero4, e0 = zero4, e1 = zero4, f0 = zero4, f1 = zero4;
                s3_mm<0, 2>(c0, c1, wF1, bh, 0, 0);
                s3_mm<0, 2>(d0, d1, wF1, bh, 0, 2);
                s3_mm<0, 2>(e0, e1, wF1, bt, 0, 0);
                s3_mm<0, 2>(f0, f1, wF1, bt, 0, 2);
                const f32x4 hA = s3_tanh4(c0 + c1 + bv1), hB = s3_tanh4(d0 + d1 + bv1);
                *(f32x4*)h1w = hA;
                *(f32x4*)t1w = s3_dtanh4(hA) * (e0 + e1);
                *(f32x4*)(h1w + HB) = hB;
                *(f32x4*)(t1w + HB) = s3_dtanh4(hB) * (f0 + f1);
            }
            S3T(0);
            s3_bar();                                                          // h1, t1 visible
            S3T(1);
            // ---- interval 1: second layer, tile `wave`, both halves, state and tangent columns: operands one k-block
            // ahead of the MFMAs (ring of 2), 16 MFMAs per k-block on 8 chains
            {
                f32x4 rb[2][4];
                rb[0][0] = *(const f32x4*)h1r; rb[0][1] = *(const f32x4*)t1r;
                rb[0][2] = *(const f32x4*)(h1r + HB); rb[0][3] = *(const f32x4*)(t1r + HB);
                f32x4 acc[4][2];
#pragma unroll
                for (int n = 0; n < 4; ++n) { acc[n][0] = zero4; acc[n][1] = zero4; }
                const f32x4 bv2 = *(const f32x4*)(bias + s3::PH + 16 * wave + 4 * q);
                // scalar rows of the PREVIOUS evaluation from its RED partials (complete since the barrier above)
                if (stg > 1 && sown) sc_set(stg, read_scalars());              // slot j holds k_j
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (u + 1 < 8) {
                        rb[(u + 1) & 1][0] = *(const f32x4*)(h1r + 16 * (u + 1));
                        rb[(u + 1) & 1][1] = *(const f32x4*)(t1r + 16 * (u + 1));
                        rb[(u + 1) & 1][2] = *(const f32x4*)(h1r + HB + 16 * (u + 1));
                        rb[(u + 1) & 1][3] = *(const f32x4*)(t1r + HB + 16 * (u + 1));
                    } else {
                        s3_load<8>(aN, nrW);                                   // W3 rows of the next interval on their way
                    }
                    S3_SB();
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[n][c & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wF2[u][c], rb[u & 1][n][c], acc[n][c & 1], 0, 0, 0);
                    }
                    S3_SB();
                }
                const f32x4 hA = s3_tanh4(acc[0][0] + acc[0][1] + bv2), hB = s3_tanh4(acc[2][0] + acc[2][1] + bv2);
                *(f32x4*)h2w = hA;
                *(f32x4*)t2w = s3_dtanh4(hA) * (acc[1][0] + acc[1][1]);
                *(f32x4*)(h2w + HB) = hB;
                *(f32x4*)(t2w + HB) = s3_dtanh4(hB) * (acc[3][0] + acc[3][1]);
            }
            S3T(2);
            s3_bar();                                                          // h2, t2 visible
            S3T(3);
            // ---- interval 2: last layer, one product per wave: rows r0..r0+3 of sample smp, state (0-3) / tangent (4-7)
            {
                f32x4 bA[8];
                s3_load<8>(bA, nr3);
                const f32x4 bv3 = *(const f32x4*)(bias + 2 * s3::PH + r0);
                S3_SB();
                f32x4 z0 = zero4, z1 = zero4;
                s3_mm<0, 8>(z0, z1, aN, bA);
                if (zown) {
                    // stage sum without the k this evaluation will produce: pre = u + h sum_{j<stg} a_{stg+1,j} k_j
                    const float* A = tab.a[stg < 6 ? stg + 1 : 6];
                    f32x4 pre = *(const f32x4*)rkw + (hstep * A[0]) * *(const f32x4*)(rkw + 32);
#pragma unroll
                    for (int jj = 1; jj < 5; ++jj) pre += (hstep * A[jj]) * *(const f32x4*)(kzw + 32 * (jj - 1));
                    const f32x4 zd = s3_tanh4(z0 + z1 + bv3);                  // padded rows: zero weights and bias -> 0
                    *(f32x4*)g3w = s3_dtanh4(zd);                              // sigma'_3 for the tangent rows
                    if (stg < 6) *(f32x4*)x0w = pre + (hstep * A[stg]) * zd;   // state of the next evaluation
                    *(f32x4*)(kzw + 32 * (stg - 1)) = zd;                      // k_{stg+1}
                    redw[0] = s3_dot4(zd, zd);
                } else {
                    tacc = z0 + z1;
                }
            }
            S3T(4);
            s3_bar();                                                          // sigma'_3 (and the next stage state) visible
            S3T(5);
        }
        if (!zown) finish_tau();                           // of the last evaluation
        S3T(18);
        s3_bar();                                          // RED of the last evaluation complete
        S3T(19);
        if (single) {
            // ---- one evaluation: f -> out, and the norms of the initial-dt phase over the rows this lane owns ----
            float* out = (single == 1 ? a.du : a.Ks0) + (size_t)(tile * s3::NB + 16 * hf + s) * D;
            auto norms = [&](const f32x4& u4, const f32x4& f0, const f32x4& f1, int nvalid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nvalid) {
                        const float sk = fmaf(fabsf(u4[c]), reltol, abstol);
                        if (single == 1) {
                            const float x = u4[c] / sk, y = f1[c] / sk;
                            errsum = fmaf(x, x, errsum); badcnt = fmaf(y, y, badcnt);
                        } else {
                            const float x = (f1[c] - f0[c]) / sk;
                            errsum = fmaf(x, x, errsum);
                        }
                    }
                }
            };
            if (live && zown) {
                const f32x4 f1 = *(const f32x4*)kzw;                    // k2 slot = this evaluation's zdot
                st4(out + r0, f1, nv);
                if (a.init_phase >= 0) norms(*(const f32x4*)rkw, *(const f32x4*)(rkw + 32), f1, nv);
            }
            if (live && sown) {
                const f32x4 f1 = read_scalars();
                out[n_in] = f1.x; out[n_in + 1] = f1.y; out[n_in + 2] = f1.z;
                if (a.init_phase >= 0) norms(sc_get(0), sc_get(1), f1, 3);
            }
        }
        // ---- error estimate and outputs: u_new (= the state evaluation 6 ran at) and k7 ----
        if (!single && live && zown) {
            const f32x4 k7z = *(const f32x4*)(kzw + 32 * 5), uz_ = *(const f32x4*)rkw;
            const f32x4 un = *(const f32x4*)x0w;           // U_7 = u_new: the state the last evaluation ran at
            f32x4 ez = TS_BT1 * *(const f32x4*)(rkw + 32) + TS_BT7 * k7z;
            ez += TS_BT2 * *(const f32x4*)(kzw) + TS_BT3 * *(const f32x4*)(kzw + 32) + TS_BT4 * *(const f32x4*)(kzw + 64) +
                  TS_BT5 * *(const f32x4*)(kzw + 96) + TS_BT6 * *(const f32x4*)(kzw + 128);
#pragma unroll
            for (int c = 0; c < 4; ++c) {                  // rows beyond n_in: u = k = 0 -> contribute exactly 0
                const float scl = fmaf(fmaxf(fabsf(uz_[c]), fabsf(un[c])), reltol, abstol);
                const float x = c < nv ? hstep * ez[c] / scl : 0.f;
                errsum = fmaf(x, x, errsum);
                badcnt += (c < nv && !(fabsf(un[c]) <= 3.0e38f)) ? 1.f : 0.f;
            }
            const size_t gc = (size_t)(tile * s3::NB + 16 * hf + s) * D;
            float* Un = Uout + gc + r0;
            float* K7 = K1out + gc + r0;
            if (nv >= 4) { st4_wide(Un, un); st4_wide(K7, k7z); }
            else { st4(Un, un, nv); st4(K7, k7z, nv); }
        }
        if (!single && live && sown) {
            f32x4 ks[7];
#pragma unroll
            for (int j = 0; j < 6; ++j) ks[j] = sc_get(1 + j);
            const f32x4 us = sc_get(0);
            ks[6] = read_scalars();                        // k7 of the scalar rows, straight from the partials
            const f32x4 uns = us + hstep * stage_acc4<6>(ks);
            err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, 3);
            const size_t gc = (size_t)(tile * s3::NB + 16 * hf + s) * D;
            float* Un = Uout + gc + n_in;
            float* K7 = K1out + gc + n_in;
            Un[0] = uns.x; Un[1] = uns.y; Un[2] = uns.z;
            K7[0] = ks[6].x; K7[1] = ks[6].y; K7[2] = ks[6].z;
        }
        S3T(30);
        S3T(31);
        s3_bar();                                          // this tile's RED / SC / KZ reads precede the next tile's writes
    }
    // deterministic block reduction of the error partial (fixed tree, fixed order)
    errsum = s3_wave_sum(errsum);
    badcnt = s3_wave_sum(badcnt);
    if (lane == 0) { msc[wave] = errsum; msc[16 + wave] = badcnt; }
    s3_bar();
    if (tid == 0) {
        float e = 0.f, b = 0.f;
        for (int w = 0; w < 8; ++w) { e += msc[w]; b += msc[16 + w]; }
        if (!single) {
            a.partials[2 * blockIdx.x] = e;
            a.partials[2 * blockIdx.x + 1] = b;
        } else if (a.init_phase >= 0) {
            // initial-dt phase: partials through agent-scope atomics, then a ticket; whoever draws the last one sums all
            // partials (fixed order) and runs the controller phase -- no separate launches
            __hip_atomic_store(a.partials + 2 * blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.partials + 2 * blockIdx.x + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            msc[40] = (tk == gridDim.x - 1) ? 1.f : 0.f;
            if (tk == gridDim.x - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (single && a.init_phase >= 0) {
        s3_bar();
        if (msc[40] != 0.f) {                        // this workgroup drew the last ticket: all its threads reduce
            float q0 = 0.f, q1 = 0.f;
            for (int i = tid; i < (int)gridDim.x; i += 512) {
                q0 += __hip_atomic_load(a.partials + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q1 += __hip_atomic_load(a.partials + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            q0 = s3_wave_sum(q0); q1 = s3_wave_sum(q1);
            if (lane == 0) { msc[wave] = q0; msc[16 + wave] = q1; }
            s3_bar();
            if (tid == 0) {
                float p0 = 0.f, p1 = 0.f;
                for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
                ctrl_phase(a.st_out, single - 1, p0, p1, a.n_total);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// k_step3jb -- k_step3j with every fp32 product formed from SIX bf16 MFMA terms: each operand is split into three bf16
// pieces (8 + 8 + 8 mantissa bits: the split is exact), v_mfma_f32_16x16x32_bf16 accumulates the six products above
// 2^-24 of the result in fp32.  Measured on a K = 128 tile (tools/ubench/bf16_split.hip): the same error as
// v_mfma_f32_16x16x4_f32 (6.7e-8 against 7.6e-8 of sum|a b|) at 2.3x its rate.  Weight fragments arrive pre-split
// (k_pack_step3b); activations are split in the epilogues and live in LDS as three bf16 images [sample][feature].
// The Runge-Kutta rows of z live in registers here (the images take their LDS).
// ---------------------------------------------------------------------------------------------------------------
// (bf16x8 / bf16x4 and the three-piece split s3b_split: cnf_split.h)
namespace s3b {
// fp32 regions (float offsets), then the bf16 images (byte offsets)
constexpr int G3 = 0, EPS = G3 + 32 * 40, RED = EPS + 32 * 40, SC = RED + 3 * 32 * 8, BIAS = SC + 32 * 24;
constexpr int MISC = BIAS + 2 * 128 + 32, FP_END = MISC + 64;
// images [piece hi | mid | lo][32 rows][K bf16]: rows without padding, 16-byte chunks XOR-swizzled by the row (as s3v below)
constexpr int WS = 256, WP = 32 * WS, WI = 3 * WP;        // K = 128
constexpr int NS = 64, NP = 32 * NS, NI = 3 * NP;         // K = 32
constexpr int H1B = FP_END * 4, T1B = H1B + WI, H2B = T1B + WI, T2B = H2B + WI, X0B = T2B + WI, T0B = X0B + NI;
constexpr int TOTAL_BYTES = T0B + NI;
static_assert(H1B % 16 == 0 && TOTAL_BYTES <= 160 * 1024, "LDS plan");
}  // namespace s3b
__global__ void __launch_bounds__(512, 2) k_step3jb(MfmaArgs a, const char* __restrict__ imgb, int n_in, int norm_z,
                                                   int norm_j, const S3Tab tab, int single) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const float* img3 = reinterpret_cast<const float*>(imgb);      // (a valid address for masked loads)
    S3_ARGS_UP_FRONT(a, imgb);
    const StepState* st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const int t = wave & 1, hf = (wave >> 1) & 1;         // last layer: row tile and sample half of this wave
    const bool zown = wave < 4;                           // waves 0-3 produce zdot: they hold the z rows of the state
    const bool sown = !zown && t == 0 && q == 0;          // waves 4, 6: lane s holds the scalar rows of sample 16 hf + s
    const int smp = 16 * hf + s;
    const int r0 = 16 * t + 4 * q;
    const int nv = n_in - r0;
    // ---- requests of the prologue (order of issue = order of return; nothing consumed before all are in flight) ----
    const StepState* stw = s3_state_words(st);
    const int v_done = stw->done, v_cur = stw->cur;
    const float v_h = stw->h, v_abstol = stw->abstol, v_reltol = stw->reltol;
    // (one unconditional load per thread: a loop here would wait for its data before anything below is even requested)
    const float* ppin = a.apply_ctrl ? a.partials_in : img3;
    const float2 pp = *reinterpret_cast<const float2*>(ppin + 2 * min(tid, (int)gridDim.x - 1));
    // resident split fragments: W1 tile `wave` (K = 32), W2 tile `wave` (4 k-blocks), W3 tile t (4 k-blocks)
    // (they arrive in fp32 and are split here, in arrival order, while the rest of the stream is in flight)
    S3bOp wF1, wF2[4], w3[4];
    {
        const char* fw = imgb + s3g::F32 + (size_t)wave * 10 * 2048 + 16 * lane;
        const char* f3 = imgb + s3g::F32 + (size_t)(80 + 4 * t) * 2048 + 16 * lane;
        auto src = [&](int f) { return f < 5 ? fw + f * 2048 : f3 + (f - 5) * 2048; };
        constexpr int AH = 5;                                  // fragments requested ahead of the one being split
        f32x4 raw[9][2];
#pragma unroll
        for (int f = 0; f < AH; ++f) { raw[f][0] = *(const f32x4*)src(f); raw[f][1] = *(const f32x4*)(src(f) + 1024); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            if (f + AH < 9) { raw[f + AH][0] = *(const f32x4*)src(f + AH); raw[f + AH][1] = *(const f32x4*)(src(f + AH) + 1024); }
            S3bOp o = s3b_split8(raw[f][0], raw[f][1]);
            s3b_pin(o);                                        // (the split stays here, between the two scheduling barriers)
            if (f == 0) wF1 = o; else if (f < 5) wF2[f - 1] = o; else w3[f - 5] = o;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    constexpr int NCB = (2 * 128 + 32) / 4;
    const f32x4 sgb = reinterpret_cast<const f32x4*>(imgb + s3g::BIASB)[min(tid, NCB - 1)];
    const int ntile = (a.B + 31) / 32;
    const bool wide = (n_in & 3) == 0;                     // every lane's four rows are all valid or all padding
    f32x4 ru[2], rk[2], re, rs[2][2];
    int ce = 0, cu = 0, cs = 0;
    float* sc = lds + s3b::SC + smp * 24;
    auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
    auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
    {
        const int b0 = blockIdx.x * 32 + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
        re = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ru[c] = ld4_issue_w(a.U[c] + gcol + r0, cu, img3, wide);
            rk[c] = ld4_issue_w(a.K1[c] + gcol + r0, cu, img3, wide);
            rs[c][0] = ld3_issue(a.U[c] + gcol + n_in, cs, img3);
            rs[c][1] = ld3_issue(a.K1[c] + gcol + n_in, cs, img3);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int st_done = __builtin_amdgcn_readfirstlane(v_done), st_cur = __builtin_amdgcn_readfirstlane(v_cur);
    const float st_h = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_h)));
    const float st_abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_abstol)));
    const float st_reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_reltol)));
    if (st_done) {       // launches queued past the end of the solve: keep the state chain intact and leave
        if (a.apply_ctrl && blockIdx.x == 0 && tid == 0) { *a.st_out = *st; publish_mirror(a, *st); }
        return;
    }
    if (sown) {
        sc_set(2, ld4_mask(rs[0][0], cs)); sc_set(3, ld4_mask(rs[0][1], cs));
        sc_set(4, ld4_mask(rs[1][0], cs)); sc_set(5, ld4_mask(rs[1][1], cs));
    }
    // the probe rows: fp32 for the trace row, split as tau_0 (the tangent operand of the first layer)
    float* epw = lds + s3b::EPS + smp * 40 + r0;
    const int nsw = (-(s >> 2)) & 3;                                 // chunk swizzle of the K = 32 images (rows s, 16 + s)
    const int nw = smp * s3b::NS + 16 * ((2 * t + (q >> 1)) ^ nsw) + 8 * (q & 1);      // this lane's 4 rows there
    char* x0w = ldsb + s3b::X0B + nw;
    {
        const f32x4 ev = ld4_mask(re, ce);
        *(f32x4*)epw = ev;
        s3b_store4(ldsb + s3b::T0B + nw, s3b::NP, ev);
    }
    float* msc = lds + s3b::MISC;
    if (a.apply_ctrl) {
        float cp0 = tid < (int)gridDim.x ? pp.x : 0.f, cp1 = tid < (int)gridDim.x ? pp.y : 0.f;
        for (int i = tid + 512; i < (int)gridDim.x; i += 512) { cp0 += a.partials_in[2 * i]; cp1 += a.partials_in[2 * i + 1]; }
        cp0 = s3_wave_sum(cp0); cp1 = s3_wave_sum(cp1);
        if (lane == 0) { msc[wave] = cp0; msc[16 + wave] = cp1; }
    }
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + s3b::BIAS)[tid] = sgb;
    s3_bar();                                              // partial sums, biases and probe images complete
    int cur = st_cur;
    float hstep = st_h, abstol = st_abstol, reltol = st_reltol;
    if (a.apply_ctrl && tid == 0) {
        float p0 = 0.f, p1 = 0.f;
        for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
        StepState ns = *st;                                // (cached by now: the state words above came from these lines)
        ctrl_after_step(&ns, p0, p1, a.n_total);
        if (blockIdx.x == 0) { *a.st_out = ns; publish_mirror(a, ns); }
        msc[32] = __int_as_float(ns.cur); msc[33] = ns.h; msc[34] = ns.abstol; msc[35] = ns.reltol;
        msc[36] = __int_as_float(ns.done);
    }
    s3_bar();                                              // controller done
    if (a.apply_ctrl) {
        cur = __builtin_amdgcn_readfirstlane(__float_as_int(msc[32]));
        hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[33])));
        abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[34])));
        reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[35])));
        if (__float_as_int(msc[36])) return;        // the controller just finished the solve
    }
    const int nstg = single ? 1 : 6;
    const float c21 = single == 1 ? 0.f : (single == 2 ? 1.f : TS_A21);
    const float* Uin = cur ? a.U[1] : a.U[0];
    const float* K1in = cur ? a.K1[1] : a.K1[0];
    float* Uout = cur ? a.U[0] : a.U[1];
    float* K1out = cur ? a.K1[0] : a.K1[1];

    float errsum = 0.f, badcnt = 0.f;
    auto red8 = [&](int kind) {
        const float* r = lds + s3b::RED + (kind * 32 + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    auto read_scalars = [&]() {
        const float e2 = red8(0), ld = red8(1), n2 = red8(2);
        return f32x4{ld, norm_z ? __builtin_sqrtf(e2) : 0.f, norm_j ? __builtin_sqrtf(n2) : 0.f, 0.f};
    };
    float* redw = lds + s3b::RED + smp * 8 + 4 * t + q;
    float* g3w = lds + s3b::G3 + smp * 40 + r0;
    const float* bias = lds + s3b::BIAS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // B operands: lane (sample s of half A, k = 8q ..): byte offsets into an image; half B = 16 rows on
    // (swizzled rows: k-block kb of a row is reached by XOR 64 kb on the byte offset)
    const int nb_rd = s * s3b::NS + 16 * (q ^ nsw), wb_rd = s * s3b::WS + 16 * (q ^ s);
    // results: lane (sample s, rows 16 wave + 4q ..) of the wide images
    const int wb_wr = s * s3b::WS + 16 * ((2 * wave + (q >> 1)) ^ s) + 8 * (q & 1);
    constexpr int HBW = 16 * s3b::WS, HBN = 16 * s3b::NS;

    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b0 = tile * 32 + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        f32x4 uz, kz[7], un = zero4;                       // Runge-Kutta rows of z (waves 0-3): u, k1..k7, u_new
#pragma unroll
        for (int j = 0; j < 7; ++j) kz[j] = zero4;
        if (tile == blockIdx.x) {                              // requested at kernel entry
            uz = ld4_mask(cur ? ru[1] : ru[0], cu);
            kz[0] = ld4_mask(cur ? rk[1] : rk[0], cu);
            if (sown) { sc_set(0, sc_get(cur ? 4 : 2)); sc_set(1, sc_get(cur ? 5 : 3)); }
        } else {
            ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
            const f32x4 e_ = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
            const f32x4 u_ = ld4_issue_w(Uin + gcol + r0, cu, img3, wide), k_ = ld4_issue_w(K1in + gcol + r0, cu, img3, wide);
            const f32x4 s0 = ld3_issue(Uin + gcol + n_in, cs, img3), s1 = ld3_issue(K1in + gcol + n_in, cs, img3);
            const f32x4 ev = ld4_mask(e_, ce);
            *(f32x4*)epw = ev;
            s3b_store4(ldsb + s3b::T0B + nw, s3b::NP, ev);
            uz = ld4_mask(u_, cu); kz[0] = ld4_mask(k_, cu);
            if (sown) { sc_set(0, ld4_mask(s0, cs)); sc_set(1, ld4_mask(s1, cs)); }
        }
        if (single == 1) { kz[0] = zero4; if (sown) sc_set(1, zero4); }      // there is no k1 yet
        if (zown) {
            un = uz + (hstep * c21) * kz[0];               // state of evaluation 1: U_2 = u + h a21 k1
            s3b_store4(x0w, s3b::NP, un);
        }
        s3_bar();

        // tau_3 = sigma'_3 .* (W3 tau_2) on waves 4-7: the product is formed in interval 2; sigma'_3 comes from the wave
        // that formed zdot of the same rows, one barrier later
        f32x4 tacc = zero4;
        auto finish_tau = [&]() {
            const f32x4 tj = tacc * *(const f32x4*)g3w;
            redw[32 * 8] = -s3_dot4(tj, *(const f32x4*)epw);
            redw[2 * 32 * 8] = s3_dot4(tj, tj);
        };

        constexpr bool S3B_COND = false;                   // (conditional models: k_mfma's step launches, or the one-launch solve)
        constexpr bool S3JB_RECORDS = false;               // (recording launches stay on k_mfma)
        float* const dmpw = nullptr; const size_t dmp_stride = 0;
        (void)dmpw; (void)dmp_stride;
#include "cnf_step3jb_eval.inc"
        if (single) {
            float* out = (single == 1 ? a.du : a.Ks0) + (size_t)(tile * 32 + 16 * hf + s) * D;
            auto norms = [&](const f32x4& u4, const f32x4& f0, const f32x4& f1, int nvalid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nvalid) {
                        const float sk = fmaf(fabsf(u4[c]), reltol, abstol);
                        if (single == 1) {
                            const float x = u4[c] / sk, y = f1[c] / sk;
                            errsum = fmaf(x, x, errsum); badcnt = fmaf(y, y, badcnt);
                        } else {
                            const float x = (f1[c] - f0[c]) / sk;
                            errsum = fmaf(x, x, errsum);
                        }
                    }
                }
            };
            if (live && zown) {
                st4(out + r0, kz[1], nv);                                      // k2 slot = this evaluation's zdot
                if (a.init_phase >= 0) norms(uz, kz[0], kz[1], nv);
            }
            if (live && sown) {
                const f32x4 f1 = read_scalars();
                out[n_in] = f1.x; out[n_in + 1] = f1.y; out[n_in + 2] = f1.z;
                if (a.init_phase >= 0) norms(sc_get(0), sc_get(1), f1, 3);
            }
        }
        // ---- error estimate and outputs: u_new (= the state evaluation 6 ran at) and k7 ----
        if (!single && live && zown) {
            const f32x4 ez = TS_BT1 * kz[0] + TS_BT2 * kz[1] + TS_BT3 * kz[2] + TS_BT4 * kz[3] + TS_BT5 * kz[4] +
                             TS_BT6 * kz[5] + TS_BT7 * kz[6];
#pragma unroll
            for (int c = 0; c < 4; ++c) {                  // rows beyond n_in: u = k = 0 -> contribute exactly 0
                const float scl = fmaf(fmaxf(fabsf(uz[c]), fabsf(un[c])), reltol, abstol);
                const float x = c < nv ? hstep * ez[c] / scl : 0.f;
                errsum = fmaf(x, x, errsum);
                badcnt += (c < nv && !(fabsf(un[c]) <= 3.0e38f)) ? 1.f : 0.f;
            }
            const size_t gc = (size_t)(tile * 32 + 16 * hf + s) * D;
            if (nv >= 4) { st4_wide(Uout + gc + r0, un); st4_wide(K1out + gc + r0, kz[6]); }
            else { st4(Uout + gc + r0, un, nv); st4(K1out + gc + r0, kz[6], nv); }
        }
        if (!single && live && sown) {
            f32x4 ks[7];
#pragma unroll
            for (int j = 0; j < 6; ++j) ks[j] = sc_get(1 + j);
            const f32x4 us = sc_get(0);
            ks[6] = read_scalars();                        // k7 of the scalar rows, straight from the partials
            const f32x4 uns = us + hstep * stage_acc4<6>(ks);
            err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, 3);
            const size_t gc = (size_t)(tile * 32 + 16 * hf + s) * D;
            float* Un = Uout + gc + n_in;
            float* K7 = K1out + gc + n_in;
            Un[0] = uns.x; Un[1] = uns.y; Un[2] = uns.z;
            K7[0] = ks[6].x; K7[1] = ks[6].y; K7[2] = ks[6].z;
        }
        s3_bar();                                          // this tile's RED / SC reads precede the next tile's writes
    }
    // deterministic block reduction of the error partial (fixed tree, fixed order)
    errsum = s3_wave_sum(errsum);
    badcnt = s3_wave_sum(badcnt);
    if (lane == 0) { msc[wave] = errsum; msc[16 + wave] = badcnt; }
    s3_bar();
    if (tid == 0) {
        float e = 0.f, b = 0.f;
        for (int w = 0; w < 8; ++w) { e += msc[w]; b += msc[16 + w]; }
        if (!single) {
            a.partials[2 * blockIdx.x] = e;
            a.partials[2 * blockIdx.x + 1] = b;
        } else if (a.init_phase >= 0) {
            __hip_atomic_store(a.partials + 2 * blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.partials + 2 * blockIdx.x + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            msc[40] = (tk == gridDim.x - 1) ? 1.f : 0.f;
            if (tk == gridDim.x - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (single && a.init_phase >= 0) {
        s3_bar();
        if (msc[40] != 0.f) {                        // this workgroup drew the last ticket: all its threads reduce
            float q0 = 0.f, q1 = 0.f;
            for (int i = tid; i < (int)gridDim.x; i += 512) {
                q0 += __hip_atomic_load(a.partials + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q1 += __hip_atomic_load(a.partials + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            q0 = s3_wave_sum(q0); q1 = s3_wave_sum(q1);
            if (lane == 0) { msc[wave] = q0; msc[16 + wave] = q1; }
            s3_bar();
            if (tid == 0) {
                float p0 = 0.f, p1 = 0.f;
                for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
                ctrl_phase(a.st_out, single - 1, p0, p1, a.n_total);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_step3b -- the VJP step kernel (k_step3) on six-term bf16 products.  What differs from k_step3jb: both orientations of
// W2 and W3^T / W1 tiles resident as split fragments (120 VGPRs), the K = 128 operands of the two narrow products (W3,
// W1^T) as split images in LDS, g2 / g1 written IN PLACE over h2 / h1 (sigma' is taken from the three pieces' exact sum),
// the K = 32 operands (state, g3) kept in fp32 and split on the way into the MFMA, the probe rows in registers.
// Six barrier intervals per evaluation, both sample halves inside each (they share the A fragments).
// ---------------------------------------------------------------------------------------------------------------
namespace s3v {
constexpr int SKZ = 264;
constexpr int KZ = 0, RED = KZ + 32 * SKZ, SC = RED + 3 * 32 * 8, BIAS = SC + 32 * 24;
constexpr int MISC = BIAS + 2 * 128 + 32, FP_END = MISC + 64;
// Split images [piece][32 rows][K bf16], rows WITHOUT padding, the 16-byte chunks of a row XOR-swizzled by the row: chunk c
// of row r sits at chunk position c ^ (r & 15) (K = 128: 16 chunks) / c ^ ((-(r >> 2)) & 3) (K = 32: 4 chunks).  An operand
// read (lane (q, s): chunk q + 4 kb of row s, ds_read_b128) then has no bank conflict in any of the instruction's four
// 16-lane groups; padded rows cannot do that together with 16-byte alignment (272-byte rows: two-way, i.e. the LDS pipe as
// busy as the matrix pipe in the wide products).  The 8-byte epilogue stores stay two-way.
constexpr int WS = 256, WP = 32 * WS, WI = 3 * WP;
constexpr int NS = 64, NP = 32 * NS, NI = 3 * NP;
constexpr int H1G = FP_END * 4, H2G = H1G + WI, W3I = H2G + WI, W1TI = W3I + WI, X0S = W1TI + WI, G3S = X0S + NI;
constexpr int TOTAL_BYTES = G3S + NI;
static_assert(H1G % 16 == 0 && TOTAL_BYTES <= 160 * 1024, "LDS plan");
static_assert(WI == s3g::WI, "global image");
}  // namespace s3v

__global__ void __launch_bounds__(512, 2) k_step3b(MfmaArgs a, const char* __restrict__ imgb, int n_in, int norm_z,
                                                   int norm_j, const S3Tab tab, int single) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const float* img3 = reinterpret_cast<const float*>(imgb);      // (a valid address for masked loads)
    S3_ARGS_UP_FRONT(a, imgb);
    const StepState* st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const int t = wave & 1, hf = (wave >> 1) & 1;         // narrow phases: row tile and sample half of this wave
    const bool zown = wave < 4;                           // waves 0-3 produce zdot: they hold the z rows of the state
    const bool sown = !zown && t == 0 && q == 0;          // waves 4, 6: lane s holds the scalar rows of sample 16 hf + s
    const int smp = 16 * hf + s;                          // sample of this lane in the narrow phases
    const int r0 = 16 * t + 4 * q;                        // first of its 4 rows there
    const int nv = n_in - r0;                             // valid rows among them (may be <= 0 or > 4)
#ifdef S3_STAMPS
    unsigned long long s3acc[36] = {0};
    unsigned long long s3last = __builtin_amdgcn_s_memtime();
    const unsigned long long s3start = s3last;
#endif
    // ---- everything the launch needs from memory is requested up front, in ONE round trip, and nothing is consumed
    // before all of it is in flight: the integrator state words first (the `done` test and the controller need them
    // soonest), then the error partials, then the tile and the weights.
    const StepState* stw = s3_state_words(st);
    const int v_done = stw->done, v_cur = stw->cur;
    const float v_h = stw->h, v_abstol = stw->abstol, v_reltol = stw->reltol;
    // (one unconditional load per thread: a loop here would wait for its data before anything below is even requested)
    const float* ppin = a.apply_ctrl ? a.partials_in : img3;
    const float2 pp = *reinterpret_cast<const float2*>(ppin + 2 * min(tid, (int)gridDim.x - 1));
    // Issue order = return order: this workgroup's first tile from BOTH buffer sets (which one is current is the
    // controller's decision), the biases, the resident fragments (fp32, split as they arrive), the two LDS images last.
    const int ntile = (a.B + 32 - 1) / 32;
    const bool wide = (n_in & 3) == 0;                     // every lane's four rows are all valid or all padding
    f32x4 ru[2], rk[2], re, rs[2][2];
    int ce = 0, cu = 0, cs = 0;
    {
        const int b0 = blockIdx.x * 32 + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
        re = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ru[c] = ld4_issue_w(a.U[c] + gcol + r0, cu, img3, wide);
            rk[c] = ld4_issue_w(a.K1[c] + gcol + r0, cu, img3, wide);
            rs[c][0] = ld3_issue(a.U[c] + gcol + n_in, cs, img3);
            rs[c][1] = ld3_issue(a.K1[c] + gcol + n_in, cs, img3);
        }
    }
    // Weights (k_pack_step3b).  Resident fragments of this wave: its 16-row tile of W1 and of W3^T (K = 32: one k-block),
    // of W2 and of W2^T (four k-blocks).  The K = 128 operands of the narrow products (rows of W3, rows of W1^T) stay in
    // LDS as split images, copied as they are stored (LDS-DMA).
    constexpr int NCI = 2 * s3v::WI / 16, NCB = (2 * 128 + 32) / 4;
    static_assert(NCI % 64 == 0, "whole wave instructions");
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef const __attribute__((address_space(1))) char* glb_c;
    const f32x4 sgb = reinterpret_cast<const f32x4*>(imgb + s3g::BIASB)[min(tid, NCB - 1)];
    // the resident fragments arrive in fp32 and are split here, in arrival order, while the rest of the stream is in flight
    S3bOp wF1, wF2[4], wB3, wB2[4];
    {
        const char* fw = imgb + s3g::F32 + (size_t)wave * 10 * 2048 + 16 * lane;
        constexpr int AH = 5;                                  // fragments requested ahead of the one being split
        f32x4 raw[10][2];
#pragma unroll
        for (int f = 0; f < AH; ++f) { raw[f][0] = *(const f32x4*)(fw + f * 2048); raw[f][1] = *(const f32x4*)(fw + f * 2048 + 1024); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 10; ++f) {
            if (f + AH < 10) {
                raw[f + AH][0] = *(const f32x4*)(fw + (f + AH) * 2048);
                raw[f + AH][1] = *(const f32x4*)(fw + (f + AH) * 2048 + 1024);
            }
            S3bOp o = s3b_split8(raw[f][0], raw[f][1]);
            s3b_pin(o);                                        // (the split stays here, between the two scheduling barriers)
            if (f == 0) wF1 = o; else if (f < 5) wF2[f - 1] = o; else if (f == 5) wB3 = o; else wB2[f - 6] = o;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the two LDS images last: while LDS-DMA pieces are outstanding the compiler waits with vmcnt(0) for ANY loaded register
#pragma unroll
    for (int i = 0; i < (NCI + 511) / 512; ++i) {
        const int c = 512 * i + 64 * wave;                     // wave-uniform chunk (16 B) index
        if (c < NCI)
            __builtin_amdgcn_global_load_lds((glb_c)(imgb + s3g::W3I + 16 * (c + lane)), (lds_c)(ldsb + s3v::W3I + 16 * c), 16, 0, 0);
    }
    float* sc = lds + s3v::SC + smp * 24;
    auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
    auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
    __builtin_amdgcn_sched_barrier(0);                     // nothing above is consumed before all of it is requested
    S3T(23);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's LDS-DMA pieces have landed
    __builtin_amdgcn_sched_barrier(0);
    S3T(24);
    // (wave-uniform values into scalar registers: the buffer pointers selected from `cur` stay out of the vector file)
    const int st_done = __builtin_amdgcn_readfirstlane(v_done), st_cur = __builtin_amdgcn_readfirstlane(v_cur);
    const float st_h = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_h)));
    const float st_abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_abstol)));
    const float st_reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v_reltol)));
    if (st_done) {       // launches queued past the end of the solve: keep the state chain intact and leave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the LDS-DMA pieces must have landed before the wave ends)
        if (a.apply_ctrl && blockIdx.x == 0 && tid == 0) { *a.st_out = *st; publish_mirror(a, *st); }
        return;
    }
    // scalar rows of both candidates wait in LDS (slots 2..5 of the scalar-row state) for the controller's choice
    if (sown) {
        sc_set(2, ld4_mask(rs[0][0], cs)); sc_set(3, ld4_mask(rs[0][1], cs));
        sc_set(4, ld4_mask(rs[1][0], cs)); sc_set(5, ld4_mask(rs[1][1], cs));
    }
    // the probe rows go straight to their LDS image (waves 0-3 need them for g3, waves 4-7 for the trace row)
    f32x4 epsr = ld4_mask(re, ce);                         // this lane's 4 probe rows of sample smp (g3 on waves 0-3, trace row on 4-7)
    float* msc = lds + s3v::MISC;
    if (a.apply_ctrl) {
        float cp0 = tid < (int)gridDim.x ? pp.x : 0.f, cp1 = tid < (int)gridDim.x ? pp.y : 0.f;
        for (int i = tid + 512; i < (int)gridDim.x; i += 512) { cp0 += a.partials_in[2 * i]; cp1 += a.partials_in[2 * i + 1]; }
        cp0 = s3_wave_sum(cp0); cp1 = s3_wave_sum(cp1);
        if (lane == 0) { msc[wave] = cp0; msc[16 + wave] = cp1; }
    }
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + s3v::BIAS)[tid] = sgb;
    s3_bar();                                              // LDS images and partial sums complete
    S3T(25);
    int cur = st_cur;
    float hstep = st_h, abstol = st_abstol, reltol = st_reltol;
    if (a.apply_ctrl && tid == 0) {
        // In-kernel step controller (as in k_mfma): every workgroup reduces the same partials in the same order
        // and takes the same decision; block 0 publishes the new state for the next launch and the host mirror.
        float p0 = 0.f, p1 = 0.f;
        for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
        StepState ns = *st;                                // (cached by now: the state words above came from these lines)
        ctrl_after_step(&ns, p0, p1, a.n_total);
        if (blockIdx.x == 0) { *a.st_out = ns; publish_mirror(a, ns); }
        msc[32] = __int_as_float(ns.cur); msc[33] = ns.h; msc[34] = ns.abstol; msc[35] = ns.reltol;
        msc[36] = __int_as_float(ns.done);
    }
    S3T(26);
    s3_bar();                                              // controller done; the staging area is free
    S3T(27);
    if (a.apply_ctrl) {
        cur = __builtin_amdgcn_readfirstlane(__float_as_int(msc[32]));
        hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[33])));
        abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[34])));
        reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[35])));
        if (__float_as_int(msc[36])) return;        // the controller just finished the solve
    }
    // (explicit selects: indexing the kernel-argument arrays with a run-time value makes the compiler fetch the
    // pointer from the argument segment through a vector load -- a memory round trip in front of the state loads)
    // single != 0: ONE evaluation instead of a step attempt -- the two launches of the automatic initial dt
    // (1: f(u) -> a.du with the norms of phase 0; 2: f(u + h k1) -> a.Ks0 with the norm of phase 1), same prologue,
    // same evaluation code; the last workgroup to finish runs the controller phase (as k_mfma does for them)
    const int nstg = single ? 1 : 6;
    const float c21 = single == 1 ? 0.f : (single == 2 ? 1.f : TS_A21);
    const float* Uin = cur ? a.U[1] : a.U[0];
    const float* K1in = cur ? a.K1[1] : a.K1[0];
    float* Uout = cur ? a.U[0] : a.U[1];
    float* K1out = cur ? a.K1[0] : a.K1[1];

    float errsum = 0.f, badcnt = 0.f;
    // the 8 partials (2 row tiles x 4 lanes) of one sample and one kind sit side by side: two b128 reads each
    auto red8 = [&](int kind) {
        const float* r = lds + s3v::RED + (kind * 32 + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    auto read_scalars = [&]() {
        const float e2 = red8(0), ld = red8(1), n2 = red8(2);
        return f32x4{ld, norm_z ? __builtin_sqrtf(e2) : 0.f, norm_j ? __builtin_sqrtf(n2) : 0.f, 0.f};
    };
    float* redw = lds + s3v::RED + smp * 8 + 4 * t + q;                 // this lane's slot of kind 0 (+ 256 per kind)
    // B operands from a split image: lane (sample s of half A, k = 8q ..); half B = 16 rows on.  Results: lane (sample s,
    // rows 16 wave + 4q ..) of the wide images
    // (swizzled rows: k-block kb of a row is reached by XOR 64 kb on the byte offset, not by adding to it)
    const int wb_rd = s * s3v::WS + 16 * (q ^ s);
    const int wb_wr = s * s3v::WS + 16 * ((2 * wave + (q >> 1)) ^ s) + 8 * (q & 1);
    constexpr int HBW = 16 * s3v::WS;
    // narrow products: A = rows 16t + s of W3 (waves 0-3) / of W1^T (waves 4-7), B = this wave's half of h2 / g1
    const char* nrA = ldsb + (zown ? s3v::W3I : s3v::W1TI) + 16 * t * s3v::WS;      // + (wb_rd ^ 64 kb): row 16t + s
    const char* nrB = ldsb + (zown ? s3v::H2G : s3v::H1G) + 16 * hf * s3v::WS;      // + (wb_rd ^ 64 kb): row smp
    const int nsw = (-(s >> 2)) & 3;                                    // chunk swizzle of the K = 32 images (rows s, 16 + s)
    const int nw = smp * s3v::NS + 16 * ((2 * t + (q >> 1)) ^ nsw) + 8 * (q & 1);
    char* x0w = ldsb + s3v::X0S + nw;                                   // this lane's 4 rows of the state / g3 images
    char* g3w = ldsb + s3v::G3S + nw;
    const int nb_rd = s * s3v::NS + 16 * (q ^ nsw);                     // their B operands: lane (sample s of half A, k = 8q ..)
    constexpr int HBN = 16 * s3v::NS;
    // Runge-Kutta state of the z rows r0..r0+3 of sample smp, written and read by this lane only:
    float* rkw = lds + s3v::KZ + smp * s3v::SKZ + r0;                   // u at rkw, k1 at rkw + 32,
    float* kzw = rkw + 64;                                            // k_{j+2} at kzw + 32 j (j = 0..5)
    const float* bias = lds + s3v::BIAS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    S3T(20);
    // a tile's z rows into the Runge-Kutta state and the state image of its first evaluation
    auto tile_in = [&](const f32x4& uz, f32x4 k1z) {
        if (single == 1) { k1z = zero4; if (sown) sc_set(1, k1z); }          // there is no k1 yet
        if (zown) {
            s3b_store4(x0w, s3v::NP, uz + (hstep * c21) * k1z);              // state of evaluation 1: U_2 = u + h a21 k1
            *(f32x4*)rkw = uz;
            *(f32x4*)(rkw + 32) = k1z;
#pragma unroll
            for (int j = 0; j < 6; ++j) *(f32x4*)(kzw + 32 * j) = zero4;     // k2..k7: not produced yet
        }
    };
    // the first tile was requested at kernel entry; it goes to LDS here, so that none of it is carried into the loop
    if (sown) { sc_set(0, sc_get(cur ? 4 : 2)); sc_set(1, sc_get(cur ? 5 : 3)); }
    tile_in(ld4_mask(cur ? ru[1] : ru[0], cu), ld4_mask(cur ? rk[1] : rk[0], cu));
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b0 = tile * 32 + 16 * hf;
        const bool live = s < max(0, min(16, a.B - b0));
        const size_t gcol = (size_t)(b0 + s) * D;
        if (tile != (int)blockIdx.x) {
            ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
            const f32x4 e_ = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
            const f32x4 u_ = ld4_issue_w(Uin + gcol + r0, cu, img3, wide), k_ = ld4_issue_w(K1in + gcol + r0, cu, img3, wide);
            const f32x4 s0 = ld3_issue(Uin + gcol + n_in, cs, img3), s1 = ld3_issue(K1in + gcol + n_in, cs, img3);
            epsr = ld4_mask(e_, ce);
            if (sown) { sc_set(0, ld4_mask(s0, cs)); sc_set(1, ld4_mask(s1, cs)); }
            tile_in(ld4_mask(u_, cu), ld4_mask(k_, cu));
        }
        s3_bar();
        S3T(21);

        constexpr bool S3B_RECORDS = false;                // (recording launches stay on k_mfma)
        constexpr bool S3B_COND = false;                   // (conditional models: k_mfma's step launches, or the one-launch solve)
        float* const dmpw = nullptr; const size_t dmp_stride = 0;
        (void)dmpw; (void)dmp_stride;
#include "cnf_step3b_eval.inc"
        if (single) {
            // ---- one evaluation: f -> out, and the norms of the initial-dt phase over the rows this lane owns ----
            float* out = (single == 1 ? a.du : a.Ks0) + (size_t)(tile * 32 + 16 * hf + s) * D;
            auto norms = [&](const f32x4& u4, const f32x4& f0, const f32x4& f1, int nvalid) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nvalid) {
                        const float sk = fmaf(fabsf(u4[c]), reltol, abstol);
                        if (single == 1) {
                            const float x = u4[c] / sk, y = f1[c] / sk;
                            errsum = fmaf(x, x, errsum); badcnt = fmaf(y, y, badcnt);
                        } else {
                            const float x = (f1[c] - f0[c]) / sk;
                            errsum = fmaf(x, x, errsum);
                        }
                    }
                }
            };
            if (live && zown) {
                const f32x4 f1 = *(const f32x4*)kzw;                    // k2 slot = this evaluation's zdot
                st4(out + r0, f1, nv);
                if (a.init_phase >= 0) norms(*(const f32x4*)rkw, *(const f32x4*)(rkw + 32), f1, nv);
            }
            if (live && sown) {
                const f32x4 f1 = read_scalars();
                out[n_in] = f1.x; out[n_in + 1] = f1.y; out[n_in + 2] = f1.z;
                if (a.init_phase >= 0) norms(sc_get(0), sc_get(1), f1, 3);
            }
        }
        // ---- error estimate and outputs: u_new (= the state evaluation 6 ran at) and k7 ----
        if (!single && live && zown) {
            const f32x4 k7z = *(const f32x4*)(kzw + 32 * 5), uz_ = *(const f32x4*)rkw;
            const f32x4 un = s3b_load4(x0w, s3v::NP);     // U_7 = u_new: the state the last evaluation ran at (the pieces sum exactly)
            f32x4 ez = TS_BT1 * *(const f32x4*)(rkw + 32) + TS_BT7 * k7z;
            ez += TS_BT2 * *(const f32x4*)(kzw) + TS_BT3 * *(const f32x4*)(kzw + 32) + TS_BT4 * *(const f32x4*)(kzw + 64) +
                  TS_BT5 * *(const f32x4*)(kzw + 96) + TS_BT6 * *(const f32x4*)(kzw + 128);
#pragma unroll
            for (int c = 0; c < 4; ++c) {                  // rows beyond n_in: u = k = 0 -> contribute exactly 0
                const float scl = fmaf(fmaxf(fabsf(uz_[c]), fabsf(un[c])), reltol, abstol);
                const float x = c < nv ? hstep * ez[c] / scl : 0.f;
                errsum = fmaf(x, x, errsum);
                badcnt += (c < nv && !(fabsf(un[c]) <= 3.0e38f)) ? 1.f : 0.f;
            }
            const size_t gc = (size_t)(tile * 32 + 16 * hf + s) * D;
            float* Un = Uout + gc + r0;
            float* K7 = K1out + gc + r0;
            if (nv >= 4) { st4_wide(Un, un); st4_wide(K7, k7z); }
            else { st4(Un, un, nv); st4(K7, k7z, nv); }
        }
        if (!single && live && sown) {
            f32x4 ks[7];
#pragma unroll
            for (int j = 0; j < 6; ++j) ks[j] = sc_get(1 + j);
            const f32x4 us = sc_get(0);
            ks[6] = read_scalars();                        // k7 of the scalar rows, straight from the partials
            const f32x4 uns = us + hstep * stage_acc4<6>(ks);
            err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, 3);
            const size_t gc = (size_t)(tile * 32 + 16 * hf + s) * D;
            float* Un = Uout + gc + n_in;
            float* K7 = K1out + gc + n_in;
            Un[0] = uns.x; Un[1] = uns.y; Un[2] = uns.z;
            K7[0] = ks[6].x; K7[1] = ks[6].y; K7[2] = ks[6].z;
        }
        s3_bar();                                          // this tile's RED / SC / KZ reads precede the next tile's writes
    }
    // deterministic block reduction of the error partial (fixed tree, fixed order)
    errsum = s3_wave_sum(errsum);
    badcnt = s3_wave_sum(badcnt);
    if (lane == 0) { msc[wave] = errsum; msc[16 + wave] = badcnt; }
    s3_bar();
    if (tid == 0) {
        float e = 0.f, b = 0.f;
        for (int w = 0; w < 8; ++w) { e += msc[w]; b += msc[16 + w]; }
        if (!single) {
            a.partials[2 * blockIdx.x] = e;
            a.partials[2 * blockIdx.x + 1] = b;
        } else if (a.init_phase >= 0) {
            // initial-dt phase: partials through agent-scope atomics, then a ticket; whoever draws the last one sums all
            // partials (fixed order) and runs the controller phase -- no separate launches
            __hip_atomic_store(a.partials + 2 * blockIdx.x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.partials + 2 * blockIdx.x + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            msc[40] = (tk == gridDim.x - 1) ? 1.f : 0.f;
            if (tk == gridDim.x - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (single && a.init_phase >= 0) {
        s3_bar();
        if (msc[40] != 0.f) {                        // this workgroup drew the last ticket: all its threads reduce
            float q0 = 0.f, q1 = 0.f;
            for (int i = tid; i < (int)gridDim.x; i += 512) {
                q0 += __hip_atomic_load(a.partials + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q1 += __hip_atomic_load(a.partials + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            q0 = s3_wave_sum(q0); q1 = s3_wave_sum(q1);
            if (lane == 0) { msc[wave] = q0; msc[16 + wave] = q1; }
            s3_bar();
            if (tid == 0) {
                float p0 = 0.f, p1 = 0.f;
                for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
                ctrl_phase(a.st_out, single - 1, p0, p1, a.n_total);
            }
        }
    }
#ifdef S3_STAMPS
    S3T(22);
    if (blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 5))
        printf("k_step3b wave %d total %llu | issue %llu landed %llu bar %llu ctrl %llu bar %llu rest %llu tileprep %llu tail %llu | F1 %llu+%llu F2 %llu+%llu F3 %llu+%llu B3 %llu+%llu B2 %llu+%llu B1 %llu+%llu\n",
               wave, s3last - s3start, s3acc[23], s3acc[24], s3acc[25], s3acc[26], s3acc[27], s3acc[20], s3acc[21], s3acc[22], s3acc[0], s3acc[1], s3acc[2], s3acc[3], s3acc[4], s3acc[5],
               s3acc[6], s3acc[7], s3acc[8], s3acc[9], s3acc[10], s3acc[11]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// k_solve3b -- the WHOLE adaptive solve of one shard in one launch: k_step3b's evaluation code inside the step loop.  What
// a launch per attempt pays every time -- the 212 KB weight stream, the state round trip through HBM, the error partials
// of the previous launch, kernel start and drain -- is paid once: weights, Runge-Kutta rows and probe rows stay in
// registers / LDS for all attempts, and the workgroups MEET once per attempt to exchange their error partials (two floats
// each): every workgroup stores its two partials as 8-byte words {meeting index, float} (agent-scope relaxed atomics, two
// buffers by index parity), thread i polls workgroup i's words until they carry this meeting's index (one 16-byte load
// past the caches per poll), then every workgroup adds the same partials in the same order and runs the same controller,
// as the launches of the streamed driver do.  No ticket, no fence, no cache flush: nothing else crosses workgroups.
// Needs every workgroup resident: an ORDINARY launch whose grid the host bounds by what the device holds at once
// (occupancy x CUs: step3b_solve_resident); every wait is bounded (sv.wait_ticks of the 100 MHz clock, sv.spin_limit polls), so a workgroup that never
// arrives -- CUs held by another stream, process or CU mask -- ends the launch with the abort word and `done` = 0, and
// the host runs the solve again on the streamed driver (cnf_abi.hip).  The two drivers run the same arithmetic from
// separately compiled code: identical step counts on well-conditioned cases, results equal to the solver tolerance, not
// bit for bit.
// ---------------------------------------------------------------------------------------------------------------
// RECORD (gradient path): every attempt files u_n and its stage states U_2..U_6 (z rows) in the trajectory slot of step
// `naccept` (a.dump, as the recording launches of k_mfma do), its signed step size in a.hs_out.
template <bool RECORD, bool MULTI, bool COND>
__global__ void __launch_bounds__(512, 2) k_solve3b(MfmaArgs a, const char* __restrict__ imgb, int n_in, int norm_z,
                                                    int norm_j, const S3Tab tab, Solve3Args sv) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const float* img3 = reinterpret_cast<const float*>(imgb);      // (a valid address for masked loads)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const int t = wave & 1, hf = (wave >> 1) & 1;
    const bool zown = wave < 4;
    const bool sown = !zown && t == 0 && q == 0;
    const int smp = 16 * hf + s;
    const int r0 = 16 * t + 4 * q;
    const int nv = n_in - r0;
    const bool wide = (n_in & 3) == 0;
    const int ntile = (a.B + 31) / 32;
    constexpr bool multi = MULTI;                          // several tiles per workgroup: the state lives in a.U / a.K1 (k_solve3jb)
    int b0 = blockIdx.x * 32 + 16 * hf;
    bool live = s < max(0, min(16, a.B - b0));
    size_t gcol = (size_t)(b0 + s) * D;
    int ce = live ? nv : 0, cu = (zown && live) ? nv : 0, cs = (sown && live) ? 3 : 0;
    auto set_tile = [&](int tile) {
        b0 = tile * 32 + 16 * hf;
        live = s < max(0, min(16, a.B - b0));
        gcol = (size_t)(b0 + s) * D;
        ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
    };
    if (sv.t_out && blockIdx.x == 0 && tid == 0) sv.t_out[0] = __builtin_amdgcn_s_memrealtime();
    // ---- one round trip: this tile's state and probe rows, the weights ----
    const f32x4 re = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
    f32x4 ru, rs;
    if (sv.xs) {                                           // u0 = (xs; zeros for the augmented and the scalar rows)
        const float* xc = sv.xs + (size_t)(b0 + s) * sv.nvars;
#pragma unroll
        for (int j = 0; j < 4; ++j) ru[j] = (cu > j && r0 + j < sv.nvars) ? xc[r0 + j] : 0.f;
        rs = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        ru = ld4_issue_w(a.U[0] + gcol + r0, cu, img3, wide);
        rs = ld3_issue(a.U[0] + gcol + n_in, cs, img3);
    }
    constexpr int NCI = 2 * s3v::WI / 16, NCB = (2 * 128 + 32) / 4;
    typedef __attribute__((address_space(3))) char* lds_c;
    typedef const __attribute__((address_space(1))) char* glb_c;
    const f32x4 sgb = reinterpret_cast<const f32x4*>(imgb + s3g::BIASB)[min(tid, NCB - 1)];
    // the resident fragments arrive in fp32 and are split here, in arrival order, while the rest of the stream is in flight
    S3bOp wF1, wF2[4], wB3, wB2[4];
    {
        const char* fw = imgb + s3g::F32 + (size_t)wave * 10 * 2048 + 16 * lane;
        constexpr int AH = 5;                                  // fragments requested ahead of the one being split
        f32x4 raw[10][2];
#pragma unroll
        for (int f = 0; f < AH; ++f) { raw[f][0] = *(const f32x4*)(fw + f * 2048); raw[f][1] = *(const f32x4*)(fw + f * 2048 + 1024); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 10; ++f) {
            if (f + AH < 10) {
                raw[f + AH][0] = *(const f32x4*)(fw + (f + AH) * 2048);
                raw[f + AH][1] = *(const f32x4*)(fw + (f + AH) * 2048 + 1024);
            }
            S3bOp o = s3b_split8(raw[f][0], raw[f][1]);
            s3b_pin(o);                                        // (the split stays here, between the two scheduling barriers)
            if (f == 0) wF1 = o; else if (f < 5) wF2[f - 1] = o; else if (f == 5) wB3 = o; else wB2[f - 6] = o;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the two LDS images last: while LDS-DMA pieces are outstanding the compiler waits with vmcnt(0) for ANY loaded register
#pragma unroll
    for (int i = 0; i < (NCI + 511) / 512; ++i) {
        const int c = 512 * i + 64 * wave;                     // wave-uniform chunk (16 B) index
        if (c < NCI)
            __builtin_amdgcn_global_load_lds((glb_c)(imgb + s3g::W3I + 16 * (c + lane)), (lds_c)(ldsb + s3v::W3I + 16 * c), 16, 0, 0);
    }
    float* sc = lds + s3v::SC + smp * 24;
    auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
    auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's LDS-DMA pieces have landed
    __builtin_amdgcn_sched_barrier(0);
    f32x4 epsr = ld4_mask(re, ce);
    float* msc = lds + s3v::MISC;
    StepState* ns = reinterpret_cast<StepState*>(msc + 44);            // the integrator state (thread 0 runs the controller on it)
    static_assert(sizeof(StepState) <= 20 * sizeof(float), "fits the scratch words");
    if (tid == 0) *ns = sv.init;
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + s3v::BIAS)[tid] = sgb;
    const int single = 0;                                              // (the evaluation code is k_step3b's)
    (void)single;
#ifdef S3_STAMPS
    unsigned long long s3acc[36] = {0};
    unsigned long long s3last = __builtin_amdgcn_s_memtime();
#endif
    // the 8 partials (2 row tiles x 4 lanes) of one sample and one kind sit side by side: two b128 reads each
    auto red8 = [&](int kind) {
        const float* r = lds + s3v::RED + (kind * 32 + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    auto read_scalars = [&]() {
        const float e2 = red8(0), ld = red8(1), n2 = red8(2);
        return f32x4{ld, norm_z ? __builtin_sqrtf(e2) : 0.f, norm_j ? __builtin_sqrtf(n2) : 0.f, 0.f};
    };
    float* redw = lds + s3v::RED + smp * 8 + 4 * t + q;                 // this lane's slot of kind 0 (+ 256 per kind)
    // B operands from a split image: lane (sample s of half A, k = 8q ..); half B = 16 rows on.  Results: lane (sample s,
    // rows 16 wave + 4q ..) of the wide images
    // (swizzled rows: k-block kb of a row is reached by XOR 64 kb on the byte offset, not by adding to it)
    const int wb_rd = s * s3v::WS + 16 * (q ^ s);
    const int wb_wr = s * s3v::WS + 16 * ((2 * wave + (q >> 1)) ^ s) + 8 * (q & 1);
    constexpr int HBW = 16 * s3v::WS;
    // narrow products: A = rows 16t + s of W3 (waves 0-3) / of W1^T (waves 4-7), B = this wave's half of h2 / g1
    const char* nrA = ldsb + (zown ? s3v::W3I : s3v::W1TI) + 16 * t * s3v::WS;      // + (wb_rd ^ 64 kb): row 16t + s
    const char* nrB = ldsb + (zown ? s3v::H2G : s3v::H1G) + 16 * hf * s3v::WS;      // + (wb_rd ^ 64 kb): row smp
    const int nsw = (-(s >> 2)) & 3;                                    // chunk swizzle of the K = 32 images (rows s, 16 + s)
    const int nw = smp * s3v::NS + 16 * ((2 * t + (q >> 1)) ^ nsw) + 8 * (q & 1);
    char* x0w = ldsb + s3v::X0S + nw;                                   // this lane's 4 rows of the state / g3 images
    char* g3w = ldsb + s3v::G3S + nw;
    const int nb_rd = s * s3v::NS + 16 * (q ^ nsw);                     // their B operands: lane (sample s of half A, k = 8q ..)
    constexpr int HBN = 16 * s3v::NS;
    // Runge-Kutta state of the z rows r0..r0+3 of sample smp, written and read by this lane only:
    float* rkw = lds + s3v::KZ + smp * s3v::SKZ + r0;                   // u at rkw, k1 at rkw + 32,
    float* kzw = rkw + 64;                                            // k_{j+2} at kzw + 32 j (j = 0..5)
    const float* bias = lds + s3v::BIAS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // this tile's rows
    if (zown) { *(f32x4*)rkw = ld4_mask(ru, cu); *(f32x4*)(rkw + 32) = zero4; }
    if (sown) { sc_set(0, ld4_mask(rs, cs)); sc_set(1, zero4); }
    s3_bar();                                              // LDS images, biases, state
    float hstep = ns->h, abstol = ns->abstol, reltol = ns->reltol;
    hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hstep)));
    abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(abstol)));
    reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(reltol)));
    int nsync = 0;                                         // meetings so far (the same count in every workgroup)
    // meetings held on this buffer by earlier launches: a device word, so that a launch can be queued behind another
    // before the host knows how many meetings that one will hold (read once; workgroup 0 advances it at the very end)
    const unsigned mbase = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sv.base_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // The workgroups' partials (e, b) -> the sums of all of them, in msc[32], msc[33] for thread 0 (what the prologue of a
    // step launch computes from the previous launch's partials, in the same order).  Returns false when a wait ran out.
    auto meet = [&](float e_lane, float b_lane) -> bool {
        float e = s3_wave_sum(e_lane), b = s3_wave_sum(b_lane);
        if (lane == 0) { msc[wave] = e; msc[16 + wave] = b; }
        s3_bar();
        // Every partial travels with the index of the meeting it belongs to in the same 8-byte word: a reader needs no
        // ticket -- thread i polls workgroup i's two words until both carry this meeting's index -- so a meeting costs one
        // store and one load round trip.  Two buffers by parity: a workgroup can be one meeting ahead of a reader, not two.
        unsigned long long* pb = reinterpret_cast<unsigned long long*>(sv.part) + (nsync & 1) * 1024;
        const unsigned tag = mbase + (unsigned)nsync + 1u;
        if (tid == 0) {
            float e8 = 0.f, b8 = 0.f;
            for (int w = 0; w < 8; ++w) { e8 += msc[w]; b8 += msc[16 + w]; }
            __hip_atomic_store(pb + 2 * blockIdx.x, ((unsigned long long)tag << 32) | __float_as_uint(e8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pb + 2 * blockIdx.x + 1, ((unsigned long long)tag << 32) | __float_as_uint(b8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        float cp0 = 0.f, cp1 = 0.f;
        int ok = 1;
        if (tid < (int)gridDim.x) {
            ok = 0;
            const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
            for (int spin = 0; spin < sv.spin_limit; ++spin) {
                // both words of workgroup `tid` with ONE 16-byte load past the caches (each half carries its own index, so a
                // torn pair is simply not accepted): half the polling traffic of two 8-byte atomic loads
                u32x4 wq;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(wq) : "v"(pb + 2 * tid) : "memory");
                const unsigned long long w0 = ((unsigned long long)wq.y << 32) | wq.x, w1 = ((unsigned long long)wq.w << 32) | wq.z;
                if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag) {
                    cp0 = __uint_as_float((unsigned)w0); cp1 = __uint_as_float((unsigned)w1); ok = 1;
                    break;
                }
                if ((spin & 255) == 255 && __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;   // bounded in TIME
                __builtin_amdgcn_s_sleep(4);
            }
            if (!ok) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cp0 = s3_wave_sum(cp0); cp1 = s3_wave_sum(cp1);
        const float bad = s3_wave_sum(ok ? 0.f : 1.f);
        s3_bar();                                          // (msc[0..7], [16..23] were read by thread 0 above)
        if (lane == 0) { msc[wave] = cp0; msc[16 + wave] = cp1; msc[24 + wave] = bad; }
        s3_bar();
        float nbad = 0.f;
        for (int w = 0; w < 8; ++w) nbad += msc[24 + w];
        if (tid == 0) {
            float p0 = 0.f, p1 = 0.f;
            for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
            msc[32] = p0; msc[33] = p1;
        }
        ++nsync;
        return nbad == 0.f;
    };
    // thread 0 ran a controller phase on *ns: the new step and tolerances to everyone
    auto share = [&]() {
        s3_bar();
        hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[36])));
        abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[37])));
        reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[38])));
        const int fl = __builtin_amdgcn_readfirstlane(__float_as_int(msc[39]));
        s3_bar();                                          // (the words are rewritten by the next phase)
        return fl;                                         // bit 0: done, bit 1: the attempt was accepted
    };
    auto post_ctrl = [&](int accepted) {                   // thread 0, after a controller phase
        msc[36] = ns->h; msc[37] = ns->abstol; msc[38] = ns->reltol;
        msc[39] = __int_as_float((ns->done ? 1 : 0) | (accepted ? 2 : 0));
    };
    // one evaluation at the state image in place; zdot -> the k2 slot, scalar rows from the RED partials
    int nstg = 1;
    float* dmpw = nullptr;                                 // RECORD: this lane's rows of the current step's slot (null: not filed)
    const size_t dmp_stride = a.dump_stride;
    auto evals = [&]() {
        constexpr bool S3B_RECORDS = RECORD;
        constexpr bool S3B_COND = COND;
#include "cnf_step3b_eval.inc"
    };
    // (x / sk)^2 onto acc, sk = atol + rtol |u|: the expressions of the single-evaluation launches, digit for digit
    auto add_norm = [&](float& acc, float u, float x) {
        const float sk = fmaf(fabsf(u), reltol, abstol);
        const float y = x / sk;
        acc = fmaf(y, y, acc);
    };
    // ---- several tiles per workgroup (as k_solve3jb): a tile's rows from / to the integrator's buffers ----
    int cur = 0;                                           // the buffer set that holds (u, k1)
    auto load_tile = [&](int tile, bool with_k1, bool from_xs) {
        set_tile(tile);
        const float* Uc = cur ? a.U[1] : a.U[0];
        const float* Kc = cur ? a.K1[1] : a.K1[0];
        const f32x4 e_ = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
        f32x4 u_, s_;
        if (from_xs) {
            const float* xc = sv.xs + (size_t)(b0 + s) * sv.nvars;
#pragma unroll
            for (int j = 0; j < 4; ++j) u_[j] = (cu > j && r0 + j < sv.nvars) ? xc[r0 + j] : 0.f;
            s_ = zero4;
        } else {
            u_ = ld4_issue_w(Uc + gcol + r0, cu, img3, wide);
            s_ = ld3_issue(Uc + gcol + n_in, cs, img3);
        }
        f32x4 k_ = zero4, ks_ = zero4;
        if (with_k1) { k_ = ld4_issue_w(Kc + gcol + r0, cu, img3, wide); ks_ = ld3_issue(Kc + gcol + n_in, cs, img3); }
        epsr = ld4_mask(e_, ce);
        if (zown) { *(f32x4*)rkw = ld4_mask(u_, cu); *(f32x4*)(rkw + 32) = ld4_mask(k_, cu); }
        if (sown) { sc_set(0, ld4_mask(s_, cs)); sc_set(1, ld4_mask(ks_, cs)); }
    };
    auto store_rows = [&](float* dst, const f32x4& z4, const f32x4& s4) {     // one D-row column of this lane's sample
        if (live && zown) { if (nv >= 4) st4_wide(dst + gcol + r0, z4); else st4(dst + gcol + r0, z4, nv); }
        if (live && sown) { float* o = dst + gcol + n_in; o[0] = s4.x; o[1] = s4.y; o[2] = s4.z; }
    };
    bool alive = true;
    {
        // ---- k1 = f(u0); with the automatic initial dt (Hairer; the two single evaluations of the streamed driver) also
        // its norms, f(u0 + h0 f0) and that norm ----
        float e = 0.f, b = 0.f;
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            if (multi) { load_tile(tile, false, sv.xs != nullptr); }
            if (zown) {
                s3b_store4(x0w, s3v::NP, *(const f32x4*)rkw);
#pragma unroll
                for (int j = 0; j < 6; ++j) *(f32x4*)(kzw + 32 * j) = zero4;
            }
            s3_bar();
            nstg = 1; evals();
            f32x4 f0z = zero4, f0s = zero4;
            if (zown) f0z = *(const f32x4*)kzw;
            if (live && zown) {
                const f32x4 u4 = *(const f32x4*)rkw;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < nv) { add_norm(e, u4[c], u4[c]); add_norm(b, u4[c], f0z[c]); }
            }
            if (zown) *(f32x4*)(rkw + 32) = f0z;                           // k1 = f(u0)
            if (sown) f0s = read_scalars();
            if (live && sown) {
                const f32x4 u4 = sc_get(0);
#pragma unroll
                for (int c = 0; c < 3; ++c) { add_norm(e, u4[c], u4[c]); add_norm(b, u4[c], f0s[c]); }
            }
            if (sown) sc_set(1, f0s);
            if (multi) {                                   // u0 (when it came from the data columns) and k1 to the buffers
                if (sv.xs) store_rows(a.U[0], zown ? *(const f32x4*)rkw : zero4, sown ? sc_get(0) : zero4);
                store_rows(a.K1[0], f0z, f0s);
                s3_bar();                                  // (this tile's LDS rows are read before the next tile's are written)
            }
        }
        if (sv.hairer) alive = meet(e, b);
        if (sv.hairer && alive) {
            if (tid == 0) { ctrl_phase(ns, 0, msc[32], msc[33], a.n_total); post_ctrl(0); }
            share();
            e = 0.f; b = 0.f;
            for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
                if (multi) { load_tile(tile, true, false); }
                if (zown) {
                    s3b_store4(x0w, s3v::NP, *(const f32x4*)rkw + hstep * *(const f32x4*)(rkw + 32));   // f(u0 + h0 f0)
#pragma unroll
                    for (int j = 0; j < 6; ++j) *(f32x4*)(kzw + 32 * j) = zero4;
                }
                s3_bar();
                nstg = 1; evals();
                if (live && zown) {
                    const f32x4 u4 = *(const f32x4*)rkw, f0 = *(const f32x4*)(rkw + 32), f1 = *(const f32x4*)kzw;
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (c < nv) add_norm(e, u4[c], f1[c] - f0[c]);
                }
                if (live && sown) {
                    const f32x4 u4 = sc_get(0), f0 = sc_get(1), f1 = read_scalars();
#pragma unroll
                    for (int c = 0; c < 3; ++c) add_norm(e, u4[c], f1[c] - f0[c]);
                }
                if (multi) s3_bar();
            }
            alive = meet(e, b);
            if (alive) {
                if (tid == 0) { ctrl_phase(ns, 1, msc[32], msc[33], a.n_total); post_ctrl(0); }
                share();
            }
        }
    }
    // ---- step attempts ----
    int done = 0;
    int nacc = 0;                                          // accepted steps so far (the same count in every workgroup)
    // Static priority for the waves that carry the narrow forward product on top of their wide tiles (0-3): the two waves
    // of a SIMD share its vector issue by priority, then age; measured on one box 33.0-33.3 -> 32.6-32.9 us per attempt
    // (the other half prioritised instead: 33.6-34.1).
    if (wave < 4) __builtin_amdgcn_s_setprio(1);
    for (int it = 0; alive && !done && it < sv.maxiters; ++it) {
      float errsum = 0.f, badcnt = 0.f;
      for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        if (multi) { load_tile(tile, true, false); }
        if (zown) {
            const f32x4 u2 = *(const f32x4*)rkw + (hstep * TS_A21) * *(const f32x4*)(rkw + 32);
            s3b_store4(x0w, s3v::NP, u2);                                    // U_2 = u + h a21 k1
#pragma unroll
            for (int j = 0; j < 6; ++j) *(f32x4*)(kzw + 32 * j) = zero4;
            if (RECORD) {
                dmpw = (live && nacc < a.dump_cap) ? a.dump + (size_t)nacc * a.dump_step_stride + gcol + r0 : nullptr;
                if (dmpw && nv > 0) {
                    const f32x4 un_ = *(const f32x4*)rkw;
                    if (nv >= 4) { st4_wide(dmpw - dmp_stride, un_); st4_wide(dmpw, u2); }
                    else { st4(dmpw - dmp_stride, un_, nv); st4(dmpw, u2, nv); }
                }
            }
        }
        if (RECORD && blockIdx.x == 0 && tid == 0 && nacc < a.dump_cap) a.hs_out[nacc] = hstep;
        s3_bar();
        nstg = 6; evals();
        if (live && zown) {
            const f32x4 k7z = *(const f32x4*)(kzw + 32 * 5), uz_ = *(const f32x4*)rkw;
            const f32x4 un = s3b_load4(x0w, s3v::NP);
            f32x4 ez = TS_BT1 * *(const f32x4*)(rkw + 32) + TS_BT7 * k7z;
            ez += TS_BT2 * *(const f32x4*)(kzw) + TS_BT3 * *(const f32x4*)(kzw + 32) + TS_BT4 * *(const f32x4*)(kzw + 64) +
                  TS_BT5 * *(const f32x4*)(kzw + 96) + TS_BT6 * *(const f32x4*)(kzw + 128);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float scl = fmaf(fmaxf(fabsf(uz_[c]), fabsf(un[c])), reltol, abstol);
                const float x = c < nv ? hstep * ez[c] / scl : 0.f;
                errsum = fmaf(x, x, errsum);
                badcnt += (c < nv && !(fabsf(un[c]) <= 3.0e38f)) ? 1.f : 0.f;
            }
        }
        f32x4 uns = zero4, k7s = zero4;
        if (sown) {
            f32x4 ks[7];
#pragma unroll
            for (int j = 0; j < 6; ++j) ks[j] = sc_get(1 + j);
            const f32x4 us = sc_get(0);
            ks[6] = read_scalars();
            k7s = ks[6];
            uns = us + hstep * stage_acc4<6>(ks);
            if (live) err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, 3);
            sc_set(7, uns);                                                // kept for an accepted attempt
        }
        if (multi) {                                       // (u_new, k7) to the other buffer set, as a step launch does
            float* Un = cur ? a.U[0] : a.U[1];
            float* Kn = cur ? a.K1[0] : a.K1[1];
            store_rows(Un, zown ? s3b_load4(x0w, s3v::NP) : zero4, uns);
            store_rows(Kn, zown ? *(const f32x4*)(kzw + 32 * 5) : zero4, k7s);
            s3_bar();
        }
      }
        alive = meet(errsum, badcnt);
        if (!alive) break;
        if (tid == 0) {
            const int acc0 = ns->naccept;
            const float t_att = ns->t, h_att = ns->h;
            ctrl_after_step(ns, msc[32], msc[33], a.n_total);
            post_ctrl(ns->naccept != acc0);
            if (sv.trace && blockIdx.x == 0 && it < sv.trace_cap)
                *(f32x4u*)(sv.trace + 4 * it) = f32x4{t_att, h_att, ns->eest, ns->naccept != acc0 ? 1.f : 0.f};
        }
        const int fl = share();
        done = fl & 1;
        if (fl & 2 && multi) { ++nacc; cur ^= 1; }
        else if (fl & 2) {                                                 // accepted: u <- u_new, k1 <- k7 (FSAL)
            ++nacc;
            if (zown) {
                *(f32x4*)rkw = s3b_load4(x0w, s3v::NP);
                *(f32x4*)(rkw + 32) = *(const f32x4*)(kzw + 32 * 5);
            }
            if (sown) { const f32x4 k7s = read_scalars(); sc_set(0, sc_get(7)); sc_set(1, k7s); }
        }
    }
    // ---- the final state to the integrator's buffer set 0 ----
    // (to the caller's columns when the launcher passed them: sv.u_out; never after an abort -- the caller may be solving
    // in place, and the streamed driver starts again from u0)
    if (!multi && (alive || !sv.u_out)) store_rows(sv.u_out ? sv.u_out : a.U[0], zown ? *(const f32x4*)rkw : zero4, sown ? sc_get(0) : zero4);
    float v4[4] = {0.f, 0.f, 0.f, 0.f};                   // this lane's share of the loss sums (waves 4 and 6)
    if (sv.logpx && alive) {
        // ---- post-processing of every tile: logp(z) - dlogp, the regulariser rows; then the loss sums of the batch ----
      for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        if (multi) {
            set_tile(tile);
            const float* Uc = cur ? a.U[1] : a.U[0];
            const f32x4 u_ = ld4_issue_w(Uc + gcol + r0, cu, img3, wide), s_ = ld3_issue(Uc + gcol + n_in, cs, img3);
            s3_bar();                                      // (the tile before has read its RED / SC words)
            if (zown) *(f32x4*)rkw = ld4_mask(u_, cu);
            if (sown) sc_set(0, ld4_mask(s_, cs));
        }
        if (zown) {
            const f32x4 u4 = *(const f32x4*)rkw;
            float ss = 0.f, sa = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nv) { ss = fmaf(u4[c], u4[c], ss); if (r0 + c >= sv.nvars) sa = fmaf(u4[c], u4[c], sa); }
            redw[0] = ss; redw[32 * 8] = sa;
        }
        s3_bar();
        if (live && sown) {
            const float ss = red8(0), sa = red8(1);
            const f32x4 us = sc_get(0);
            const float log2pi = 1.8378770664093453f;
            const float lp = -0.5f * fmaf((float)n_in, log2pi, ss) - us.x;        // base_icnf.jl:177-178
            const float aa = (sv.norm_z_aug && sv.naugs > 0) ? sqrtf(sa) : 0.f;   // :179-187
            const size_t b = (size_t)(b0 + s), Bz = (size_t)a.B;
            sv.logpx[b] = lp; sv.regs[b] = us.y; sv.regs[Bz + b] = us.z; sv.regs[2 * Bz + b] = aa;
            v4[0] += lp; v4[1] += us.y; v4[2] += us.z; v4[3] += aa;
        }
      }
        if (sv.sums5) {
            // workgroup partials (waves 4 and 6 hold them) -> tagged words, one more meeting index; workgroup 0 adds them in
            // workgroup order
            unsigned long long* qb = reinterpret_cast<unsigned long long*>(sv.part) + 2048;
            const unsigned tag = mbase + (unsigned)nsync + 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = s3_wave_sum(v4[j]);
            s3_bar();                                      // (red8 above read RED; msc below)
            if (lane == 0 && (wave == 4 || wave == 6)) for (int j = 0; j < 4; ++j) msc[(wave == 4 ? 0 : 8) + j] = v4[j];
            s3_bar();
            if (tid < 4)
                __hip_atomic_store(qb + 4 * blockIdx.x + tid, ((unsigned long long)tag << 32) | __float_as_uint(msc[tid] + msc[8 + tid]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) {
                float c4[4] = {0.f, 0.f, 0.f, 0.f};
                float late = 0.f;                              // a partial that never arrived: the abort path, as in a meeting
                if (tid < (int)gridDim.x) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int got = 0;
                        const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
                        for (int spin = 0; spin < sv.spin_limit; ++spin) {
                            const unsigned long long w = __hip_atomic_load(qb + 4 * tid + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(w >> 32) == tag) { c4[j] = __uint_as_float((unsigned)w); got = 1; break; }
                            if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (!got) late = 1.f;
                    }
                    if (late != 0.f) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[j] = s3_wave_sum(c4[j]);
                late = s3_wave_sum(late);
                s3_bar();
                if (lane == 0) { for (int j = 0; j < 4; ++j) msc[4 * wave + j] = c4[j]; msc[32 + wave] = late; }
                s3_bar();
                if (tid < 4) {
                    float r = 0.f;
                    for (int w = 0; w < 8; ++w) r += msc[4 * w + tid];
                    sv.sums5[tid] = r;
                }
                if (tid == 0) {
                    sv.sums5[4] = (float)a.B;
                    float nl = 0.f;
                    for (int w = 0; w < 8; ++w) nl += msc[32 + w];
                    if (nl != 0.f) { ns->done = 0; ns->n_partials = -1; }               // the host sees the abort word and runs the solve again, streamed
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        // A wait that ran out anywhere (this workgroup's `alive`, or another's abort word -- set before this workgroup
        // could have passed the meeting in question) makes the launch void: the published state says so by itself
        // (n_partials < 0), so that the host can tell for THIS launch even when others are queued behind it.
        if (!alive || __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ns->done = 0; ns->n_partials = -1; }
        __hip_atomic_store(sv.base_dev, mbase + (unsigned)nsync + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (workgroups differ by <= 1 meeting; + the sums)
        ns->cur = multi ? cur : 0;
        *a.st_out = *ns;
        if (sv.t_out) { sv.t_out[1] += __builtin_amdgcn_s_memrealtime() - sv.t_out[0]; sv.t_out[2] += 1; }
        publish_mirror(a, *ns);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_solve3jb -- the one-launch solve (k_solve3b) around k_step3jb's evaluation code: the JVP compute mode of the headline
// shape (and VJP handles without the |eps^T J| row: FFJORD).  Same meetings, same controller, same fused assembly of u0 /
// post-processing / loss sums.
//   One tile per workgroup (B <= 32 x the resident workgroups): the Runge-Kutta rows of z stay in registers for the whole
//   solve (as in k_step3jb), the scalar rows in LDS; nothing but the meeting words leaves the CU between attempts.
//   SEVERAL tiles per workgroup (larger batches: BASELINE config 4 unsharded, lock-step-free shards of any size): every
//   attempt loops over the workgroup's tiles, taking (u, k1) and the probe rows from the integrator's buffer set `cur`
//   and filing (u_new, k7) in the other set -- what a step launch of k_step3jb does per tile --, the workgroups meet
//   once per attempt over the sum of their tiles' error partials, an accepted attempt flips `cur`.  Per attempt and
//   sample that is 2 D + n_in floats read and 2 D written (45 MB at B = 65 536: microseconds), against the weight stream,
//   prologue and launch of a step kernel per attempt that it replaces.
// ---------------------------------------------------------------------------------------------------------------
// RECORD (gradient path of JVP-mode handles, as k_solve3b<RECORD>): every attempt files u_n and U_2..U_6 (z rows) in the trajectory
// slot of step `naccept`, its signed step size in a.hs_out; one tile per workgroup, no conditioning.
template <bool MULTI, bool COND, bool RECORD = false>
__global__ void __launch_bounds__(512, 2) k_solve3jb(MfmaArgs a, const char* __restrict__ imgb, int n_in, int norm_z,
                                                     int norm_j, const S3Tab tab, Solve3Args sv) {
    static_assert(!RECORD || (!MULTI && !COND), "the recording form: one tile per workgroup, no conditioning");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const float* img3 = reinterpret_cast<const float*>(imgb);      // (a valid address for masked loads)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const int t = wave & 1, hf = (wave >> 1) & 1;
    const bool zown = wave < 4;
    const bool sown = !zown && t == 0 && q == 0;
    const int smp = 16 * hf + s;
    const int r0 = 16 * t + 4 * q;
    const int nv = n_in - r0;
    const bool wide = (n_in & 3) == 0;
    const int ntile = (a.B + 31) / 32;
    constexpr bool multi = MULTI;                          // several tiles per workgroup: the state lives in a.U / a.K1
    // the tile this workgroup is working on
    int b0 = blockIdx.x * 32 + 16 * hf;
    bool live = s < max(0, min(16, a.B - b0));
    size_t gcol = (size_t)(b0 + s) * D;
    int ce = live ? nv : 0, cu = (zown && live) ? nv : 0, cs = (sown && live) ? 3 : 0;
    auto set_tile = [&](int tile) {
        b0 = tile * 32 + 16 * hf;
        live = s < max(0, min(16, a.B - b0));
        gcol = (size_t)(b0 + s) * D;
        ce = live ? nv : 0; cu = (zown && live) ? nv : 0; cs = (sown && live) ? 3 : 0;
    };
    if (sv.t_out && blockIdx.x == 0 && tid == 0) sv.t_out[0] = __builtin_amdgcn_s_memrealtime();
    const f32x4 re = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
    f32x4 ru, rs;
    if (sv.xs) {                                           // u0 = (xs; zeros for the augmented and the scalar rows)
        const float* xc = sv.xs + (size_t)(b0 + s) * sv.nvars;
#pragma unroll
        for (int j = 0; j < 4; ++j) ru[j] = (cu > j && r0 + j < sv.nvars) ? xc[r0 + j] : 0.f;
        rs = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        ru = ld4_issue_w(a.U[0] + gcol + r0, cu, img3, wide);
        rs = ld3_issue(a.U[0] + gcol + n_in, cs, img3);
    }
    // (they arrive in fp32 and are split here, in arrival order, while the rest of the stream is in flight)
    S3bOp wF1, wF2[4], w3[4];
    {
        const char* fw = imgb + s3g::F32 + (size_t)wave * 10 * 2048 + 16 * lane;
        const char* f3 = imgb + s3g::F32 + (size_t)(80 + 4 * t) * 2048 + 16 * lane;
        auto src = [&](int f) { return f < 5 ? fw + f * 2048 : f3 + (f - 5) * 2048; };
        constexpr int AH = 5;                                  // fragments requested ahead of the one being split
        f32x4 raw[9][2];
#pragma unroll
        for (int f = 0; f < AH; ++f) { raw[f][0] = *(const f32x4*)src(f); raw[f][1] = *(const f32x4*)(src(f) + 1024); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            if (f + AH < 9) { raw[f + AH][0] = *(const f32x4*)src(f + AH); raw[f + AH][1] = *(const f32x4*)(src(f + AH) + 1024); }
            S3bOp o = s3b_split8(raw[f][0], raw[f][1]);
            s3b_pin(o);                                        // (the split stays here, between the two scheduling barriers)
            if (f == 0) wF1 = o; else if (f < 5) wF2[f - 1] = o; else w3[f - 5] = o;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    constexpr int NCB = (2 * 128 + 32) / 4;
    const f32x4 sgb = reinterpret_cast<const f32x4*>(imgb + s3g::BIASB)[min(tid, NCB - 1)];
    float* sc = lds + s3b::SC + smp * 24;
    auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
    auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
    __builtin_amdgcn_sched_barrier(0);
    // the probe rows: fp32 for the trace row, split as tau_0 (the tangent operand of the first layer)
    float* epw = lds + s3b::EPS + smp * 40 + r0;
    const int nsw = (-(s >> 2)) & 3;                                 // chunk swizzle of the K = 32 images (rows s, 16 + s)
    const int nw = smp * s3b::NS + 16 * ((2 * t + (q >> 1)) ^ nsw) + 8 * (q & 1);      // this lane's 4 rows there
    char* x0w = ldsb + s3b::X0B + nw;
    auto put_eps = [&](const f32x4& ev) {
        *(f32x4*)epw = ev;
        s3b_store4(ldsb + s3b::T0B + nw, s3b::NP, ev);
    };
    put_eps(ld4_mask(re, ce));
    float* msc = lds + s3b::MISC;
    StepState* ns = reinterpret_cast<StepState*>(msc + 44);            // the integrator state (thread 0 runs the controller on it)
    if (tid == 0) *ns = sv.init;
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + s3b::BIAS)[tid] = sgb;
    const int single = 0;
    (void)single;
    auto red8 = [&](int kind) {
        const float* r = lds + s3b::RED + (kind * 32 + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    auto read_scalars = [&]() {
        const float e2 = red8(0), ld = red8(1), n2 = red8(2);
        return f32x4{ld, norm_z ? __builtin_sqrtf(e2) : 0.f, norm_j ? __builtin_sqrtf(n2) : 0.f, 0.f};
    };
    float* redw = lds + s3b::RED + smp * 8 + 4 * t + q;
    float* g3w = lds + s3b::G3 + smp * 40 + r0;
    const float* bias = lds + s3b::BIAS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // B operands: lane (sample s of half A, k = 8q ..): byte offsets into an image; half B = 16 rows on
    // (swizzled rows: k-block kb of a row is reached by XOR 64 kb on the byte offset)
    const int nb_rd = s * s3b::NS + 16 * (q ^ nsw), wb_rd = s * s3b::WS + 16 * (q ^ s);
    // results: lane (sample s, rows 16 wave + 4q ..) of the wide images
    const int wb_wr = s * s3b::WS + 16 * ((2 * wave + (q >> 1)) ^ s) + 8 * (q & 1);
    constexpr int HBW = 16 * s3b::WS, HBN = 16 * s3b::NS;

    f32x4 uz = ld4_mask(ru, cu), kz[7], un = zero4;        // Runge-Kutta rows of z (waves 0-3): u, k1..k7, u_new
#pragma unroll
    for (int j = 0; j < 7; ++j) kz[j] = zero4;
    if (sown) { sc_set(0, ld4_mask(rs, cs)); sc_set(1, zero4); }
    s3_bar();                                              // biases, probe images, state
    float hstep = ns->h, abstol = ns->abstol, reltol = ns->reltol;
    hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hstep)));
    abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(abstol)));
    reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(reltol)));
    int nsync = 0;                                         // meetings so far (the same count in every workgroup)
    // meetings held on this buffer by earlier launches: a device word, so that a launch can be queued behind another
    // before the host knows how many meetings that one will hold (read once; workgroup 0 advances it at the very end)
    const unsigned mbase = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sv.base_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    // The workgroups' partials (e, b) -> the sums of all of them, in msc[32], msc[33] for thread 0 (what the prologue of a
    // step launch computes from the previous launch's partials, in the same order).  Returns false when a wait ran out.
    auto meet = [&](float e_lane, float b_lane) -> bool {
        float e = s3_wave_sum(e_lane), b = s3_wave_sum(b_lane);
        if (lane == 0) { msc[wave] = e; msc[16 + wave] = b; }
        s3_bar();
        // Every partial travels with the index of the meeting it belongs to in the same 8-byte word: a reader needs no
        // ticket -- thread i polls workgroup i's two words until both carry this meeting's index -- so a meeting costs one
        // store and one load round trip.  Two buffers by parity: a workgroup can be one meeting ahead of a reader, not two.
        unsigned long long* pb = reinterpret_cast<unsigned long long*>(sv.part) + (nsync & 1) * 1024;
        const unsigned tag = mbase + (unsigned)nsync + 1u;
        if (tid == 0) {
            float e8 = 0.f, b8 = 0.f;
            for (int w = 0; w < 8; ++w) { e8 += msc[w]; b8 += msc[16 + w]; }
            __hip_atomic_store(pb + 2 * blockIdx.x, ((unsigned long long)tag << 32) | __float_as_uint(e8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pb + 2 * blockIdx.x + 1, ((unsigned long long)tag << 32) | __float_as_uint(b8), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        float cp0 = 0.f, cp1 = 0.f;
        int ok = 1;
        if (tid < (int)gridDim.x) {
            ok = 0;
            const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
            for (int spin = 0; spin < sv.spin_limit; ++spin) {
                // both words of workgroup `tid` with ONE 16-byte load past the caches (each half carries its own index, so a
                // torn pair is simply not accepted): half the polling traffic of two 8-byte atomic loads
                u32x4 wq;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(wq) : "v"(pb + 2 * tid) : "memory");
                const unsigned long long w0 = ((unsigned long long)wq.y << 32) | wq.x, w1 = ((unsigned long long)wq.w << 32) | wq.z;
                if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag) {
                    cp0 = __uint_as_float((unsigned)w0); cp1 = __uint_as_float((unsigned)w1); ok = 1;
                    break;
                }
                if ((spin & 255) == 255 && __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;   // bounded in TIME
                __builtin_amdgcn_s_sleep(4);
            }
            if (!ok) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cp0 = s3_wave_sum(cp0); cp1 = s3_wave_sum(cp1);
        const float bad = s3_wave_sum(ok ? 0.f : 1.f);
        s3_bar();                                          // (msc[0..7], [16..23] were read by thread 0 above)
        if (lane == 0) { msc[wave] = cp0; msc[16 + wave] = cp1; msc[24 + wave] = bad; }
        s3_bar();
        float nbad = 0.f;
        for (int w = 0; w < 8; ++w) nbad += msc[24 + w];
        if (tid == 0) {
            float p0 = 0.f, p1 = 0.f;
            for (int w = 0; w < 8; ++w) { p0 += msc[w]; p1 += msc[16 + w]; }
            msc[32] = p0; msc[33] = p1;
        }
        ++nsync;
        return nbad == 0.f;
    };
    // thread 0 ran a controller phase on *ns: the new step and tolerances to everyone
    auto share = [&]() {
        s3_bar();
        hstep = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[36])));
        abstol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[37])));
        reltol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(msc[38])));
        const int fl = __builtin_amdgcn_readfirstlane(__float_as_int(msc[39]));
        s3_bar();                                          // (the words are rewritten by the next phase)
        return fl;                                         // bit 0: done, bit 1: the attempt was accepted
    };
    auto post_ctrl = [&](int accepted) {                   // thread 0, after a controller phase
        msc[36] = ns->h; msc[37] = ns->abstol; msc[38] = ns->reltol;
        msc[39] = __int_as_float((ns->done ? 1 : 0) | (accepted ? 2 : 0));
    };
    // the evaluations of one attempt (nstg = 6) or one evaluation (nstg = 1) at the state image in place
    int nstg = 1;
    f32x4 tacc = zero4;
    auto finish_tau = [&]() {
        const f32x4 tj = tacc * *(const f32x4*)g3w;
        redw[32 * 8] = -s3_dot4(tj, *(const f32x4*)epw);
        redw[2 * 32 * 8] = s3_dot4(tj, tj);
    };
    float* dmpw = nullptr;                                 // RECORD: this lane's rows of the current step's slot (null: not filed)
    const size_t dmp_stride = a.dump_stride;
    auto evals = [&]() {
        constexpr bool S3B_COND = COND;
        constexpr bool S3JB_RECORDS = RECORD;
#include "cnf_step3jb_eval.inc"
    };
    auto add_norm = [&](float& acc, float u, float x) {
        const float sk = fmaf(fabsf(u), reltol, abstol);
        const float y = x / sk;
        acc = fmaf(y, y, acc);
    };
    // ---- several tiles per workgroup: a tile's rows from / to the integrator's buffers ----
    int cur = 0;                                           // the buffer set that holds (u, k1)
    // (u, k1) and the probe rows of tile `tile`; k1 only when `with_k1`; from the data columns when `from_xs`
    auto load_tile = [&](int tile, bool with_k1, bool from_xs) {
        set_tile(tile);
        const float* Uc = cur ? a.U[1] : a.U[0];
        const float* Kc = cur ? a.K1[1] : a.K1[0];
        const f32x4 e_ = ld4_issue_w(a.eps + (size_t)(b0 + s) * n_in + r0, ce, img3, wide);
        f32x4 u_, s_;
        if (from_xs) {
            const float* xc = sv.xs + (size_t)(b0 + s) * sv.nvars;
#pragma unroll
            for (int j = 0; j < 4; ++j) u_[j] = (cu > j && r0 + j < sv.nvars) ? xc[r0 + j] : 0.f;
            s_ = zero4;
        } else {
            u_ = ld4_issue_w(Uc + gcol + r0, cu, img3, wide);
            s_ = ld3_issue(Uc + gcol + n_in, cs, img3);
        }
        f32x4 k_ = zero4, ks_ = zero4;
        if (with_k1) { k_ = ld4_issue_w(Kc + gcol + r0, cu, img3, wide); ks_ = ld3_issue(Kc + gcol + n_in, cs, img3); }
        put_eps(ld4_mask(e_, ce));
        uz = ld4_mask(u_, cu);
        kz[0] = ld4_mask(k_, cu);
        if (sown) { sc_set(0, ld4_mask(s_, cs)); sc_set(1, ld4_mask(ks_, cs)); }
    };
    auto store_rows = [&](float* dst, const f32x4& z4, const f32x4& s4) {     // one D-row column of this lane's sample
        if (live && zown) { if (nv >= 4) st4_wide(dst + gcol + r0, z4); else st4(dst + gcol + r0, z4, nv); }
        if (live && sown) { float* o = dst + gcol + n_in; o[0] = s4.x; o[1] = s4.y; o[2] = s4.z; }
    };
    bool alive = true;
    {
        // ---- k1 = f(u0); with the automatic initial dt also its norms, f(u0 + h0 f0) and that norm ----
        float e = 0.f, b = 0.f;
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            if (multi) { load_tile(tile, false, sv.xs != nullptr); s3_bar(); }
            if (zown) { un = uz; s3b_store4(x0w, s3b::NP, un); }
            s3_bar();
            nstg = 1; evals();
            if (live && zown) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < nv) { add_norm(e, uz[c], uz[c]); add_norm(b, uz[c], kz[1][c]); }
            }
            if (zown) kz[0] = kz[1];                                           // k1 = f(u0)
            f32x4 f0s = zero4;
            if (sown) f0s = read_scalars();
            if (live && sown) {
                const f32x4 u4 = sc_get(0);
#pragma unroll
                for (int c = 0; c < 3; ++c) { add_norm(e, u4[c], u4[c]); add_norm(b, u4[c], f0s[c]); }
            }
            if (sown) sc_set(1, f0s);
            if (multi) {                                   // u0 (when it came from the data columns) and k1 to the buffers
                if (sv.xs) store_rows(a.U[0], uz, sc_get(0));
                store_rows(a.K1[0], kz[0], f0s);
                s3_bar();                                  // (this tile's LDS words are read before the next tile's are written)
            }
        }
        if (sv.hairer) alive = meet(e, b);
        if (sv.hairer && alive) {
            if (tid == 0) { ctrl_phase(ns, 0, msc[32], msc[33], a.n_total); post_ctrl(0); }
            share();
            e = 0.f; b = 0.f;
            for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
                if (multi) { load_tile(tile, true, false); s3_bar(); }
                if (zown) { un = uz + hstep * kz[0]; s3b_store4(x0w, s3b::NP, un); }
                s3_bar();
                nstg = 1; evals();
                if (live && zown) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (c < nv) add_norm(e, uz[c], kz[1][c] - kz[0][c]);
                }
                if (live && sown) {
                    const f32x4 u4 = sc_get(0), f0 = sc_get(1), f1 = read_scalars();
#pragma unroll
                    for (int c = 0; c < 3; ++c) add_norm(e, u4[c], f1[c] - f0[c]);
                }
                if (multi) s3_bar();
            }
            alive = meet(e, b);
            if (alive) {
                if (tid == 0) { ctrl_phase(ns, 1, msc[32], msc[33], a.n_total); post_ctrl(0); }
                share();
            }
        }
    }
    // ---- step attempts ----
    int done = 0;
    int nacc = 0;                                          // accepted steps so far (the same count in every workgroup)
    for (int it = 0; alive && !done && it < sv.maxiters; ++it) {
        float errsum = 0.f, badcnt = 0.f;
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            if (multi) { load_tile(tile, true, false); s3_bar(); }
#pragma unroll
            for (int j = 1; j < 7; ++j) kz[j] = zero4;
            if (zown) { un = uz + (hstep * TS_A21) * kz[0]; s3b_store4(x0w, s3b::NP, un); }      // U_2 = u + h a21 k1
            if (RECORD) {
                if (zown) {
                    dmpw = (live && nacc < a.dump_cap) ? a.dump + (size_t)nacc * a.dump_step_stride + gcol + r0 : nullptr;
                    if (dmpw && nv > 0) {
                        if (nv >= 4) { st4_wide(dmpw - dmp_stride, uz); st4_wide(dmpw, un); }
                        else { st4(dmpw - dmp_stride, uz, nv); st4(dmpw, un, nv); }
                    }
                }
                if (blockIdx.x == 0 && tid == 0 && nacc < a.dump_cap) a.hs_out[nacc] = hstep;
            }
            s3_bar();
            nstg = 6; evals();
            if (live && zown) {
                const f32x4 ez = TS_BT1 * kz[0] + TS_BT2 * kz[1] + TS_BT3 * kz[2] + TS_BT4 * kz[3] + TS_BT5 * kz[4] +
                                 TS_BT6 * kz[5] + TS_BT7 * kz[6];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float scl = fmaf(fmaxf(fabsf(uz[c]), fabsf(un[c])), reltol, abstol);
                    const float x = c < nv ? hstep * ez[c] / scl : 0.f;
                    errsum = fmaf(x, x, errsum);
                    badcnt += (c < nv && !(fabsf(un[c]) <= 3.0e38f)) ? 1.f : 0.f;
                }
            }
            f32x4 uns = zero4, k7s = zero4;
            if (sown) {
                f32x4 ks[7];
#pragma unroll
                for (int j = 0; j < 6; ++j) ks[j] = sc_get(1 + j);
                const f32x4 us = sc_get(0);
                ks[6] = read_scalars();
                k7s = ks[6];
                uns = us + hstep * stage_acc4<6>(ks);
                if (live) err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, 3);
                sc_set(7, uns);                                                // kept for an accepted attempt
            }
            if (multi) {                                   // (u_new, k7) to the other buffer set, as a step launch does
                float* Un = cur ? a.U[0] : a.U[1];
                float* Kn = cur ? a.K1[0] : a.K1[1];
                store_rows(Un, un, uns);
                store_rows(Kn, kz[6], k7s);
                s3_bar();
            }
        }
        alive = meet(errsum, badcnt);
        if (!alive) break;
        if (tid == 0) {
            const int acc0 = ns->naccept;
            const float t_att = ns->t, h_att = ns->h;
            ctrl_after_step(ns, msc[32], msc[33], a.n_total);
            post_ctrl(ns->naccept != acc0);
            if (sv.trace && blockIdx.x == 0 && it < sv.trace_cap)
                *(f32x4u*)(sv.trace + 4 * it) = f32x4{t_att, h_att, ns->eest, ns->naccept != acc0 ? 1.f : 0.f};
        }
        const int fl = share();
        done = fl & 1;
        if (fl & 2) {                                                      // accepted: u <- u_new, k1 <- k7 (FSAL)
            ++nacc;
            if (multi) cur ^= 1;
            else {
                if (zown) { uz = un; kz[0] = kz[6]; }
                if (sown) { const f32x4 k7s = read_scalars(); sc_set(0, sc_get(7)); sc_set(1, k7s); }
            }
        }
    }
    // ---- the final state: one tile per workgroup -> the integrator's buffer set 0; several: it is in set `cur` ----
    if (!multi && (alive || !sv.u_out)) store_rows(sv.u_out ? sv.u_out : a.U[0], uz, sc_get(0));
    float v4[4] = {0.f, 0.f, 0.f, 0.f};                   // this lane's share of the loss sums (waves 4 and 6)
    if (sv.logpx && alive) {
        // ---- post-processing of every tile: logp(z) - dlogp, the regulariser rows; then the loss sums of the batch ----
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            if (multi) {
                set_tile(tile);
                const float* Uc = cur ? a.U[1] : a.U[0];
                const f32x4 u_ = ld4_issue_w(Uc + gcol + r0, cu, img3, wide), s_ = ld3_issue(Uc + gcol + n_in, cs, img3);
                uz = ld4_mask(u_, cu);
                s3_bar();                                  // (the tile before has read its RED / SC words)
                if (sown) sc_set(0, ld4_mask(s_, cs));
            }
            if (zown) {
                const f32x4 u4 = uz;
                float ss = 0.f, sa = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nv) { ss = fmaf(u4[c], u4[c], ss); if (r0 + c >= sv.nvars) sa = fmaf(u4[c], u4[c], sa); }
                redw[0] = ss; redw[32 * 8] = sa;
            }
            s3_bar();
            if (live && sown) {
                const float ss = red8(0), sa = red8(1);
                const f32x4 us = sc_get(0);
                const float log2pi = 1.8378770664093453f;
                const float lp = -0.5f * fmaf((float)n_in, log2pi, ss) - us.x;        // base_icnf.jl:177-178
                const float aa = (sv.norm_z_aug && sv.naugs > 0) ? sqrtf(sa) : 0.f;   // :179-187
                const size_t b = (size_t)(b0 + s), Bz = (size_t)a.B;
                sv.logpx[b] = lp; sv.regs[b] = us.y; sv.regs[Bz + b] = us.z; sv.regs[2 * Bz + b] = aa;
                v4[0] += lp; v4[1] += us.y; v4[2] += us.z; v4[3] += aa;
            }
        }
        if (sv.sums5) {
            // workgroup partials (waves 4 and 6 hold them) -> tagged words, one more meeting index; workgroup 0 adds them in
            // workgroup order
            unsigned long long* qb = reinterpret_cast<unsigned long long*>(sv.part) + 2048;
            const unsigned tag = mbase + (unsigned)nsync + 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = s3_wave_sum(v4[j]);
            s3_bar();                                      // (red8 above read RED; msc below)
            if (lane == 0 && (wave == 4 || wave == 6)) for (int j = 0; j < 4; ++j) msc[(wave == 4 ? 0 : 8) + j] = v4[j];
            s3_bar();
            if (tid < 4)
                __hip_atomic_store(qb + 4 * blockIdx.x + tid, ((unsigned long long)tag << 32) | __float_as_uint(msc[tid] + msc[8 + tid]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) {
                float c4[4] = {0.f, 0.f, 0.f, 0.f};
                float late = 0.f;                              // a partial that never arrived: the abort path, as in a meeting
                if (tid < (int)gridDim.x) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int got = 0;
                        const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
                        for (int spin = 0; spin < sv.spin_limit; ++spin) {
                            const unsigned long long w = __hip_atomic_load(qb + 4 * tid + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(w >> 32) == tag) { c4[j] = __uint_as_float((unsigned)w); got = 1; break; }
                            if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (!got) late = 1.f;
                    }
                    if (late != 0.f) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[j] = s3_wave_sum(c4[j]);
                late = s3_wave_sum(late);
                s3_bar();
                if (lane == 0) { for (int j = 0; j < 4; ++j) msc[4 * wave + j] = c4[j]; msc[32 + wave] = late; }
                s3_bar();
                if (tid < 4) {
                    float r = 0.f;
                    for (int w = 0; w < 8; ++w) r += msc[4 * w + tid];
                    sv.sums5[tid] = r;
                }
                if (tid == 0) {
                    sv.sums5[4] = (float)a.B;
                    float nl = 0.f;
                    for (int w = 0; w < 8; ++w) nl += msc[32 + w];
                    if (nl != 0.f) { ns->done = 0; ns->n_partials = -1; }               // the host sees the abort word and runs the solve again, streamed
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        // A wait that ran out anywhere (this workgroup's `alive`, or another's abort word -- set before this workgroup
        // could have passed the meeting in question) makes the launch void: the published state says so by itself
        // (n_partials < 0), so that the host can tell for THIS launch even when others are queued behind it.
        if (!alive || __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ns->done = 0; ns->n_partials = -1; }
        __hip_atomic_store(sv.base_dev, mbase + (unsigned)nsync + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (workgroups differ by <= 1 meeting; + the sums)
        ns->cur = multi ? cur : 0;
        *a.st_out = *ns;
        if (sv.t_out) { sv.t_out[1] += __builtin_amdgcn_s_memrealtime() - sv.t_out[0]; sv.t_out[2] += 1; }
        publish_mirror(a, *ns);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Self-test of the arithmetic of the split kernels (cnf_selftest_split_product): C = A Bt^T for A, Bt of 16 x K fp32 on ONE
// wave, with the operand split (s3b_split8) and the six-term product (s3b_mm, smallest term first, fp32 accumulate over the
// k-blocks in order) that k_step3b / k_step3jb / k_solve3b / k_solve3jb run.  The parity suite bounds its error against
// float64 on random, wide-dynamic-range and cancelling inputs.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_split_product_test(const float* __restrict__ A, const float* __restrict__ Bt,
                                                           float* __restrict__ Cm, int K) {
    const int lane = threadIdx.x, x = lane & 15, q = lane >> 4;
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int kb = 0; kb < K / 32; ++kb) {
        const float* ap = A + (size_t)x * K + 32 * kb + 8 * q;
        const float* bp = Bt + (size_t)x * K + 32 * kb + 8 * q;
        const S3bOp a = s3b_split8(*(const f32x4*)ap, *(const f32x4*)(ap + 4));      // (hipMalloc base, offsets of 8 floats)
        S3bOp b[1];
        b[0] = s3b_split8(*(const f32x4*)bp, *(const f32x4*)(bp + 4));
        s3b_mm<1>(acc, a, b);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) Cm[(size_t)(4 * q + i) * 16 + x] = acc[0][i];       // lane (column x, rows 4q..4q+3)
}
hipError_t split_product_test_launch(const float* dA, const float* dBt, float* dC, int K, hipStream_t s) {
    hipLaunchKernelGGL(k_split_product_test, dim3(1), dim3(64), 0, s, dA, dBt, dC, K);
    return hipGetLastError();
}

// Image of k_step3jb / k_step3b (layout: namespace s3g): biases, the fp32 fragments, the two split LDS images of k_step3b.
__global__ void k_pack_step3b(NetDesc nd, const float* __restrict__ P, char* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int NIMG = 2 * 32 * 16;
    // W_l[o][k] with zero padding
    auto W = [&](int l, int o, int k) {
        return (o < nd.dims[l + 1] && k < nd.dims[l]) ? P[nd.w_off[l] + o + (size_t)k * nd.dims[l + 1]] : 0.f;
    };
    if (i < (2 * 128 + 32)) {
        const int l = i < 128 ? 0 : (i < 256 ? 1 : 2), o = i < 128 ? i : (i < 256 ? i - 128 : i - 256);
        reinterpret_cast<float*>(img + s3g::BIASB)[i] = o < nd.dims[l + 1] ? P[nd.b_off[l] + o] : 0.f;
    }
    if (i < s3g::NFR * 64) {
        const int f = i >> 6, lane = i & 63, x = lane & 15, q = lane >> 4;
        int l, tile, kb, tr = 0;                   // tr: the fragment is a tile of the TRANSPOSED matrix
        if (f < 80) {                              // wave w: W1 | W2 x4 | W3^T | W2^T x4
            const int g = f % 10;
            tile = f / 10;
            l = g == 0 ? 0 : (g == 5 ? 2 : 1); tr = g >= 5; kb = (g == 0 || g == 5) ? 0 : (g < 5 ? g - 1 : g - 6);
        } else { l = 2; tile = (f - 80) >> 2; kb = (f - 80) & 3; }      // W3 tiles (k_step3jb)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * tile + x, k = 32 * kb + 8 * q + j;
            v[j] = tr ? W(l, k, r) : W(l, r, k);
        }
        char* d = img + s3g::F32 + (size_t)f * 2048 + 16 * lane;
        *(f32x4*)d = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(d + 1024) = f32x4{v[4], v[5], v[6], v[7]};
    } else if (i < s3g::NFR * 64 + NIMG) {
        const int e = i - s3g::NFR * 64, which = e / (32 * 16), r = (e / 16) % 32, g = e % 16;     // 16 chunks of 8 bf16 per row
        float vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g + j;
            vv[j] = k < 128 ? (which == 0 ? W(2, r, k) : W(0, k, r)) : 0.f;      // W3[r][k] | W1^T[r][k] = W1[k][r]
        }
        const S3bOp sp = s3b_split8(f32x4{vv[0], vv[1], vv[2], vv[3]}, f32x4{vv[4], vv[5], vv[6], vv[7]});
        const bf16x8 h = sp.h, m = sp.m, lo = sp.l;
        char* dst = img + s3g::W3I + (size_t)which * s3v::WI + r * s3v::WS + 16 * (g ^ (r & 15));      // (swizzled rows: s3v)
        *(bf16x8*)dst = h; *(bf16x8*)(dst + s3v::WP) = m; *(bf16x8*)(dst + 2 * s3v::WP) = lo;
    }
}
size_t step3b_img_bytes() { return (size_t)s3g::IMG_BYTES; }
void step3b_pack(const NetDesc& nd, const float* d_params, void* d_imgb, hipStream_t s) {
    constexpr int NT = s3g::NFR * 64 + 2 * 32 * 16;
    hipLaunchKernelGGL(k_pack_step3b, dim3((NT + 255) / 256), dim3(256), 0, s, nd, d_params, (char*)d_imgb);
}
// Function attributes and occupancy belong to a (function, device) pair: kept per device, set on first use there.
static const void* solve3_fn(bool jvp, bool record, bool multi = false, bool cond = false) {
    if (jvp && record) return (const void*)k_solve3jb<false, false, true>;
    if (jvp) return cond ? (multi ? (const void*)k_solve3jb<true, true> : (const void*)k_solve3jb<false, true>)
                         : (multi ? (const void*)k_solve3jb<true, false> : (const void*)k_solve3jb<false, false>);
    if (record) return (const void*)k_solve3b<true, false, false>;
    return cond ? (multi ? (const void*)k_solve3b<false, true, true> : (const void*)k_solve3b<false, false, true>)
                : (multi ? (const void*)k_solve3b<false, true, false> : (const void*)k_solve3b<false, false, false>);
}
static size_t solve3_shm(bool jvp) { return jvp ? (size_t)s3b::TOTAL_BYTES : (size_t)s3v::TOTAL_BYTES; }
int step3b_solve_resident(bool jvp, bool record, int device) {
    constexpr int MAXDEV = 64;
    static std::mutex mu;
    static int resident[MAXDEV][4];                         // 0: not asked yet, -1: unusable, else workgroups the device holds
    if (device < 0 || device >= MAXDEV) return 0;
    const int which = jvp ? (record ? 3 : 2) : (record ? 1 : 0);
    std::lock_guard<std::mutex> lk(mu);
    int& r = resident[device][which];
    if (r == 0) {
        r = -1;
        int cur = -1, n_cu = 0, per_cu = 0;
        const void* fn = solve3_fn(jvp, record);
        const void* fn_multi = solve3_fn(jvp, record, !record);           // (the several-tiles-per-workgroup instantiation)
        const size_t shm = solve3_shm(jvp);
        if (hipGetDevice(&cur) == hipSuccess && (cur == device || hipSetDevice(device) == hipSuccess)) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess &&
                hipFuncSetAttribute(fn_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess &&
                (record || (hipFuncSetAttribute(solve3_fn(jvp, false, false, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess &&
                            hipFuncSetAttribute(solve3_fn(jvp, false, true, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess)) &&
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, shm) == hipSuccess && per_cu >= 1 && n_cu >= 1)
                // (148 KB of LDS per workgroup: one per CU whatever the API says about registers -- it is known to report
                // one block too many at some SGPR counts, MI355X_MICROARCH.md "Correctness boundaries")
                r = n_cu * 1;
            else (void)hipGetLastError();
            if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        }
    }
    return r > 0 ? r : 0;
}
cnf_status step3b_solve_launch(const MfmaArgs& a, const void* d_imgb, int n_in, int norm_z, int norm_j, int grid, hipStream_t s,
                               const Solve3Args& sv_, bool jvp, int device) {
    const bool record = a.dump != nullptr;
    // Every workgroup must be resident for the whole launch: the grid is bounded by what THIS device holds at once.  An
    // ordinary launch places all of them as soon as the CUs are free; the caller keeps the one-launch solves of this process
    // apart (a mutex); whatever else holds CUs (another stream, process, a CU mask) makes a wait run out, and the caller
    // then streams step launches instead.
    const int resident = step3b_solve_resident(jvp, record, device);
    if (grid < 1 || grid > resident) return CNF_ERR_UNSUPPORTED;
    MfmaArgs a_ = a;
    const char* img = (const char*)d_imgb;
    S3Tab tab = kS3Tab;
    Solve3Args sv = sv_;
    void* args[] = {&a_, &img, &n_in, &norm_z, &norm_j, &tab, &sv};
    const bool multi = (a.B + 31) / 32 > grid;             // several tiles per workgroup: the state lives in a.U / a.K1
    if (multi && (record || !a.K1[0] || !a.K1[1])) return CNF_ERR_UNSUPPORTED;
    const bool cond = a.cond != nullptr;                   // conditional model: per-sample first-layer bias rows
    if (cond && record) return CNF_ERR_UNSUPPORTED;
    if (hipLaunchKernel(solve3_fn(jvp, record, multi, cond), dim3(grid), dim3(512), args, solve3_shm(jvp), s) != hipSuccess) {
        (void)hipGetLastError();
        return CNF_ERR_UNSUPPORTED;
    }
    return CNF_OK;
}
void step3b_launch(const MfmaArgs& a, const void* d_imgb, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_step3b, hipFuncAttributeMaxDynamicSharedMemorySize, s3v::TOTAL_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL(k_step3b, grid, dim3(512), s3v::TOTAL_BYTES, s, a, (const char*)d_imgb, n_in, norm_z, norm_j, kS3Tab, single);
}
void step3jb_launch(const MfmaArgs& a, const void* d_imgb, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_step3jb, hipFuncAttributeMaxDynamicSharedMemorySize, s3b::TOTAL_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL(k_step3jb, grid, dim3(512), s3b::TOTAL_BYTES, s, a, (const char*)d_imgb, n_in, norm_z, norm_j, kS3Tab, single);
}

// Weight image of k_step3 (layout: namespace s3).  Fragment element (wave w, fragment j, lane = 16q + s, c):
//   FR1 (waves 0-3): W1 rows: tile w + 4 (j >> 1), k-block j & 1:  W1[16 (w + 4 (j >> 1)) + s][16 (j & 1) + 4q + c]
//   FR3 (all waves): W3^T rows:                                   W3[16 j + 4q + c][16 w + s]
__global__ void k_pack_step3(NetDesc nd, const float* __restrict__ P, float* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s3::IMG_FLOATS) return;
    float v = 0.f;
    int l = -1, o = 0, k = 0;
    if (i < s3::IMG_W1T) { l = 1; o = i / s3::SW2; k = i % s3::SW2; }                                        // W2[o][k]
    else if (i < s3::IMG_W3R) { const int r = i - s3::IMG_W1T; l = 0; k = r / s3::SXH; o = r % s3::SXH; }     // W1T[k][o]
    else if (i < s3::IMG_B1) { const int r = i - s3::IMG_W3R; l = 2; o = r / s3::SXH; k = r % s3::SXH; }      // W3R[o][k]
    else if (i < s3::IMG_FR1) {
        const int r = i - s3::IMG_B1;
        const int lb = r < s3::PH ? 0 : (r < 2 * s3::PH ? 1 : 2);
        const int ob = r - (lb == 0 ? 0 : (lb == 1 ? s3::PH : 2 * s3::PH));
        if (ob < nd.dims[lb + 1]) v = P[nd.b_off[lb] + ob];
    } else {
        const bool f1 = i < s3::IMG_FR3;
        const int r = i - (f1 ? s3::IMG_FR1 : s3::IMG_FR3);
        const int nf = f1 ? 4 : 2;
        const int w = r / (nf * 256), rem = r % (nf * 256);
        const int j = rem / 256, lane = (rem % 256) / 4, c = rem % 4, s = lane & 15, q = lane >> 4;
        if (f1) { l = 0; o = 16 * (w + 4 * (j >> 1)) + s; k = 16 * (j & 1) + 4 * q + c; }
        else    { l = 2; o = 16 * j + 4 * q + c; k = 16 * w + s; }
    }
    if (l >= 0 && o < nd.dims[l + 1] && k < nd.dims[l]) v = P[nd.w_off[l] + o + (size_t)k * nd.dims[l + 1]];
    img[i] = v;
}

size_t step3_img_floats() { return (size_t)s3::IMG_FLOATS; }

void step3_pack(const NetDesc& nd, const float* d_params, float* d_img3, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_step3, dim3((s3::IMG_FLOATS + 255) / 256), dim3(256), 0, s, nd, d_params, d_img3);
}

void step3_launch(const MfmaArgs& a, const float* d_img3, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_step3, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(s3::TOTAL * sizeof(float)));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_step3, grid, dim3(512), (size_t)s3::TOTAL * sizeof(float), s, a, d_img3, n_in, norm_z, norm_j, kS3Tab, single);
}

void step3j_launch(const MfmaArgs& a, const float* d_img3, int n_in, int norm_z, int norm_j, dim3 grid, hipStream_t s, int single) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_step3j, hipFuncAttributeMaxDynamicSharedMemorySize, s3::TOTAL * (int)sizeof(float));
        attr = true;
    }
    hipLaunchKernelGGL(k_step3j, grid, dim3(512), s3::TOTAL * sizeof(float), s, a, d_img3, n_in, norm_z, norm_j, kS3Tab, single);
}
