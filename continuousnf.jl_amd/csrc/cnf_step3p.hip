// k_solve3p -- the one-launch solve of the headline shape 32 -> 128 -> 128 -> 32 (tanh) in the VJP compute mode
// (src/icnf.jl:318-350 inside base_sol, src/base_icnf.jl:137-143), re-cut so that the matrix pipe and the vector ALU of a
// SIMD work at the same time.  Same arithmetic, meetings, controller and fused input / output as k_solve3b (cnf_step3.hip);
// what differs is WHEN each piece of an evaluation runs:
//
//  * TWO INDEPENDENT CHAINS PER INTERVAL.  The forward sweep of evaluation e+1 needs only zdot of evaluation e (the next
//    stage state), not its reverse sweep.  So the reverse sweep R(e) = [R3: W3^T g3, R2: W2^T g2, R1: W1^T g1] runs
//    interleaved with the forward sweep F(e+1) = [F1: W1 x, F2: W2 h1, F3: W3 h2]:
//        R3(e) F1(e+1) R2(e) F2(e+1) R1(e) F3(e+1)      (one barrier interval each; consecutive products are independent)
//    A step attempt = F(1), five such cycles, R(6).
//  * THE TWO WAVES OF A SIMD IN COMPLEMENTARY ROLES (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Waves 4-7 (Y) run
//    a product's MFMAs and then its epilogue (bias, tanh, sigma', three-piece split, LDS stores); waves 0-3 (X) DEFER each
//    epilogue by one interval -- legal because the next product does not read it -- and run it in front of the next
//    product's MFMAs, holding the accumulators across the barrier (8 VGPRs).  Within an interval a SIMD then has one wave in
//    the matrix pipe while its partner is in the vector ALU, instead of both doing the same thing at the same time.
//  * Y also owns the two narrow products (F3 on W3 rows, R1 on W1^T rows: one wave per SIMD, as before); X owns the
//    Runge-Kutta rows of z IN REGISTERS (u, k1..k6: the 33 KB of LDS they took pay for the two extra activation images the
//    interleaving needs) and does the stage bookkeeping in its idle slots: Y hands zdot over through a 4 KB LDS buffer, X
//    hands back the stage sum without the newest k (PRE), so Y's critical epilogue is one fma per element.
//
// LDS: four K = 128 split images A (h1), B (g2), C (h2), D (g1) -- h1 / h2 can no longer be overwritten in place by g1 /
// g2, the next evaluation's forward sweep is already writing them --, the W3 / W1^T images, and 12.4 KB of fp32 words.  The
// K = 32 images (stage state x, g3) and the zdot hand-over live inside D while g1 is dead.  sigma'_2 is rebuilt from C when
// R3's epilogue needs it; sigma'_1 is taken from A by F1's epilogue of the NEXT evaluation just before it overwrites the
// same elements (8 VGPRs for one interval).
#include "cnf_step3_dev.h"
#include <mutex>

namespace s3p {
constexpr int WS = 256, WP = 32 * WS, WI = 3 * WP;        // K = 128 images: [piece][32 rows][128 bf16], swizzled as s3v
constexpr int NS = 64, NP = 32 * NS, NI = 3 * NP;         // K = 32 images
// fp32 words (float offsets)
constexpr int RED = 0;                                    // partials [|zdot|^2 even | odd evaluation | trace | |eJ|^2][32][8]
constexpr int SC = RED + 4 * 256;                         // scalar-row Runge-Kutta state [32][8][3]
constexpr int BIAS = SC + 32 * 24;                        // b1 (128), b2 (128), b3 (32)
constexpr int MISC = BIAS + 2 * 128 + 32;                 // controller scratch, block reductions, the integrator state
constexpr int PRE = MISC + 64;                            // stage sum without the newest k [32 samples][32 rows]
constexpr int FP_END = PRE + 32 * 32;
// images (byte offsets)
constexpr int BA = FP_END * 4, BB = BA + WI, BC = BB + WI, BD = BC + WI, W3I = BD + WI, W1TI = W3I + WI;
#ifdef S3P_STAMPS
constexpr int STAMPS = W1TI + WI;                         // diagnostics: [wave 0 | wave 4][36 intervals][3 phases] cycle sums
constexpr int TOTAL_BYTES = STAMPS + 2 * 36 * 3 * 4;
#else
constexpr int TOTAL_BYTES = W1TI + WI;
#endif
constexpr int X0S = BD, G3S = BD + NI, ZDB = BD + 2 * NI; // inside D while g1 is dead: x image, g3 image, zdot [32][32] fp32
static_assert(BA % 16 == 0 && (PRE * 4) % 16 == 0 && ZDB % 16 == 0, "16-byte accesses");
static_assert(ZDB + 32 * 32 * 4 <= BD + WI, "the hand-over buffers fit inside D");
static_assert(TOTAL_BYTES <= 160 * 1024, "LDS plan");
static_assert(WI == s3g::WI, "global image");
}  // namespace s3p

template <bool RECORD>
__global__ void __launch_bounds__(512, 2) k_solve3p(MfmaArgs a, const char* __restrict__ imgb, int n_in, int norm_z,
                                                    int norm_j, const S3Tab tab, Solve3Args sv) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* ldsb = reinterpret_cast<char*>(lds);
    const float* img3 = reinterpret_cast<const float*>(imgb);      // (a valid address for masked loads)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = n_in + 3;
    const int s = lane & 15, q = lane >> 4;
    const bool isX = wave < 4;                            // waves 0-3: deferred epilogues, Runge-Kutta rows of z
    const int jj = wave & 3, t = jj & 1, hf = jj >> 1;    // X wave j mirrors Y wave 4 + j: row tile t, sample half hf
    const bool sown = !isX && t == 0 && q == 0;           // waves 4, 6: lane s holds the scalar rows of sample 16 hf + s
    const int smp = 16 * hf + s;                          // sample of this lane in the narrow products / the z rows
    const int r0 = 16 * t + 4 * q;                        // first of its 4 rows there
    const int nv = n_in - r0;                             // valid rows among them (may be <= 0 or > 4)
    const bool wide = (n_in & 3) == 0;
    const int b0 = blockIdx.x * 32 + 16 * hf;
    const bool live = s < max(0, min(16, a.B - b0));
    const size_t gcol = (size_t)(b0 + s) * D;
    const int ce = (!isX && live) ? nv : 0, cu = (isX && live) ? nv : 0, cs = (sown && live) ? 3 : 0;
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
    constexpr int NCI = 2 * s3p::WI / 16, NCB = (2 * 128 + 32) / 4;
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
            __builtin_amdgcn_global_load_lds((glb_c)(imgb + s3g::W3I + 16 * (c + lane)), (lds_c)(ldsb + s3p::W3I + 16 * c), 16, 0, 0);
    }
    float* sc = lds + s3p::SC + smp * 24;
    auto sc_get = [&](int j) { return f32x4{sc[3 * j], sc[3 * j + 1], sc[3 * j + 2], 0.f}; };
    auto sc_set = [&](int j, const f32x4& v) { sc[3 * j] = v.x; sc[3 * j + 1] = v.y; sc[3 * j + 2] = v.z; };
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's LDS-DMA pieces have landed
    __builtin_amdgcn_sched_barrier(0);
    float* msc = lds + s3p::MISC;
    StepState* ns = reinterpret_cast<StepState*>(msc + 44);            // the integrator state (thread 0 runs the controller on it)
    static_assert(sizeof(StepState) <= 20 * sizeof(float), "fits the scratch words");
    if (tid == 0) *ns = sv.init;
    if (tid < NCB) reinterpret_cast<f32x4*>(lds + s3p::BIAS)[tid] = sgb;
    // X: u, k1..k6 of this lane's 4 rows of z.  Y: slot 0 holds its 4 probe rows (g3 and the trace row need them).
    f32x4 rk[7];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 1; j < 7; ++j) rk[j] = zero4;
    rk[0] = isX ? ld4_mask(ru, cu) : ld4_mask(re, ce);
    f32x4 knew = zero4;                                    // X: zdot of the evaluation that finished last
    // the 8 partials (2 row tiles x 4 lanes) of one sample and one kind sit side by side: two b128 reads each
    auto red8 = [&](int kind) {
        const float* r = lds + s3p::RED + (kind * 32 + smp) * 8;
        const f32x4 a_ = *(const f32x4*)r, b_ = *(const f32x4*)(r + 4);
        return ((a_.x + a_.y) + (a_.z + a_.w)) + ((b_.x + b_.y) + (b_.z + b_.w));
    };
    // scalar rows (ldot, Edot, ndot) of the evaluation whose |zdot|^2 partials went to parity p
    auto read_scalars = [&](int p) {
        const float e2 = red8(p), ld = red8(2), n2 = red8(3);
        return f32x4{ld, norm_z ? __builtin_sqrtf(e2) : 0.f, norm_j ? __builtin_sqrtf(n2) : 0.f, 0.f};
    };
    float* redw = lds + s3p::RED + smp * 8 + 4 * t + q;                 // this lane's slot of kind 0 (+ 256 per kind)
    // B operands from a split image: lane (sample s of half A, k = 8q ..); half B = 16 rows on.  Results: lane (sample s,
    // rows 16 wave + 4q ..) of the wide images (swizzled rows: k-block kb is reached by XOR 64 kb on the byte offset)
    const int wb_rd = s * s3p::WS + 16 * (q ^ s);
    const int wb_wr = s * s3p::WS + 16 * ((2 * wave + (q >> 1)) ^ s) + 8 * (q & 1);
    constexpr int HBW = 16 * s3p::WS;
    const int nsw = (-(s >> 2)) & 3;                                    // chunk swizzle of the K = 32 images (rows s, 16 + s)
    const int nw = smp * s3p::NS + 16 * ((2 * t + (q >> 1)) ^ nsw) + 8 * (q & 1);
    char* x0w = ldsb + s3p::X0S + nw;                                   // this lane's 4 rows of the state / g3 images
    char* g3w = ldsb + s3p::G3S + nw;
    const int nb_rd = s * s3p::NS + 16 * (q ^ nsw);                     // their B operands: lane (sample s of half A, k = 8q ..)
    constexpr int HBN = 16 * s3p::NS;
    float* prew = lds + s3p::PRE + smp * 32 + r0;                       // X writes, the mirror lane of Y reads
    float* zdw = reinterpret_cast<float*>(ldsb + s3p::ZDB) + smp * 32 + r0;       // Y writes, the mirror lane of X reads
    const float* bias = lds + s3p::BIAS;

    // this tile's rows
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
    // The workgroups' partials (e, b) -> the sums of all of them, in msc[32], msc[33] for thread 0 (k_solve3b's meeting).
    auto meet = [&](float e_lane, float b_lane) -> bool {
        float e = s3_wave_sum(e_lane), b = s3_wave_sum(b_lane);
        if (lane == 0) { msc[wave] = e; msc[16 + wave] = b; }
        s3_bar();
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
            for (int spin = 0; spin < sv.spin_limit; ++spin) {
                u32x4 wq;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(wq) : "v"(pb + 2 * tid) : "memory");
                const unsigned long long w0 = ((unsigned long long)wq.y << 32) | wq.x, w1 = ((unsigned long long)wq.w << 32) | wq.z;
                if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag) {
                    cp0 = __uint_as_float((unsigned)w0); cp1 = __uint_as_float((unsigned)w1); ok = 1;
                    break;
                }
                if ((spin & 255) == 255 && __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
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

    // ---- the products of an evaluation ----
    enum { P_F1 = 0, P_F2 = 1, P_F3 = 2, P_R3 = 3, P_R2 = 4, P_R1 = 5 };
    f32x4 sg1a = zero4, sg1b = zero4;                      // sigma'_1 of the evaluation whose reverse sweep is under way
    float* dmpw = nullptr;                                 // RECORD: this lane's rows of the current step's slot (null: not filed)
    const size_t dmp_stride = a.dump_stride;
    // MFMAs of a wide product (all waves: tile `wave` of the layer, both sample halves share the A fragments)
    auto mfma_wide = [&](int P, f32x4 (&acc)[2]) __attribute__((always_inline)) {
        acc[0] = zero4; acc[1] = zero4;
        if (P == P_F1 || P == P_R3) {                      // K = 32: one k-block
            const char* src = ldsb + (P == P_F1 ? s3p::X0S : s3p::G3S) + nb_rd;
            S3bOp b[2];
            b[0] = s3b_load(src, s3p::NP);
            b[1] = s3b_load(src + HBN, s3p::NP);
            S3_SB();
            if (P == P_F1) s3b_mm<2>(acc, wF1, b); else s3b_mm<2>(acc, wB3, b);
        } else {                                           // K = 128: four k-blocks
            const char* src = ldsb + (P == P_F2 ? s3p::BA : s3p::BB);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                S3bOp b[2];
                b[0] = s3b_load(src + (wb_rd ^ (64 * kb)), s3p::WP);
                b[1] = s3b_load(src + HBW + (wb_rd ^ (64 * kb)), s3p::WP);
                S3_SB();
                if (P == P_F2) s3b_mm<2>(acc, wF2[kb], b); else s3b_mm<2>(acc, wB2[kb], b);
                S3_SB();
            }
        }
    };
    // epilogue of a wide product.  pipe: the forward sweep of the NEXT evaluation runs beside this reverse sweep -- F1 takes
    // sigma'_1 of the evaluation before it from A before overwriting the same elements, R2 uses those registers
    auto epi_wide = [&](int P, const f32x4 (&acc)[2], bool pipe) __attribute__((always_inline)) {
        if (P == P_F1) {
            const f32x4 bv1 = *(const f32x4*)(bias + 16 * wave + 4 * q);
            char* pa = ldsb + s3p::BA + wb_wr;
            if (pipe) { sg1a = s3_dtanh4(s3b_load4(pa, s3p::WP)); sg1b = s3_dtanh4(s3b_load4(pa + HBW, s3p::WP)); }
            s3b_store4(pa, s3p::WP, s3_tanh4(acc[0] + bv1));
            s3b_store4(pa + HBW, s3p::WP, s3_tanh4(acc[1] + bv1));
        } else if (P == P_F2) {
            const f32x4 bv2 = *(const f32x4*)(bias + 128 + 16 * wave + 4 * q);
            char* pc = ldsb + s3p::BC + wb_wr;
            s3b_store4(pc, s3p::WP, s3_tanh4(acc[0] + bv2));
            s3b_store4(pc + HBW, s3p::WP, s3_tanh4(acc[1] + bv2));
        } else if (P == P_R3) {                            // g2 = (W3^T g3) .* sigma'_2, sigma'_2 from this lane's h2 in C
            const char* pc = ldsb + s3p::BC + wb_wr;
            const f32x4 d2a = s3_dtanh4(s3b_load4(pc, s3p::WP)), d2b = s3_dtanh4(s3b_load4(pc + HBW, s3p::WP));
            char* pb = ldsb + s3p::BB + wb_wr;
            s3b_store4(pb, s3p::WP, acc[0] * d2a);
            s3b_store4(pb + HBW, s3p::WP, acc[1] * d2b);
        } else {                                           // P_R2: g1 = (W2^T g2) .* sigma'_1
            if (!pipe) {
                const char* pa = ldsb + s3p::BA + wb_wr;
                sg1a = s3_dtanh4(s3b_load4(pa, s3p::WP)); sg1b = s3_dtanh4(s3b_load4(pa + HBW, s3p::WP));
            }
            char* pd = ldsb + s3p::BD + wb_wr;
            s3b_store4(pd, s3p::WP, acc[0] * sg1a);
            s3b_store4(pd + HBW, s3p::WP, acc[1] * sg1b);
        }
    };
    // the narrow products (Y, one wave per SIMD): rows 16t..16t+15 of W3 / W1^T against this wave's sample half
    auto narrow_mm = [&](const char* nrA, const char* nrB, f32x4& z0, f32x4& z1) __attribute__((always_inline)) {
        z0 = zero4; z1 = zero4;                                            // two chains (terms 0-2 / 3-5)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const S3bOp av = s3b_load(nrA + (wb_rd ^ (64 * kb)), s3p::WP), bvv = s3b_load(nrB + (wb_rd ^ (64 * kb)), s3p::WP);
            S3_SB();
            z0 = s3b_term<0>(av, bvv, z0); z1 = s3b_term<3>(av, bvv, z1);
            z0 = s3b_term<1>(av, bvv, z0); z1 = s3b_term<4>(av, bvv, z1);
            z0 = s3b_term<2>(av, bvv, z0); z1 = s3b_term<5>(av, bvv, z1);
            S3_SB();
        }
    };
    // F3 of evaluation e (0: a single evaluation): zdot, g3, the |zdot|^2 partial; for e = 1..5 the state of evaluation e + 1
    auto prod_F3 = [&](int e) __attribute__((always_inline)) {
        f32x4 z0, z1;
        narrow_mm(ldsb + s3p::W3I + 16 * t * s3p::WS, ldsb + s3p::BC + 16 * hf * s3p::WS, z0, z1);
        const f32x4 bv3 = *(const f32x4*)(bias + 256 + r0);
        const f32x4 zd = s3_tanh4(z0 + z1 + bv3);                          // padded rows: zero weights and bias -> 0
        s3b_store4(g3w, s3p::NP, rk[0] * s3_dtanh4(zd));                   // g3 = eps .* sigma'_3
        *(f32x4*)zdw = zd;
        redw[(e & 1) * 256] = s3_dot4(zd, zd);
        if (e >= 1 && e <= 5) {                                            // U_{e+2} = PRE(e) + h a_{e+2,e+1} k_{e+1}
            const float c = hstep * tab.a[e + 1][e];
            const f32x4 pre = *(const f32x4*)prew;
            f32x4 xn;
#pragma unroll
            for (int j = 0; j < 4; ++j) xn[j] = fmaf(c, zd[j], pre[j]);
            s3b_store4(x0w, s3p::NP, xn);
            // (gradient path: U_{e+2} filed behind u_n and U_2 of this step; the pointer is null when nothing is recorded)
            if (RECORD && dmpw && e < 5 && nv > 0) { if (nv >= 4) st4_wide(dmpw + (size_t)e * dmp_stride, xn); else st4(dmpw + (size_t)e * dmp_stride, xn, nv); }
        }
    };
    auto prod_R1 = [&]() __attribute__((always_inline)) {                  // eJ = W1^T g1: trace and norm partials (src/icnf.jl:334, :343)
        f32x4 j0, j1;
        narrow_mm(ldsb + s3p::W1TI + 16 * t * s3p::WS, ldsb + s3p::BD + 16 * hf * s3p::WS, j0, j1);
        const f32x4 ej = j0 + j1;
        redw[2 * 256] = -s3_dot4(ej, rk[0]);
        redw[3 * 256] = s3_dot4(ej, ej);
    };
    // X, at the R3 interval of evaluation e (zdot of e is in the hand-over buffer since the barrier before): take it; inside
    // an attempt file it as k_{e+1} and hand back PRE(e + 1) = u + h sum_{j <= e+1} a_{e+3,j} k_j for F3 of evaluation e + 1
    auto rk_take = [&](int e, bool attempt) __attribute__((always_inline)) {
        knew = *(const f32x4*)zdw;
        if (!attempt || e > 5) return;
#pragma unroll
        for (int j = 1; j < 6; ++j) rk[1 + j] = (e == j) ? knew : rk[1 + j];      // k_{e+1} (selects: the array stays in registers)
        if (e > 4) return;
        const float* A = tab.a[e + 2];
        f32x4 pre = rk[0];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const float cj = j <= e ? hstep * A[j] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) pre[c] = fmaf(cj, rk[1 + j][c], pre[c]);
        }
        *(f32x4*)prew = pre;
    };
    // One evaluation at the state image (prog 0: F1 F2 F3 R3 R2 R1, every epilogue in its own interval) or the six
    // evaluations of a step attempt (prog 1: F(1), the five interleaved cycles, R(6)).  ONE loop body, so that every
    // product's code exists once; P, e and the flags are wave-uniform.
#ifdef S3P_STAMPS
    unsigned* stq = reinterpret_cast<unsigned*>(ldsb + s3p::STAMPS) + (wave == 4 ? 108 : 0);
    const bool stamper = blockIdx.x == 7 && lane == 0 && (wave == 0 || wave == 4);
    if (blockIdx.x == 7 && tid < 216) reinterpret_cast<unsigned*>(ldsb + s3p::STAMPS)[tid] = 0;
    auto cyc = [&]() { return (unsigned)__builtin_amdgcn_s_memtime(); };
#define S3P_T(ph) do { if (prog) { const unsigned t_ = cyc(); if (stamper) stq[3 * i + (ph)] += (t_ - tlast) & 0xFFFFF; tlast = t_; } } while (0)
#else
#define S3P_T(ph) do {} while (0)
#endif
    auto run = [&](const int prog) {
        const int nint = prog ? 36 : 6;
        int pend = -1;                                     // X: the wide product whose epilogue is still to run
        f32x4 acc[2], accD[2];
        accD[0] = zero4; accD[1] = zero4;
        int r = 0, c = 1;                                  // position inside the cycles of an attempt
#ifdef S3P_STAMPS
        unsigned tlast = cyc();
#endif
        for (int i = 0; i < nint; ++i) {
            int P, e;
            bool pipe = false;
            if (!prog) { P = i < 3 ? i : (i == 3 ? P_R3 : (i == 4 ? P_R2 : P_R1)); e = 0; }
            else if (i < 3) { P = i; e = 1; }
            else if (i < 33) {
                P = r == 0 ? P_R3 : (r == 1 ? P_F1 : (r == 2 ? P_R2 : (r == 3 ? P_F2 : (r == 4 ? P_R1 : P_F3))));
                e = (r & 1) ? c + 1 : c;
                pipe = true;
                if (++r == 6) { r = 0; ++c; }
            } else { P = i == 33 ? P_R3 : (i == 34 ? P_R2 : P_R1); e = 6; }
            const bool is_wide = P != P_F3 && P != P_R1;
            if (isX) {
                if (pend >= 0) { epi_wide(pend, accD, true); pend = -1; }
                if (P == P_R3) rk_take(e, prog != 0);
                S3_SB(); S3P_T(0); S3_SB();
                if (is_wide) {
                    mfma_wide(P, acc);
                    if (pipe) { accD[0] = acc[0]; accD[1] = acc[1]; pend = P; }
                    else epi_wide(P, acc, false);
                }
            } else {
                if (is_wide) { mfma_wide(P, acc); S3_SB(); S3P_T(0); S3_SB(); epi_wide(P, acc, pipe); }
                else if (P == P_F3) {
                    // scalar rows of the evaluation before (its partials are complete since the last barrier): slot j holds k_j
                    if (pipe && sown) sc_set(e, read_scalars((e - 1) & 1));
                    S3_SB(); S3P_T(0); S3_SB();
                    prod_F3(e);
                } else { S3_SB(); S3P_T(0); S3_SB(); prod_R1(); }
            }
            S3_SB(); S3P_T(1); S3_SB();
            s3_bar();
            S3P_T(2);
        }
    };
    // (x / sk)^2 onto acc, sk = atol + rtol |u|: the expressions of the single-evaluation launches, digit for digit
    auto add_norm = [&](float& acc, float u, float x) {
        const float sk = fmaf(fabsf(u), reltol, abstol);
        const float y = x / sk;
        acc = fmaf(y, y, acc);
    };
    bool alive = true;
    {
        // ---- k1 = f(u0); with the automatic initial dt (Hairer; the two single evaluations of the streamed driver) also
        // its norms, f(u0 + h0 f0) and that norm ----
        if (isX) s3b_store4(x0w, s3p::NP, rk[0]);
        s3_bar();
        run(0);
        float e = 0.f, b = 0.f;
        if (live && isX) {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < nv) { add_norm(e, rk[0][c], rk[0][c]); add_norm(b, rk[0][c], knew[c]); }
        }
        if (isX) rk[1] = knew;                                             // k1 = f(u0)
        if (live && sown) {
            const f32x4 u4 = sc_get(0), f0 = read_scalars(0);
#pragma unroll
            for (int c = 0; c < 3; ++c) { add_norm(e, u4[c], u4[c]); add_norm(b, u4[c], f0[c]); }
            sc_set(1, f0);
        }
        if (sv.hairer) alive = meet(e, b);
        if (sv.hairer && alive) {
            if (tid == 0) { ctrl_phase(ns, 0, msc[32], msc[33], a.n_total); post_ctrl(0); }
            share();
            if (isX) s3b_store4(x0w, s3p::NP, rk[0] + hstep * rk[1]);      // f(u0 + h0 f0)
            s3_bar();
            run(0);
            e = 0.f; b = 0.f;
            if (live && isX) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < nv) add_norm(e, rk[0][c], knew[c] - rk[1][c]);
            }
            if (live && sown) {
                const f32x4 u4 = sc_get(0), f0 = sc_get(1), f1 = read_scalars(0);
#pragma unroll
                for (int c = 0; c < 3; ++c) add_norm(e, u4[c], f1[c] - f0[c]);
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
        if (isX) {
            f32x4 u2, p1;
            const float c21 = hstep * TS_A21, c31 = hstep * tab.a[2][0];
#pragma unroll
            for (int c = 0; c < 4; ++c) { u2[c] = fmaf(c21, rk[1][c], rk[0][c]); p1[c] = fmaf(c31, rk[1][c], rk[0][c]); }
            s3b_store4(x0w, s3p::NP, u2);                                  // U_2 = u + h a21 k1
            *(f32x4*)prew = p1;                                            // PRE(1) = u + h a31 k1
            if (RECORD) {
                dmpw = (live && nacc < a.dump_cap) ? a.dump + (size_t)nacc * a.dump_step_stride + gcol + r0 : nullptr;
                if (dmpw && nv > 0) {
                    if (nv >= 4) { st4_wide(dmpw - dmp_stride, rk[0]); st4_wide(dmpw, u2); }
                    else { st4(dmpw - dmp_stride, rk[0], nv); st4(dmpw, u2, nv); }
                }
            }
        } else if (RECORD) {                               // (Y files the later stage states: the same slot)
            dmpw = (live && nacc < a.dump_cap) ? a.dump + (size_t)nacc * a.dump_step_stride + gcol + r0 : nullptr;
        }
        if (RECORD && blockIdx.x == 0 && tid == 0 && nacc < a.dump_cap) a.hs_out[nacc] = hstep;
        s3_bar();
        run(1);
        float errsum = 0.f, badcnt = 0.f;
        f32x4 un = zero4;
        if (isX) {                                         // u_new = U_7 = PRE(5) + h a76 k6, exactly the state evaluation 6 ran at
            const float c76 = hstep * tab.a[6][5];
            const f32x4 pre = *(const f32x4*)prew;
#pragma unroll
            for (int c = 0; c < 4; ++c) un[c] = fmaf(c76, rk[6][c], pre[c]);
        }
        if (live && isX) {
            f32x4 ez = TS_BT1 * rk[1] + TS_BT7 * knew;
            ez += TS_BT2 * rk[2] + TS_BT3 * rk[3] + TS_BT4 * rk[4] + TS_BT5 * rk[5] + TS_BT6 * rk[6];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float scl = fmaf(fmaxf(fabsf(rk[0][c]), fabsf(un[c])), reltol, abstol);
                const float x = c < nv ? hstep * ez[c] / scl : 0.f;
                errsum = fmaf(x, x, errsum);
                badcnt += (c < nv && !(fabsf(un[c]) <= 3.0e38f)) ? 1.f : 0.f;
            }
        }
        if (live && sown) {
            f32x4 ks[7];
#pragma unroll
            for (int j = 0; j < 6; ++j) ks[j] = sc_get(1 + j);
            const f32x4 us = sc_get(0);
            ks[6] = read_scalars(0);                                       // k7 of the scalar rows (evaluation 6: even parity)
            const f32x4 uns = us + hstep * stage_acc4<6>(ks);
            err_acc(errsum, badcnt, ks, us, uns, hstep, abstol, reltol, 3);
            sc_set(7, uns);                                                // kept for an accepted attempt
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
            if (isX) { rk[0] = un; rk[1] = knew; }
            if (sown) { const f32x4 k7s = read_scalars(0); sc_set(0, sc_get(7)); sc_set(1, k7s); }
        }
    }
    // ---- the final state to the integrator's buffer set 0 ----
    if (live && isX) { if (nv >= 4) st4_wide(a.U[0] + gcol + r0, rk[0]); else st4(a.U[0] + gcol + r0, rk[0], nv); }
    if (live && sown) { const f32x4 us = sc_get(0); float* o = a.U[0] + gcol + n_in; o[0] = us.x; o[1] = us.y; o[2] = us.z; }
    if (sv.logpx && alive) {
        // ---- post-processing of this tile: logp(z) - dlogp, the regulariser rows; then the loss sums of the batch ----
        if (isX) {
            float ss = 0.f, sa = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nv) { ss = fmaf(rk[0][c], rk[0][c], ss); if (r0 + c >= sv.nvars) sa = fmaf(rk[0][c], rk[0][c], sa); }
            redw[0] = ss; redw[256] = sa;
        }
        s3_bar();
        float v4[4] = {0.f, 0.f, 0.f, 0.f};
        if (live && sown) {
            const float ss = red8(0), sa = red8(1);
            const f32x4 us = sc_get(0);
            const float log2pi = 1.8378770664093453f;
            v4[0] = -0.5f * fmaf((float)n_in, log2pi, ss) - us.x;         // base_icnf.jl:177-178
            v4[1] = us.y; v4[2] = us.z;
            v4[3] = (sv.norm_z_aug && sv.naugs > 0) ? sqrtf(sa) : 0.f;    // :179-187
            const size_t b = (size_t)(b0 + s), Bz = (size_t)a.B;
            sv.logpx[b] = v4[0]; sv.regs[b] = v4[1]; sv.regs[Bz + b] = v4[2]; sv.regs[2 * Bz + b] = v4[3];
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
                        for (int spin = 0; spin < sv.spin_limit; ++spin) {
                            const unsigned long long w = __hip_atomic_load(qb + 4 * tid + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(w >> 32) == tag) { c4[j] = __uint_as_float((unsigned)w); got = 1; break; }
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
#ifdef S3P_STAMPS
    s3_bar();
    if (blockIdx.x == 7 && tid == 0) {
        const unsigned* z = reinterpret_cast<const unsigned*>(ldsb + s3p::STAMPS);
        for (int i = 0; i < 36; ++i)
            printf("k_solve3p interval %2d | X: first %7u second %7u barrier %7u | Y: first %7u second %7u barrier %7u\n", i,
                   z[3 * i], z[3 * i + 1], z[3 * i + 2], z[108 + 3 * i], z[108 + 3 * i + 1], z[108 + 3 * i + 2]);
    }
#endif
    if (blockIdx.x == 0 && tid == 0) {
        // A wait that ran out anywhere (this workgroup's `alive`, or another's abort word -- set before this workgroup
        // could have passed the meeting in question) makes the launch void: the published state says so by itself
        // (n_partials < 0), so that the host can tell for THIS launch even when others are queued behind it.
        if (!alive || __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ns->done = 0; ns->n_partials = -1; }
        __hip_atomic_store(sv.base_dev, mbase + (unsigned)nsync + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (workgroups differ by <= 1 meeting; + the sums)
        ns->cur = 0;
        *a.st_out = *ns;
        if (sv.t_out) { sv.t_out[1] += __builtin_amdgcn_s_memrealtime() - sv.t_out[0]; sv.t_out[2] += 1; }
        publish_mirror(a, *ns);
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
static const void* solve3p_fn(bool record) { return record ? (const void*)k_solve3p<true> : (const void*)k_solve3p<false>; }
int step3p_solve_resident(bool record, int device) {
    constexpr int MAXDEV = 64;
    static std::mutex mu;
    static int resident[MAXDEV][2];                         // 0: not asked yet, -1: unusable, else workgroups the device holds
    if (device < 0 || device >= MAXDEV) return 0;
    std::lock_guard<std::mutex> lk(mu);
    int& r = resident[device][record ? 1 : 0];
    if (r == 0) {
        r = -1;
        int cur = -1, n_cu = 0, per_cu = 0;
        const void* fn = solve3p_fn(record);
        if (hipGetDevice(&cur) == hipSuccess && (cur == device || hipSetDevice(device) == hipSuccess)) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, s3p::TOTAL_BYTES) == hipSuccess &&
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, s3p::TOTAL_BYTES) == hipSuccess && per_cu >= 1 && n_cu >= 1)
                r = n_cu;                                       // (156 KB of LDS per workgroup: one per CU)
            else (void)hipGetLastError();
            if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        }
    }
    return r > 0 ? r : 0;
}
cnf_status step3p_solve_launch(const MfmaArgs& a, const void* d_imgb, int n_in, int norm_z, int norm_j, int grid, hipStream_t s,
                               const Solve3Args& sv_, int device) {
    const bool record = a.dump != nullptr;
    if (grid < 1 || grid > step3p_solve_resident(record, device)) return CNF_ERR_UNSUPPORTED;
    MfmaArgs a_ = a;
    const char* img = (const char*)d_imgb;
    S3Tab tab = kS3Tab;
    Solve3Args sv = sv_;
    void* args[] = {&a_, &img, &n_in, &norm_z, &norm_j, &tab, &sv};
    if (hipLaunchKernel(solve3p_fn(record), dim3(grid), dim3(512), args, s3p::TOTAL_BYTES, s) != hipSuccess) {
        (void)hipGetLastError();
        return CNF_ERR_UNSUPPORTED;
    }
    return CNF_OK;
}
