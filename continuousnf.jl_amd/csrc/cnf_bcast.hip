// k_solve_bcast -- the whole solve of BASELINE config 5's network (n_in <= 128, hidden <= 384, two tanh layers) at EIGHT samples
// per CU, in one launch.
//
// What the reference runs here: `augmented_f` in TrainMode / VJP (src/icnf.jl:318-350) and in TestMode with the exact trace
// (src/icnf.jl:148-164 over src/utils.jl:1-36: n_in one-hot pullbacks and an n_in x n_in x B tensor -- here the closed
// form of two-layer networks, tr J = sigma'_1^T (W_1 .* W_2^T) sigma'_2, one extra product) inside `base_sol`
// (src/base_icnf.jl:137-143) with `inference_prob` / `inference_sol` (:266-286, :167-189) around it.
//
// Why a kernel of its own.  BASELINE config 5 has 2048 columns: 8 per CU.  k_mfma's v_mfma_f32_16x16x4_f32 needs 16 columns
// per wave, so half of the chip idles, and every workgroup streams ALL weights (786 KB in both orientations) from L2 per
// evaluation -- at the ~60-70 GB/s a CU sustains from L2 (tools/ubench/mfma4x4_stream.hip) that alone is 11 us per
// evaluation: the round-3 kernel (16 us) was bound by that stream, not by its arithmetic.  Here:
//   * v_mfma_f32_4x4x1_16B_f32 with CBSZ = 4: the A operand of block ABID is broadcast to all 16 blocks, so with A = the
//     activations (lane 4j + s = x[k0 + j][sample s]: ONE register carries 16 k's of 4 samples, ABID = j picks one) and B =
//     weights (lane = output feature) an instruction forms D[sample 0..3][64 features] += x[k][sample] W[feature][k]:
//     exact fp32, the full fp32 MFMA rate (8 cycles, measured 8.1-8.5), and NO lane spent on samples that do not exist.
//     A workgroup of 4 waves (one per SIMD, 512 registers each) owns 8 samples = two sample groups that share every B
//     register.
//   * W_1 is RESIDENT in registers for the whole solve, in both orientations (2 x 192 registers per lane: forward, and W_1^T
//     for the reverse sweep -- in TestMode C = W_1 .* W_2^T takes the second set); only W_2 / W_2^T stream from L2, as ONE
//     cyclic stream per wave that runs a ring of requests ahead across phase and evaluation boundaries (393 KB per
//     evaluation and CU in TrainMode, 197 KB in TestMode).
//   * per layer the waves split the (output tile, k) work evenly -- 768 B registers' worth each way, 192 per wave: a full
//     tile and a half, the halves of a shared tile added in the elementwise phase that follows --, exchange through two
//     small LDS images (partial outputs [feature][4 samples]; the next layer's A image in operand order), and meet the
//     other workgroups once per step attempt through the tagged words of k_solve3b.
#include <cstdlib>
#include <mutex>

#include "cnf_bcast.h"
#include "cnf_mfma_dev.h"

namespace {

constexpr int BC_RING = 8;                                 // 16-byte requests in flight per lane on the W2 stream

struct BcArgs {
    NetDesc nd;
    const float* P;        // flat parameters (biases)
    const float* img;      // packed images (bcast_pack)
    const float* eps;
    int B;
    float n_total;
    float* U0;
    StepState* st_out;
    void* mirror;
    unsigned seq;
    float* store;          // MULTI: the tiles' Runge-Kutta rows (bcast_store_floats)
    const float* cond;     // conditional models: the per-sample first-layer bias W1y ys + b1, [B][cbs]   (src/base_icnf.jl:288-309)
    int cbs;
    // RECORD (gradient path, as k_solve3b<RECORD>): every attempt files u_n and its stage states U_2..U_6 (z rows, [B][D]) in the
    // trajectory slot of step `naccept` -- dump = the U_2 array of slot 0, u_n sits dump_n floats in front of it -- and its signed
    // step size in hs_out
    float* dump; size_t dump_n, dump_slot; int dump_cap; float* hs_out;
};
struct BcTab { float a[7][8]; };
static const BcTab kBcTab = {{
    {0, 0, 0, 0, 0, 0, 0, 0},
    {TS_A21, 0, 0, 0, 0, 0, 0, 0},
    {TS_A31, TS_A32, 0, 0, 0, 0, 0, 0},
    {TS_A41, TS_A42, TS_A43, 0, 0, 0, 0, 0},
    {TS_A51, TS_A52, TS_A53, TS_A54, 0, 0, 0, 0},
    {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65, 0, 0, 0},
    {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76, 0, 0}}};

// Work and images.  A product is 12 blocks of 16 k's per wave: B register `flat = 192 w + 16 i + j` (block i, k j) of
//   out 384 x K 128 products (KIND 0: W1, W2^T, C): tile = flat / 128, k = flat % 128
//   out 128 x K 384 products (KIND 1: W2, W1^T):    tile = flat / 384, k = flat % 384          value = M[64 tile + lane][k]
// HALF of every product's blocks are resident, half stream, so that the stream flows evenly over the evaluation (with W1
// wholly resident and W2 wholly streamed the streamed phases ran at the stream's rate, 6-9 k cycles against 4 k):
//   TrainMode: phases W1, W2, W2^T, W1^T; odd blocks stream (6 x 4 requests per phase, 96 per evaluation), even blocks are
//              resident (96 registers per phase, 384 in all);
//   TestMode:  phases W1, W2, C; blocks 2, 5, 8, 11 stream (16 requests per phase, 48 per evaluation), the others are
//              resident (128 per phase, 384 in all).
// Image per mode and wave: [resident: 96 x (lane 64 x 4)] then [stream: SLEN x (lane 64 x 4)] floats.
//   TrainMode, JVP compute mode (src/icnf.jl:384-420): ONE forward sweep of the state and the tangent columns -- phases
//              W1 [z, eps] and W2 [h1, tau1], no transposed weights -- so W1 and W2 are resident WHOLLY (2 x 192) and nothing
//              streams at all.
constexpr int BC_NRES = 384;
constexpr int BC_VJP = 0, BC_TESTM = 1, BC_JVP = 2;       // kernel modes
// MULTI (more tiles than the device holds workgroups: B > 8 x CUs): every workgroup carries several 8-sample tiles, one after the
// other per stage, and their Runge-Kutta rows live in a store in global memory instead of registers / LDS -- per tile
// BC_TROWS rows of 256 x f32x4 in the owner threads' own order (a thread only ever reads what it wrote itself: no visibility
// question, fully coalesced 4 KB rows, L2 / MALL resident) and 8 x 32 scalars.  Rows: u and k1 twice (accepted / candidate, by the
// parity `cur` -- an accepted step flips it, nothing is copied), k2..k6, the probe:
constexpr int BC_TROWS = 10, BC_R_U = 0, BC_R_K1 = 2, BC_R_K2 = 4, BC_R_EP = 9;
constexpr int BC_TSC = 8 * 32;                             // scalars per tile: [sample][row 0..8: u x2, k1 x2, k2..k6][3], padded to 32
__host__ __device__ constexpr int bc_slen(int mode) { return mode == BC_VJP ? 96 : (mode == BC_TESTM ? 48 : 0); }
__host__ __device__ constexpr int bc_wave_floats(int mode) { return (BC_NRES + 4 * bc_slen(mode)) * 64; }
constexpr int BC_TRAIN = 0, BC_TEST = 4 * bc_wave_floats(BC_VJP), BC_JVPI = BC_TEST + 4 * bc_wave_floats(BC_TESTM),
              BC_IMG_FLOATS = BC_JVPI + 4 * bc_wave_floats(BC_JVP);
__host__ __device__ constexpr bool bc_streamed(int mode, int i) { return mode == BC_VJP ? (i & 1) : (mode == BC_TESTM ? (i % 3 == 2) : false); }

__global__ void k_bcast_pack(NetDesc nd, const float* __restrict__ P, float* __restrict__ img) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= BC_IMG_FLOATS) return;
    const int mode = e >= BC_JVPI ? BC_JVP : (e >= BC_TEST ? BC_TESTM : BC_VJP);
    const bool test = mode == BC_TESTM;
    const int x = e - (mode == BC_JVP ? BC_JVPI : (test ? BC_TEST : BC_TRAIN));
    const int wf = bc_wave_floats(mode);
    const int w = x / wf, y = x % wf;
    const int lane = (y / 4) % 64, c = y % 4;
    int p, i, j;                                           // phase, block, k within the block
    if (y < BC_NRES * 64) {                                // resident register r = 4 (y / 256) + c
        const int r = 4 * (y / 256) + c;
        if (mode == BC_JVP) { p = r / 192; i = (r % 192) / 16; }
        else if (test) { p = r / 128; const int rb = (r % 128) / 16; i = rb + rb / 2; }
        else { p = r / 96; i = 2 * ((r % 96) / 16); }
        j = r % 16;
    } else {                                               // stream request q, element c
        const int q = (y - BC_NRES * 64) / 256;
        if (test) { p = q / 16; i = 3 * ((q % 16) / 4) + 2; }
        else { p = q / 24; i = 2 * ((q % 24) / 4) + 1; }
        j = 4 * (q % 4) + c;
    }
    const int flat = 192 * w + 16 * i + j;
    const int n_in = nd.n_in, nh = nd.dims[1];
    auto w1 = [&](int o, int k) { return (o < nh && k < n_in) ? P[nd.w_off[0] + o + (size_t)k * nh] : 0.f; };
    auto w2 = [&](int o, int k) { return (o < n_in && k < nh) ? P[nd.w_off[1] + o + (size_t)k * n_in] : 0.f; };
    const bool kind1 = p == 1 || p == 3;                   // out 128 x K 384
    const int o = 64 * (flat / (kind1 ? 384 : 128)) + lane, k = flat % (kind1 ? 384 : 128);
    float v;
    if (p == 0) v = w1(o, k);
    else if (p == 1) v = w2(o, k);
    else if (p == 2) v = test ? w1(o, k) * w2(k, o) : w2(k, o);
    else v = w1(k, o);
    img[e] = v;
}

__device__ __forceinline__ f32x4 mfma_bc(float a, float b, f32x4 c, int abid) {   // ABID is an instruction field: a literal per call
    switch (abid & 15) {
#define BC_CASE(J) case J: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, J, 0);
        BC_CASE(0) BC_CASE(1) BC_CASE(2) BC_CASE(3) BC_CASE(4) BC_CASE(5) BC_CASE(6) BC_CASE(7)
        BC_CASE(8) BC_CASE(9) BC_CASE(10) BC_CASE(11) BC_CASE(12) BC_CASE(13) BC_CASE(14)
        default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
#undef BC_CASE
    }
}
__device__ __forceinline__ void bc_bar() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float bc_uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// sum over the lanes of the wave with this lane's parity (the two sample groups alternate along the lanes)
__device__ __forceinline__ float bc_parity_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]: lane ^ 2
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));    // row_ror:4
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));    // row_ror:8
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float bc_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}

struct I0 { static constexpr int value = 0; };
struct I1 { static constexpr int value = 1; };
struct I2 { static constexpr int value = 2; };
struct I3 { static constexpr int value = 3; };

template <int MODE, bool MULTI, bool RECORD = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_solve_bcast(BcArgs a, Solve3Args sv, const BcTab tab) {
    static_assert(!RECORD || (MODE != BC_TESTM && !MULTI), "the recording form: TrainMode (either compute mode), one tile per workgroup");
    constexpr bool TEST = MODE == BC_TESTM, JVP = MODE == BC_JVP;
    constexpr int NS = TEST ? 1 : 3;
    constexpr int NK = JVP ? 2 : 1;                        // operand kinds side by side: the state columns, and (JVP) the tangent columns
    constexpr int NLB = JVP ? 3 : 4;                       // 16-register blocks of the resident set that live in LDS
    __shared__ __attribute__((aligned(16))) float Aimg[NK][2][24][64];      // the next product's A operands: [kind][sample group][16 k's][lane 4 j + s]
    __shared__ __attribute__((aligned(16))) float Pbuf[NK][6][2][2][64][4]; // partial outputs: [kind][tile][K half][sample group][feature][4 samples]
    __shared__ __attribute__((aligned(16))) float Kz[MULTI ? 1 : 7][256][4]; // Runge-Kutta rows k1..k7 of the z rows, next to their owner threads (MULTI: in the store)
    __shared__ __attribute__((aligned(16))) float cbias[JVP || MULTI ? 1 : 384][8];   // conditional models: this tile's first-layer bias, [hidden unit][sample]
    __shared__ __attribute__((aligned(16))) float red[4][2][3][4];          // per wave, sample group, quantity: 4 samples
    __shared__ __attribute__((aligned(16))) float wBL[4][NLB][4][64][4];     // the last resident B registers of each wave: [wave][block][quad][lane][4] (a lane's quad = one ds_read_b128)
    __shared__ float Ssc[8][8][3];                                           // scalar rows: [sample][u, k1..k7][dlogp, E, n]
    __shared__ float msc[48];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NetDesc& nd = a.nd;
    const int n_in = nd.n_in, nh = nd.dims[1], D = n_in + NS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (sv.t_out && blockIdx.x == 0 && tid == 0) sv.t_out[0] = __builtin_amdgcn_s_memrealtime();
    // ---- resident B registers: 384 per lane, once per solve.  The first 240 are PINNED to the accumulator half of the register
    // file (an "a" operand: an MFMA reads its B operand from there directly -- left to itself the allocator keeps all the
    // load destinations in the 256 architectural registers at once and spills the weights to scratch for the whole solve),
    // 80 are ordinary registers, the last 64 live in LDS and are read a block ahead of their use. ----
    constexpr int NRA = 240, NRV = BC_NRES - NRA - 16 * NLB;
    float RA[NRA], RV[NRV];
    const float* wimg = a.img + (JVP ? BC_JVPI : (TEST ? BC_TEST : BC_TRAIN)) + (size_t)wave * bc_wave_floats(MODE);
    {
        const f32x4* pr = reinterpret_cast<const f32x4*>(wimg) + lane;
#pragma unroll
        for (int i = 0; i < BC_NRES / 4; ++i) {
            const f32x4 v = pr[(size_t)i * 64];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int r = 4 * i + c;
                if (r < NRA) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(RA[r]) : "v"(v[c]));
                else if (r < NRA + NRV) RV[r - NRA] = v[c];
                else { const int t = r - NRA - NRV; wBL[wave][t >> 4][(t >> 2) & 3][lane][t & 3] = v[c]; }
            }
        }
    }
    // ---- the stream: SLEN requests per evaluation and wave, cyclic, a ring of them in flight across phase and evaluation
    // boundaries.  Buffer loads: descriptor + the lane's 16-byte slot + a SCALAR offset per request (with flat addresses the
    // compiler precomputed the 64-bit addresses of a whole cycle, 192 registers of them). ----
    constexpr int SLEN = bc_slen(MODE);
    const auto srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wimg + BC_NRES * 64), 0, (SLEN ? SLEN : 1) * 1024, 0x00020000);
    auto sload = [&](int pos) __attribute__((always_inline)) {       // request number `pos` of the cycle
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, 16 * lane, pos * 1024, 0));
    };
    f32x4 ring[BC_RING];
    int spos = 0;                                          // next request to issue (0 .. SLEN-1)
    if (SLEN > 0) {
#pragma unroll
        for (int p = 0; p < BC_RING; ++p) { ring[p] = sload(spos); spos = spos + 1 == SLEN ? 0 : spos + 1; }
    }

    // ---- ownership of the elementwise work ----
    // 128-row arrays (z, zdot, eps^T J): thread -> (tile ot, feature of, sample group osg): one f32x4 = 4 samples
    const int ot = tid >> 7, of = (tid & 127) >> 1, osg = tid & 1;
    const int ok_row = 64 * ot + of;                       // feature
    const bool orow_ok = ok_row < n_in;
    int tile = blockIdx.x;                                 // the 8-sample tile being worked on (MULTI: blockIdx.x, + gridDim.x, ...)
    int smp0 = tile * 8 + 4 * osg;                         // first of this thread's 4 samples
    float omask[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) omask[s] = (orow_ok && smp0 + s < a.B) ? 1.f : 0.f;
    auto set_tile = [&](int t) __attribute__((always_inline)) {
        tile = t; smp0 = t * 8 + 4 * osg;
#pragma unroll
        for (int s = 0; s < 4; ++s) omask[s] = (orow_ok && smp0 + s < a.B) ? 1.f : 0.f;
    };
    float* const aimg_own = &Aimg[0][osg][ok_row >> 4][4 * (ok_row & 15)];
    // 384-row arrays: three (tile, feature, sample group) units per thread
    int h_t[3], h_f[3], h_sg[3];
    float b1v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = tid + 256 * i;
        h_t[i] = u >> 7; h_f[i] = (u & 127) >> 1; h_sg[i] = u & 1;
        const int k = 64 * h_t[i] + h_f[i];
        b1v[i] = k < nh ? a.P[nd.b_off[0] + k] : 0.f;
    }
    const float b2v = orow_ok ? a.P[nd.b_off[1] + ok_row] : 0.f;
    // state of the owned z rows; probes
    f32x4 uz = zero4, ep = zero4;
    auto load_u0 = [&]() __attribute__((always_inline)) {   // (of the current tile)
        uz = zero4; ep = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (omask[s] != 0.f) {
                const size_t sb = (size_t)(smp0 + s);
                if (sv.xs) uz[s] = ok_row < sv.nvars ? sv.xs[sb * sv.nvars + ok_row] : 0.f;
                else uz[s] = sv.u0[sb * D + ok_row];
                if (!TEST) ep[s] = a.eps[sb * n_in + ok_row];
            }
        }
    };
    if (!MULTI) {
        load_u0();
#pragma unroll
        for (int j = 0; j < (MULTI ? 1 : 7); ++j) *(f32x4*)Kz[j][tid] = zero4;
        if (tid < 8) {
            const int smp = blockIdx.x * 8 + tid;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) Ssc[tid][j][q] = 0.f;
            if (!sv.xs && smp < a.B)
                for (int q = 0; q < NS; ++q) Ssc[tid][0][q] = sv.u0[(size_t)smp * D + n_in + q];
        }
    }
    // conditional models: nn(vcat(z, ys)) -- the conditioning columns of W1 act as a per-sample bias that the handle forms once per
    // call (cnf_set_cond); the tile's 8 x nh of them are staged in LDS once per solve and replace b1 in the first elementwise phase
    const bool has_cond = !JVP && !MULTI && a.cond != nullptr;
    if (has_cond) {
        for (int idx = tid; idx < 8 * 384; idx += 256) {
            const int s8 = idx / 384, r = idx - 384 * s8, smp = blockIdx.x * 8 + s8;
            cbias[JVP || MULTI ? 0 : r][s8] = (r < nh && smp < a.B) ? a.cond[(size_t)smp * a.cbs + r] : 0.f;
        }
    }
    const bool even = (wave & 1) == 0;

    // ---- one evaluation of augmented_f at the owned rows `z` -> zdot (owned rows); the scalar rows land in msc[8 q + sample] ----
    f32x4 d1[3];
    // One product (phase PH of the evaluation): 12 blocks of 16 k's, resident or streamed by the mode's pattern.  Block i's A
    // registers (one per sample group and operand kind: 16 k's x 4 samples) and, for the resident registers that live in LDS, its
    // 16 B values are requested one block AHEAD; sample groups, kinds and the two parities of k keep 4 (JVP: 8) accumulator
    // chains going (2 chains: 8.5 cycles per instruction measured, 4: 8.3).
    //   KIND 0: out 384 x K 128 (three 64-k segments -> Pbuf[.][(3 w + seg) / 2][(3 w + seg) % 2]);  1: out 128 x K 384 (-> Pbuf[.][w / 2][w % 2])
    auto product = [&](auto kind_c, auto phase_c) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value, PH = decltype(phase_c)::value;
        constexpr int NRP = JVP ? 192 : (TEST ? 128 : 96); // resident registers per phase
        constexpr int NA = 2 * NK;                         // A registers per block: [kind][sample group]
        f32x4 acc[NA][2];                                  // [..][parity of k]; a segment's sums leave for Pbuf when it ends
#pragma unroll
        for (int x = 0; x < 2 * NA; ++x) acc[x >> 1][x & 1] = zero4;
        auto kb_of = [&](int i) { return KIND == 0 ? (even ? (i & 7) : ((i + 4) & 7)) : 12 * (wave & 1) + i; };
        float a_cur[NA], a_nxt[NA];
#pragma unroll
        for (int x = 0; x < NA; ++x) { a_cur[x] = Aimg[x >> 1][x & 1][kb_of(0)][lane]; a_nxt[x] = 0.f; }
        float bl_cur[16], bl_nxt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { bl_cur[j] = 0.f; bl_nxt[j] = 0.f; }
        // resident register of block i (a resident one), k j
        auto rbase = [&](int i) { return PH * NRP + 16 * (JVP ? i : (TEST ? i - (i + 1) / 3 : i / 2)); };
        auto tail = [&](float (&dst)[16], int rb) __attribute__((always_inline)) {       // the 16 values of an LDS-resident block: four 16-byte reads
            const int blk = (rb - NRA - NRV) >> 4;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const f32x4 v = *(const f32x4*)wBL[wave][blk][c4][lane];
#pragma unroll
                for (int c = 0; c < 4; ++c) dst[4 * c4 + c] = v[c];
            }
        };
        if (!bc_streamed(MODE, 0) && rbase(0) >= NRA + NRV) tail(bl_cur, rbase(0));
        int nreq = 0;                                      // requests consumed in this phase (a compile-time count after unrolling)
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (i + 1 < 12) {
#pragma unroll
                for (int x = 0; x < NA; ++x) a_nxt[x] = Aimg[x >> 1][x & 1][kb_of(i + 1)][lane];
                if (!bc_streamed(MODE, i + 1) && rbase(i + 1) >= NRA + NRV) tail(bl_nxt, rbase(i + 1));
            }
            if (bc_streamed(MODE, i)) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int slot = (nreq + c4) % BC_RING;
#ifdef BC_ABL_NOSTREAM
                    // (ablation: the streamed half of the weights replaced by resident registers -- wrong numbers, the time of a
                    // kernel whose stream costs nothing)
                    const f32x4 bq = {RV[(4 * c4) % NRV], RV[(4 * c4 + 1) % NRV], RV[(4 * c4 + 2) % NRV], RV[(4 * c4 + 3) % NRV]};
#else
                    const f32x4 bq = ring[slot];
                    ring[slot] = sload(spos); spos = spos + 1 == SLEN ? 0 : spos + 1;
#endif
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int x = 0; x < NA; ++x) acc[x][c & 1] = mfma_bc(a_cur[x], bq[c], acc[x][c & 1], 4 * c4 + c);
                }
                nreq += 4;
            } else {
                const int rb = rbase(i);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int r = rb + j;
                    const float bw = r < NRA ? RA[r < NRA ? r : 0] : (r < NRA + NRV ? RV[(r >= NRA && r < NRA + NRV) ? r - NRA : 0] : bl_cur[j]);
#pragma unroll
                    for (int x = 0; x < NA; ++x) acc[x][j & 1] = mfma_bc(a_cur[x], bw, acc[x][j & 1], j);
                }
            }
#pragma unroll
            for (int x = 0; x < NA; ++x) a_cur[x] = a_nxt[x];
#pragma unroll
            for (int j = 0; j < 16; ++j) bl_cur[j] = bl_nxt[j];
            if (KIND == 0 ? (i & 3) == 3 : i == 11) {      // the segment is complete
                const int fs = KIND == 0 ? 3 * wave + (i >> 2) : wave;
#pragma unroll
                for (int x = 0; x < NA; ++x) {
                    *(f32x4*)Pbuf[x >> 1][fs >> 1][fs & 1][x & 1][lane] = acc[x][0] + acc[x][1];
                    acc[x][0] = zero4; acc[x][1] = zero4;
                }
            }
        }
    };
#ifdef BC_STAMPS
    unsigned long long bst[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long bt_ = 0;
#define BC_T(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); bst[i] += n_ - bt_; bt_ = n_; }
#else
#define BC_T(i)
#endif
    auto rhs = [&](const f32x4& z, f32x4& zd) __attribute__((always_inline)) {
#ifdef BC_STAMPS
        bt_ = __builtin_amdgcn_s_memtime();
#endif
        *(f32x4*)aimg_own = z;
        if (JVP) *(f32x4*)&Aimg[NK - 1][osg][ok_row >> 4][4 * (ok_row & 15)] = ep;       // the tangent seed tau0 = eps beside the state
        bc_bar();
        BC_T(0)
        product(I0{}, I0{});                               // phase 0: W1 z  (JVP: W1 [z, eps])
        BC_T(1)
        bc_bar();
        BC_T(0)
        f32x4 ldc = zero4, n2c = zero4, e2c = zero4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = 64 * h_t[i] + h_f[i];
            f32x4 v = *(const f32x4*)Pbuf[0][h_t[i]][0][h_sg[i]][h_f[i]] + *(const f32x4*)Pbuf[0][h_t[i]][1][h_sg[i]][h_f[i]];
            if (has_cond) v += *(const f32x4*)&cbias[JVP || MULTI ? 0 : k][4 * h_sg[i]];
            else v += b1v[i];
            f32x4 h;
#pragma unroll
            for (int s = 0; s < 4; ++s) { h[s] = tanh_fast(v[s]); d1[i][s] = fmaf(-h[s], h[s], 1.f); }
            *(f32x4*)&Aimg[0][h_sg[i]][k >> 4][4 * (k & 15)] = h;
            if (JVP) {                                     // tau1 = sigma'_1 .* (W1 eps)
                const f32x4 w = *(const f32x4*)Pbuf[NK - 1][h_t[i]][0][h_sg[i]][h_f[i]] + *(const f32x4*)Pbuf[NK - 1][h_t[i]][1][h_sg[i]][h_f[i]];
                *(f32x4*)&Aimg[NK - 1][h_sg[i]][k >> 4][4 * (k & 15)] = w * d1[i];
            }
        }
        BC_T(5)
        bc_bar();
        BC_T(0)
        product(I1{}, I1{});                               // phase 1: W2 h1  (JVP: W2 [h1, tau1])
        BC_T(2)
        bc_bar();
        BC_T(0)
        f32x4 d2;
        {
            const f32x4 v = *(const f32x4*)Pbuf[0][ot][0][osg][of] + *(const f32x4*)Pbuf[0][ot][1][osg][of] + b2v;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float h = tanh_fast(v[s]) * omask[s];
                zd[s] = h; d2[s] = fmaf(-h, h, 1.f) * omask[s]; e2c[s] = h * h;
            }
            if (JVP) {                                     // J eps = sigma'_2 .* (W2 tau1):  ldot = -eps . (J eps),  ndot = |J eps|   (src/icnf.jl:404-413)
                const f32x4 w = *(const f32x4*)Pbuf[NK - 1][ot][0][osg][of] + *(const f32x4*)Pbuf[NK - 1][ot][1][osg][of];
#pragma unroll
                for (int s = 0; s < 4; ++s) { const float t2 = w[s] * d2[s]; ldc[s] = t2 * ep[s]; n2c[s] = t2 * t2; }
            } else {
                *(f32x4*)aimg_own = TEST ? d2 : ep * d2;    // g2 = eps .* sigma'_2 (VJP seed)  |  sigma'_2 (exact trace)
            }
        }
        if (!JVP) {
            BC_T(5)
            bc_bar();
            BC_T(0)
            product(I0{}, I2{});                           // phase 2: W2^T g2  |  C sigma'_2
            BC_T(3)
            bc_bar();
            BC_T(0)
            if (TEST) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x4 v = *(const f32x4*)Pbuf[0][h_t[i]][0][h_sg[i]][h_f[i]] + *(const f32x4*)Pbuf[0][h_t[i]][1][h_sg[i]][h_f[i]];
                    ldc += v * d1[i];                      // (the units of a thread share its sample group: tid & 1)
                }
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f32x4 v = *(const f32x4*)Pbuf[0][h_t[i]][0][h_sg[i]][h_f[i]] + *(const f32x4*)Pbuf[0][h_t[i]][1][h_sg[i]][h_f[i]];
                    const int k = 64 * h_t[i] + h_f[i];
                    *(f32x4*)&Aimg[0][h_sg[i]][k >> 4][4 * (k & 15)] = v * d1[i];      // g1
                }
                BC_T(5)
                bc_bar();
                BC_T(0)
                product(I1{}, I3{});                       // phase 3: eps^T J = W1^T g1
                BC_T(4)
                bc_bar();
                BC_T(0)
                const f32x4 eJ = *(const f32x4*)Pbuf[0][ot][0][osg][of] + *(const f32x4*)Pbuf[0][ot][1][osg][of];
#pragma unroll
                for (int s = 0; s < 4; ++s) { const float e = eJ[s] * omask[s]; ldc[s] = e * ep[s]; n2c[s] = e * e; }
            }
        }
        // sums over the features: the lanes of a wave with this lane's parity, then the four waves
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float l_ = bc_parity_sum(ldc[s]);
            const float e_ = TEST ? 0.f : bc_parity_sum(e2c[s]), n_ = TEST ? 0.f : bc_parity_sum(n2c[s]);
            if (lane < 2) { red[wave][lane][0][s] = l_; red[wave][lane][1][s] = e_; red[wave][lane][2][s] = n_; }
        }
        bc_bar();
        if (tid < 24) {                                    // thread -> (quantity q, sample s8)
            const int q = tid >> 3, s8 = tid & 7;
            const float t = (red[0][s8 >> 2][q][s8 & 3] + red[1][s8 >> 2][q][s8 & 3]) + (red[2][s8 >> 2][q][s8 & 3] + red[3][s8 >> 2][q][s8 & 3]);
            float r = q == 0 ? -t : __builtin_amdgcn_sqrtf(t);
            if (q == 1 && !nd.norm_z) r = 0.f;
            if (q == 2 && !nd.norm_j) r = 0.f;
            msc[8 * q + s8] = r;                           // (read by the scalar owners after the next barrier)
        }
        BC_T(6)
    };

    // ---- integrator: every thread carries the state and runs the controller on the same sums ----
    StepState ns = sv.init;
    float hstep = ns.h, abstol = ns.abstol, reltol = ns.reltol;
    int nsync = 0;
    const unsigned mbase = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sv.base_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float p0 = 0.f, p1 = 0.f;
    auto meet = [&](float e_lane, float b_lane) -> bool {
        float e = bc_wave_sum(e_lane), b = bc_wave_sum(b_lane);
        if (lane == 0) { msc[24 + wave] = e; msc[28 + wave] = b; }
        bc_bar();
        unsigned long long* pb = reinterpret_cast<unsigned long long*>(sv.part) + (nsync & 1) * 1024;
        const unsigned tag = mbase + (unsigned)nsync + 1u;
        if (tid == 0) {
            const float e4 = (msc[24] + msc[25]) + (msc[26] + msc[27]), b4 = (msc[28] + msc[29]) + (msc[30] + msc[31]);
            __hip_atomic_store(pb + 2 * blockIdx.x, ((unsigned long long)tag << 32) | __float_as_uint(e4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pb + 2 * blockIdx.x + 1, ((unsigned long long)tag << 32) | __float_as_uint(b4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float c0 = 0.f, c1 = 0.f;
        int ok = 1;
        if (tid < (int)gridDim.x) {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            const auto prs = __builtin_amdgcn_make_buffer_rsrc(pb, 0, 16 * 512, 0x00020000);
            ok = 0;
            const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
            for (int spin = 0; spin < sv.spin_limit; ++spin) {
                const u32x4_ wq = __builtin_bit_cast(u32x4_, __builtin_amdgcn_raw_buffer_load_b128(prs, 16 * tid, 0, 0x11));
                if (wq.y == tag && wq.w == tag) { c0 = __uint_as_float(wq.x); c1 = __uint_as_float(wq.z); ok = 1; break; }
                if ((spin & 63) == 63 && __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        c0 = bc_wave_sum(c0); c1 = bc_wave_sum(c1);
        const float bad = bc_wave_sum(ok ? 0.f : 1.f);
        bc_bar();                                          // (msc[24..31] were read by thread 0)
        if (lane == 0) { msc[24 + wave] = c0; msc[28 + wave] = c1; msc[32 + wave] = bad; }
        bc_bar();
        p0 = (msc[24] + msc[25]) + (msc[26] + msc[27]);
        p1 = (msc[28] + msc[29]) + (msc[30] + msc[31]);
        const float nbad = (msc[32] + msc[33]) + (msc[34] + msc[35]);
        bc_bar();
        ++nsync;
        return bc_uni(nbad) == 0.f;
    };
    auto after_ctrl = [&]() { hstep = bc_uni(ns.h); abstol = bc_uni(ns.abstol); reltol = bc_uni(ns.reltol); };
    auto add_norm = [&](float& acc, float u, float x) {
        const float sk = fmaf(fabsf(u), reltol, abstol);
        const float y = x / sk;
        acc = fmaf(y, y, acc);
    };
    const int my_s = tid & 7;                              // scalar owner threads: tid < 8 (sample tid)
    const bool s_live = tid < 8 && blockIdx.x * 8 + tid < a.B;

    // ---- inference_sol (src/base_icnf.jl:167-189) of one tile: logp(z) - dlogp, the regulariser rows, |z_aug|; the four loss sums
    // of the workgroup's tiles accumulate in v4 (owner threads tid < 8) ----
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
    auto post_tile = [&](const f32x4& uzt, float sc0, float sc1, float sc2, bool slive) __attribute__((always_inline)) {
        f32x4 ssc, sac;
#pragma unroll
        for (int s = 0; s < 4; ++s) { ssc[s] = uzt[s] * uzt[s]; sac[s] = ok_row >= sv.nvars ? uzt[s] * uzt[s] : 0.f; }
        bc_bar();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float x = bc_parity_sum(ssc[s]), y = bc_parity_sum(sac[s]);
            if (lane < 2) { red[wave][lane][0][s] = x; red[wave][lane][1][s] = y; }
        }
        bc_bar();
        if (slive) {
            const int g = tid >> 2, s = tid & 3;
            const float ss = (red[0][g][0][s] + red[1][g][0][s]) + (red[2][g][0][s] + red[3][g][0][s]);
            const float sa = (red[0][g][1][s] + red[1][g][1][s]) + (red[2][g][1][s] + red[3][g][1][s]);
            const float log2pi = 1.8378770664093453f;
            const float lp = -0.5f * fmaf((float)n_in, log2pi, ss) - sc0;
            const float aa = (sv.norm_z_aug && sv.naugs > 0) ? sqrtf(sa) : 0.f;
            const float Ev = TEST ? 0.f : sc1, Nv = TEST ? 0.f : sc2;
            const size_t bb = (size_t)tile * 8 + tid, Bz = (size_t)a.B;
            sv.logpx[bb] = lp; sv.regs[bb] = Ev; sv.regs[Bz + bb] = Nv; sv.regs[2 * Bz + bb] = aa;
            v4[0] += lp; v4[1] += Ev; v4[2] += Nv; v4[3] += aa;
        }
    };
    bool alive = true;
    if constexpr (!MULTI) {
        {
            // ---- k1 = f(u0); automatic initial dt ----
            float e = 0.f, b = 0.f;
            f32x4 k1;
            rhs(uz, k1);
            bc_bar();
            *(f32x4*)Kz[0][tid] = k1;
            if (tid < 8) for (int q = 0; q < NS; ++q) Ssc[tid][1][q] = msc[8 * q + my_s];
#pragma unroll
            for (int s = 0; s < 4; ++s) if (omask[s] != 0.f) { add_norm(e, uz[s], uz[s]); add_norm(b, uz[s], k1[s]); }
            if (s_live) for (int q = 0; q < NS; ++q) { add_norm(e, Ssc[tid][0][q], Ssc[tid][0][q]); add_norm(b, Ssc[tid][0][q], Ssc[tid][1][q]); }
            if (sv.hairer) alive = meet(e, b);
            if (sv.hairer && alive) {
                ctrl_phase(&ns, 0, p0, p1, a.n_total);
                after_ctrl();
                e = 0.f;
                f32x4 f1;
                rhs(uz + hstep * k1, f1);
                bc_bar();
#pragma unroll
                for (int s = 0; s < 4; ++s) if (omask[s] != 0.f) add_norm(e, uz[s], f1[s] - k1[s]);
                if (s_live) for (int q = 0; q < NS; ++q) add_norm(e, Ssc[tid][0][q], msc[8 * q + my_s] - Ssc[tid][1][q]);
                alive = meet(e, 0.f);
                if (alive) { ctrl_phase(&ns, 1, p0, p1, a.n_total); after_ctrl(); }
            }
        }
        constexpr float BT[7] = {TS_BT1, TS_BT2, TS_BT3, TS_BT4, TS_BT5, TS_BT6, TS_BT7};
        f32x4 zt = zero4;
        for (int it = 0; alive && !__builtin_amdgcn_readfirstlane(ns.done) && it < sv.maxiters; ++it) {
            float errsum = 0.f, badcnt = 0.f;
            // RECORD: where this attempt's stage states go (null: more steps than slots -- the host grows the store and solves again)
            float* dmp = nullptr;
            if (RECORD) {
                const int nacc = __builtin_amdgcn_readfirstlane(ns.naccept);
                if (nacc < a.dump_cap) {
                    dmp = a.dump + (size_t)nacc * a.dump_slot + (size_t)smp0 * D + ok_row;
                    if (blockIdx.x == 0 && tid == 0) a.hs_out[nacc] = hstep;
#pragma unroll
                    for (int x = 0; x < 4; ++x) if (omask[x] != 0.f) (dmp - a.dump_n)[(size_t)x * D] = uz[x];      // u_n = U_1
                }
            }
#pragma unroll 1
            for (int s = 1; s <= 6; ++s) {
                float as[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) as[j] = tab.a[s][j];
                f32x4 acc = as[0] * *(const f32x4*)Kz[0][tid];
#pragma unroll
                for (int j = 1; j < 6; ++j) acc += as[j] * *(const f32x4*)Kz[j][tid];     // (rows beyond the stage are zero or stale times a zero coefficient)
                zt = uz + hstep * acc;
                if (RECORD && dmp && s < 6) {                  // U_{s+1}
#pragma unroll
                    for (int x = 0; x < 4; ++x) if (omask[x] != 0.f) (dmp + (size_t)(s - 1) * a.dump_n)[(size_t)x * D] = zt[x];
                }
                f32x4 zd;
                rhs(zt, zd);
                bc_bar();
                *(f32x4*)Kz[s][tid] = zd;
                if (tid < 8) for (int q = 0; q < NS; ++q) Ssc[tid][1 + s][q] = msc[8 * q + my_s];
            }
            // the new solution is the last stage state (a_7j = b_j); error estimate
            {
                f32x4 ez = BT[0] * *(const f32x4*)Kz[0][tid];
#pragma unroll
                for (int j = 1; j < 7; ++j) ez += BT[j] * *(const f32x4*)Kz[j][tid];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (omask[s] != 0.f) {
                        const float scl = fmaf(fmaxf(fabsf(uz[s]), fabsf(zt[s])), reltol, abstol);
                        const float x = hstep * ez[s] / scl;
                        errsum = fmaf(x, x, errsum);
                        badcnt += !(fabsf(zt[s]) <= 3.0e38f) ? 1.f : 0.f;
                    }
            }
            float uns[3] = {0.f, 0.f, 0.f};
            if (tid < 8) {
                for (int q = 0; q < NS; ++q) {
                    float acc = 0.f, es = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc = fmaf(tab.a[6][j], Ssc[tid][1 + j][q], acc);
#pragma unroll
                    for (int j = 0; j < 7; ++j) es = fmaf(BT[j], Ssc[tid][1 + j][q], es);
                    const float us = Ssc[tid][0][q];
                    uns[q] = us + hstep * acc;
                    if (s_live) {
                        const float scl = fmaf(fmaxf(fabsf(us), fabsf(uns[q])), reltol, abstol);
                        const float x = hstep * es / scl;
                        errsum = fmaf(x, x, errsum);
                        badcnt += !(fabsf(uns[q]) <= 3.0e38f) ? 1.f : 0.f;
                    }
                }
            }
#ifdef BC_STAMPS
            bt_ = __builtin_amdgcn_s_memtime();
#endif
            alive = meet(errsum, badcnt);
            BC_T(7)
            if (!alive) break;
            const int acc0 = ns.naccept;
            const float t_att = ns.t, h_att = ns.h;
            ctrl_after_step(&ns, p0, p1, a.n_total);
            const bool accepted = __builtin_amdgcn_readfirstlane(ns.naccept != acc0);
            if (sv.trace && blockIdx.x == 0 && tid == 0 && it < sv.trace_cap) {
                float* tr = sv.trace + 4 * it;
                tr[0] = t_att; tr[1] = h_att; tr[2] = ns.eest; tr[3] = accepted ? 1.f : 0.f;
            }
            after_ctrl();
            if (accepted) {                                    // u <- u_new, k1 <- k7
                uz = zt;
                *(f32x4*)Kz[0][tid] = *(const f32x4*)Kz[6][tid];
                if (tid < 8) for (int q = 0; q < NS; ++q) { Ssc[tid][0][q] = uns[q]; Ssc[tid][1][q] = Ssc[tid][7][q]; }
            }
        }
        // ---- final state ----
        float* out = sv.u_out ? sv.u_out : a.U0;
        if (alive || !sv.u_out) {
#pragma unroll
            for (int s = 0; s < 4; ++s) if (omask[s] != 0.f) out[(size_t)(smp0 + s) * D + ok_row] = uz[s];
            if (s_live) for (int q = 0; q < NS; ++q) out[(size_t)(blockIdx.x * 8 + tid) * D + n_in + q] = Ssc[tid][0][q];
        }
        if (sv.logpx && alive) post_tile(uz, Ssc[my_s][0][0], Ssc[my_s][0][1], Ssc[my_s][0][2], s_live);
    } else {
        // ================= several tiles per workgroup: the Runge-Kutta rows in the store =================
        const int ntiles = (a.B + 7) / 8;
        f32x4* const rows = reinterpret_cast<f32x4*>(a.store);
        float* const gsc = a.store + (size_t)ntiles * BC_TROWS * 1024 + 32 * tid;      // + BC_TSC * tile: the scalar rows of sample `tid` (tid < 8)
        auto trow = [&](int r) -> f32x4* { return rows + ((size_t)tile * BC_TROWS + r) * 256 + tid; };
        auto tlive = [&]() { return tid < 8 && tile * 8 + tid < a.B; };
        int cur = 0;                                       // accepted u / k1 are rows BC_R_U + cur, BC_R_K1 + cur (scalars: rows cur, 2 + cur)
        {
            // ---- u0 and the probes into the store; k1 = f(u0); automatic initial dt ----
            float e = 0.f, b = 0.f;
            for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
                set_tile(t);
                load_u0();
                *trow(BC_R_U) = uz;
                if (!TEST) *trow(BC_R_EP) = ep;
                float* gs = gsc + (size_t)BC_TSC * t;
                float su[3] = {0.f, 0.f, 0.f};
                if (tid < 8) {
                    if (!sv.xs && tlive()) for (int q = 0; q < NS; ++q) su[q] = sv.u0[(size_t)(t * 8 + tid) * D + n_in + q];
#pragma unroll
                    for (int j = 0; j < 32; ++j) gs[j] = j < 3 ? su[j] : 0.f;
                }
                f32x4 k1;
                rhs(uz, k1);
                bc_bar();
                *trow(BC_R_K1) = k1;
#pragma unroll
                for (int s = 0; s < 4; ++s) if (omask[s] != 0.f) { add_norm(e, uz[s], uz[s]); add_norm(b, uz[s], k1[s]); }
                if (tid < 8) {
                    for (int q = 0; q < NS; ++q) {
                        const float kq = msc[8 * q + my_s];
                        gs[3 * 2 + q] = kq;
                        if (tlive()) { add_norm(e, su[q], su[q]); add_norm(b, su[q], kq); }
                    }
                }
            }
            if (sv.hairer) alive = meet(e, b);
            if (sv.hairer && alive) {
                ctrl_phase(&ns, 0, p0, p1, a.n_total);
                after_ctrl();
                e = 0.f;
                for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
                    set_tile(t);
                    uz = *trow(BC_R_U);
                    const f32x4 k1 = *trow(BC_R_K1);
                    if (!TEST) ep = *trow(BC_R_EP);
                    f32x4 f1;
                    rhs(uz + hstep * k1, f1);
                    bc_bar();
#pragma unroll
                    for (int s = 0; s < 4; ++s) if (omask[s] != 0.f) add_norm(e, uz[s], f1[s] - k1[s]);
                    if (tlive()) {
                        const float* gs = gsc + (size_t)BC_TSC * t;
                        for (int q = 0; q < NS; ++q) add_norm(e, gs[q], msc[8 * q + my_s] - gs[3 * 2 + q]);
                    }
                }
                alive = meet(e, 0.f);
                if (alive) { ctrl_phase(&ns, 1, p0, p1, a.n_total); after_ctrl(); }
            }
        }
        constexpr float BT[7] = {TS_BT1, TS_BT2, TS_BT3, TS_BT4, TS_BT5, TS_BT6, TS_BT7};
        for (int it = 0; alive && !__builtin_amdgcn_readfirstlane(ns.done) && it < sv.maxiters; ++it) {
            float errsum = 0.f, badcnt = 0.f;
#pragma unroll 1
            for (int s = 1; s <= 6; ++s) {
                float as[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) as[j] = tab.a[s][j];
                for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
                    set_tile(t);
                    // the rows of the stages done so far (k_j, j < s: row BC_R_K1 + cur, then BC_R_K2 ...; the others are not
                    // read: the store is not initialised), all requested at once
                    uz = *trow(BC_R_U + cur);
                    if (!TEST) ep = *trow(BC_R_EP);
                    f32x4 kk[6];
                    kk[0] = *trow(BC_R_K1 + cur);
#pragma unroll
                    for (int j = 1; j < 6; ++j) kk[j] = j < s ? *trow(BC_R_K2 + j - 1) : zero4;
                    f32x4 acc = as[0] * kk[0];
#pragma unroll
                    for (int j = 1; j < 6; ++j) acc += as[j] * kk[j];
                    const f32x4 zt = uz + hstep * acc;
                    f32x4 ezp = zero4;                     // the error estimate's share of k1..k6 (the last stage adds k7)
                    if (s == 6) {
                        ezp = BT[0] * kk[0];
#pragma unroll
                        for (int j = 1; j < 6; ++j) ezp += BT[j] * kk[j];
                        *trow(BC_R_U + (cur ^ 1)) = zt;    // the candidate solution is the last stage state (a_7j = b_j)
                    }
                    f32x4 zd;
                    rhs(zt, zd);
                    bc_bar();
                    *trow(s < 6 ? BC_R_K2 + s - 1 : BC_R_K1 + (cur ^ 1)) = zd;
                    float* gs = gsc + (size_t)BC_TSC * t;
                    if (tid < 8) for (int q = 0; q < NS; ++q) gs[3 * (s < 6 ? 3 + s : 2 + (cur ^ 1)) + q] = msc[8 * q + my_s];
                    if (s == 6) {
                        const f32x4 ez = ezp + BT[6] * zd;
#pragma unroll
                        for (int x = 0; x < 4; ++x)
                            if (omask[x] != 0.f) {
                                const float scl = fmaf(fmaxf(fabsf(uz[x]), fabsf(zt[x])), reltol, abstol);
                                const float y = hstep * ez[x] / scl;
                                errsum = fmaf(y, y, errsum);
                                badcnt += !(fabsf(zt[x]) <= 3.0e38f) ? 1.f : 0.f;
                            }
                        if (tid < 8) {
                            for (int q = 0; q < NS; ++q) {
                                float kq[7];
                                kq[0] = gs[3 * (2 + cur) + q];
#pragma unroll
                                for (int j = 1; j < 6; ++j) kq[j] = gs[3 * (3 + j) + q];
                                kq[6] = msc[8 * q + my_s];
                                float acq = 0.f, es = 0.f;
#pragma unroll
                                for (int j = 0; j < 6; ++j) acq = fmaf(tab.a[6][j], kq[j], acq);
#pragma unroll
                                for (int j = 0; j < 7; ++j) es = fmaf(BT[j], kq[j], es);
                                const float us = gs[3 * cur + q];
                                const float un = us + hstep * acq;
                                gs[3 * (cur ^ 1) + q] = un;
                                if (tlive()) {
                                    const float scl = fmaf(fmaxf(fabsf(us), fabsf(un)), reltol, abstol);
                                    const float y = hstep * es / scl;
                                    errsum = fmaf(y, y, errsum);
                                    badcnt += !(fabsf(un) <= 3.0e38f) ? 1.f : 0.f;
                                }
                            }
                        }
                    }
                }
            }
            alive = meet(errsum, badcnt);
            if (!alive) break;
            const int acc0 = ns.naccept;
            const float t_att = ns.t, h_att = ns.h;
            ctrl_after_step(&ns, p0, p1, a.n_total);
            const bool accepted = __builtin_amdgcn_readfirstlane(ns.naccept != acc0);
            if (sv.trace && blockIdx.x == 0 && tid == 0 && it < sv.trace_cap) {
                float* tr = sv.trace + 4 * it;
                tr[0] = t_att; tr[1] = h_att; tr[2] = ns.eest; tr[3] = accepted ? 1.f : 0.f;
            }
            after_ctrl();
            if (accepted) cur ^= 1;                        // u <- u_new, k1 <- k7: the other set of rows
        }
        // ---- final state, post-processing ----
        float* out = sv.u_out ? sv.u_out : a.U0;
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            set_tile(t);
            uz = *trow(BC_R_U + cur);
            const float* gs = gsc + (size_t)BC_TSC * t;
            float sc[3] = {0.f, 0.f, 0.f};
            if (tid < 8) for (int q = 0; q < NS; ++q) sc[q] = gs[3 * cur + q];
            if (alive || !sv.u_out) {
#pragma unroll
                for (int s = 0; s < 4; ++s) if (omask[s] != 0.f) out[(size_t)(smp0 + s) * D + ok_row] = uz[s];
                if (tlive()) for (int q = 0; q < NS; ++q) out[(size_t)(t * 8 + tid) * D + n_in + q] = sc[q];
            }
            if (sv.logpx && alive) post_tile(uz, sc[0], sc[1], sc[2], tlive());
        }
    }
    if (sv.logpx && alive) {
        if (sv.sums5) {
            unsigned long long* qb = reinterpret_cast<unsigned long long*>(sv.part) + 2048;
            const unsigned tag = mbase + (unsigned)nsync + 1u;
            if (wave == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = bc_wave_sum(v4[j]);
                if (lane < 4)
                    __hip_atomic_store(qb + 4 * blockIdx.x + lane, ((unsigned long long)tag << 32) | __float_as_uint(lane == 0 ? v4[0] : lane == 1 ? v4[1] : lane == 2 ? v4[2] : v4[3]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (blockIdx.x == 0) {
                float c4[4] = {0.f, 0.f, 0.f, 0.f};
                float late = 0.f;
                if (tid < (int)gridDim.x) {
                    const unsigned long long wait0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int got = 0;
                        for (int spin = 0; spin < sv.spin_limit; ++spin) {
                            const unsigned long long ww = __hip_atomic_load(qb + 4 * tid + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)(ww >> 32) == tag) { c4[j] = __uint_as_float((unsigned)ww); got = 1; break; }
                            if ((spin & 15) == 15 && __builtin_amdgcn_s_memrealtime() - wait0 > sv.wait_ticks) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (!got) late = 1.f;
                    }
                    if (late != 0.f) __hip_atomic_store(sv.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) c4[j] = bc_wave_sum(c4[j]);
                late = bc_wave_sum(late);
                bc_bar();
                if (lane == 0) { for (int j = 0; j < 4; ++j) msc[4 * wave + j] = c4[j]; msc[16 + wave] = late; }
                bc_bar();
                if (tid < 4) sv.sums5[tid] = (msc[tid] + msc[4 + tid]) + (msc[8 + tid] + msc[12 + tid]);
                if (tid == 0) {
                    sv.sums5[4] = (float)a.B;
                    if ((msc[16] + msc[17]) + (msc[18] + msc[19]) != 0.f) alive = false;
                }
            }
        }
    }
#ifdef BC_STAMPS
    if (sv.trace && blockIdx.x == 3 && tid == 64 && sv.trace_cap >= 8) for (int i = 0; i < 10; ++i) sv.trace[4 * 40 + i] = (float)bst[i];
#endif
    if (blockIdx.x == 0 && tid == 0) {
        if (!alive || __hip_atomic_load(sv.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ns.done = 0; ns.n_partials = -1; }
        __hip_atomic_store(sv.base_dev, mbase + (unsigned)nsync + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ns.cur = 0;
        *a.st_out = ns;
        if (sv.t_out) { sv.t_out[1] += __builtin_amdgcn_s_memrealtime() - sv.t_out[0]; sv.t_out[2] += 1; }
        mirror_store(a.mirror, a.seq, ns);
    }
}

}  // namespace

size_t bcast_img_floats() { return (size_t)BC_IMG_FLOATS; }

void bcast_pack(const NetDesc& nd, const float* d_params, float* d_img, hipStream_t s) {
    hipLaunchKernelGGL(k_bcast_pack, dim3((BC_IMG_FLOATS + 255) / 256), dim3(256), 0, s, nd, d_params, d_img);
}

// workgroups the device holds at once: one per CU (four waves of 512 registers)
// CNF_BCAST_FORCE_MULTI=1 (measurements): the several-tiles-per-workgroup form even when every tile has a CU of its own
static bool bcast_force_multi() { static const bool v = [] { const char* e = getenv("CNF_BCAST_FORCE_MULTI"); return e && e[0] == '1'; }(); return v; }
static int bcast_resident(int device) {
    constexpr int MAXDEV = 64;
    static std::mutex mu;
    static int res[MAXDEV];
    if (device < 0 || device >= MAXDEV) return 0;
    std::lock_guard<std::mutex> lk(mu);
    if (res[device] == 0) {
        int n_cu = 0;
        res[device] = (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) ? n_cu : -1;
        if (res[device] < 0) (void)hipGetLastError();
    }
    return res[device] > 0 ? res[device] : 0;
}

bool bcast_solve_supported(const NetDesc& nd, bool train, int B, int device) {
    if (nd.n_layers != 2 || nd.acts[0] != 1 || nd.acts[1] != 1) return false;
    if (nd.n_in <= 64 || nd.n_in > 128 || nd.dims[1] <= 256 || nd.dims[1] > 384) return false;
    static const bool off = [] { const char* e = getenv("CNF_PERSISTENT"); const char* w = getenv("CNF_BCAST"); return (e && e[0] == '0') || (w && w[0] == '0'); }();
    if (off || B < 1) return false;
    const int res = bcast_resident(device);
    if (res <= 0) return false;
    const bool multi = (B + 7) / 8 > res || bcast_force_multi();
    // conditional models: the per-sample first-layer bias is staged in LDS per tile -- one tile per workgroup, VJP / TestMode
    if (nd.n_cond > 0 && (multi || (train && nd.jvp))) return false;
    return true;
}

// floats of the tile store a launch needs (0: every workgroup owns one tile, the rows stay on the CU)
size_t bcast_store_floats(int B, int device) {
    const int res = bcast_resident(device), ntiles = (B + 7) / 8;
    if (res <= 0 || (ntiles <= res && !bcast_force_multi())) return 0;
    return (size_t)ntiles * (BC_TROWS * 1024 + BC_TSC);
}

cnf_status bcast_solve_launch(const NetDesc& nd, bool train, const float* d_params, const float* d_img, StepState* st_out, float* U0,
                              const float* eps, int B, hipStream_t s, void* mirror, unsigned seq, const Solve3Args& sv_, int device,
                              float* store, const float* cond, int cbs, const BcastRecord* rec) {
    if (!bcast_solve_supported(nd, train, B, device) || !d_img) return CNF_ERR_UNSUPPORTED;
    const int res = bcast_resident(device), ntiles = (B + 7) / 8;
    const bool multi = ntiles > res || bcast_force_multi();
    if (rec && (multi || !train)) return CNF_ERR_UNSUPPORTED;                // (the recording form: TrainMode, one tile per workgroup)
    if (multi && !store) return CNF_ERR_BAD_ARG;
    if (nd.n_cond > 0 && !cond) return CNF_ERR_BAD_ARG;
    BcArgs a{};
    a.nd = nd; a.P = d_params; a.img = d_img; a.eps = eps; a.B = B;
    a.n_total = (float)((size_t)(nd.n_in + (train ? 3 : 1)) * B);
    a.U0 = U0; a.st_out = st_out; a.mirror = mirror; a.seq = seq;
    a.store = store; a.cond = nd.n_cond > 0 ? cond : nullptr; a.cbs = cbs;
    if (rec) { a.dump = rec->dump; a.dump_n = rec->n; a.dump_slot = rec->slot; a.dump_cap = rec->cap; a.hs_out = rec->hs_out; }
    Solve3Args sv = sv_;
    sv.nvars = nd.nvars; sv.naugs = nd.naugs; sv.norm_z_aug = nd.norm_z_aug;
    if (!sv.xs && !sv.u0) return CNF_ERR_BAD_ARG;
    const int grid = multi ? (ntiles < res ? ntiles : res) : ntiles;
    if (multi) {                                           // (a wait lasts as long as the slowest workgroup's tiles take)
        const unsigned long long w = (unsigned long long)sv.wait_ticks * (unsigned)((ntiles + grid - 1) / grid);
        sv.wait_ticks = w > 2000000000ull ? 2000000000u : (unsigned)w;
    }
    BcTab tab = kBcTab;
    void* args[] = {&a, &sv, &tab};
    const void* fn;
    if (multi) fn = !train ? (const void*)k_solve_bcast<BC_TESTM, true> : (nd.jvp ? (const void*)k_solve_bcast<BC_JVP, true> : (const void*)k_solve_bcast<BC_VJP, true>);
    else fn = !train ? (const void*)k_solve_bcast<BC_TESTM, false> : (nd.jvp ? (const void*)k_solve_bcast<BC_JVP, false> : (const void*)k_solve_bcast<BC_VJP, false>);
    if (rec) fn = nd.jvp ? (const void*)k_solve_bcast<BC_JVP, false, true> : (const void*)k_solve_bcast<BC_VJP, false, true>;
    if (hipLaunchKernel(fn, dim3(grid), dim3(256), args, 0, s) != hipSuccess) {
        (void)hipGetLastError();
        return CNF_ERR_UNSUPPORTED;
    }
    return CNF_OK;
}
