// Device helpers shared by the step kernels of libcnfhip (cnf_mfma.hip: k_mfma; cnf_step3.hip: k_step3).
#pragma once
#include "cnf_mfma.h"
#include "cnf_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MfmaArgs {
    int mode;                 // 0: plain RHS (u -> du), 1: probe f(u + h k1) -> Ks[0], 2: Tsit5 step
    int B;
    const float* img;         // weight+bias image
    const float* eps;         // n_in x B
    const float* u;           // mode 0
    float* du;                // mode 0
    const StepState* st;      // mode 1, 2: state to run from (mode 2 + apply_ctrl: state BEFORE the controller)
    StepState* st_out;        // mode 2 + apply_ctrl: where block 0 stores the state after the controller
    const float* partials_in; // mode 2 + apply_ctrl: error partials of the previous attempt
    int apply_ctrl;
    float n_total;            // D * B
    const float* cond;        // conditional models: per-sample first-layer bias [B][cbs], else null
    int cbs;
    int test;                 // TestMode: exact trace (2-layer closed form), state rows = n_in + 1
    const float* cimg;        // TestMode: row-major image of C = W_1 .* W_2^T  (P1 x SWC)
    int SWC;
    float* U[2];
    float* K1[2];
    float* Ks0;               // mode 1 output
    float* partials;          // mode 2: 2 floats per workgroup
    int init_phase;           // modes 0/1 inside a solve: also produce the norm partials of initial-dt phase 0/1 and let
    unsigned* ticket;         //   the last workgroup to finish run that controller phase on *st_out (-1: off)
    void* mirror;             // streamed solve: pinned host mirror of the state after each controller run (granules
                              //   {launch index, word}: cnf_mirror.h)
    unsigned seq;
    float* dump;              // mode 2, gradient path: z rows of the stage states U_2..U_6 go to dump + (stage - 2) * dump_stride
    size_t dump_stride;       //   ([B][D] arrays like the state), else null
    size_t dump_step_stride;  // != 0: trajectory store indexed on the device -- this attempt files into the slot of step
    int dump_cap;             //   `naccept` (u_n one array before `dump`, then U_2..U_6), if naccept < dump_cap, and its
    float* hs_out;            //   signed step size into hs_out[naccept]
};


__device__ __forceinline__ float quad_sum(float v) {     // sum over the 4 lanes l, l^16, l^32, l^48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Tsit5 stage combination sum_j a_{S+1,j} k_j (S = 1..6) on 4 rows at once
template <int S>
__device__ __forceinline__ f32x4 stage_acc4(const f32x4 (&k)[7]) {
    constexpr float A[7][6] = {
        {0, 0, 0, 0, 0, 0},
        {TS_A21, 0, 0, 0, 0, 0},
        {TS_A31, TS_A32, 0, 0, 0, 0},
        {TS_A41, TS_A42, TS_A43, 0, 0, 0},
        {TS_A51, TS_A52, TS_A53, TS_A54, 0, 0},
        {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65, 0},
        {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76}};
    f32x4 acc = A[S][0] * k[0];
#pragma unroll
    for (int j = 1; j < S; ++j) acc += A[S][j] * k[j];
    return acc;
}
__device__ __forceinline__ f32x4 stage_acc4_rt(int stg, const f32x4 (&k)[7]) {
    switch (stg) {
        case 1: return stage_acc4<1>(k);
        case 2: return stage_acc4<2>(k);
        case 3: return stage_acc4<3>(k);
        case 4: return stage_acc4<4>(k);
        case 5: return stage_acc4<5>(k);
        default: return stage_acc4<6>(k);
    }
}
__device__ __forceinline__ void set_k(f32x4 (&k)[7], int idx, const f32x4& v) {
    // select per slot: keeps every k[i] in registers (a switch turns into an indexed store
    // and sends the array to scratch)
#pragma unroll
    for (int i = 1; i < 7; ++i) k[i] = idx == i ? v : k[i];
}
__device__ __forceinline__ f32x4 ld4(const float* p, int nvalid4) {   // rows beyond n_in read as 0
    // one branch for "nothing valid" (never touches memory then); otherwise branch-free:
    // out-of-range elements re-read element 0 and are zeroed, so the four loads issue back to back
    if (nvalid4 <= 0) return f32x4{0.f, 0.f, 0.f, 0.f};
    const int i1 = nvalid4 > 1 ? 1 : 0, i2 = nvalid4 > 2 ? 2 : 0, i3 = nvalid4 > 3 ? 3 : 0;
    const float v0 = p[0], v1 = p[i1], v2 = p[i2], v3 = p[i3];
    return f32x4{nvalid4 > 0 ? v0 : 0.f, nvalid4 > 1 ? v1 : 0.f, nvalid4 > 2 ? v2 : 0.f, nvalid4 > 3 ? v3 : 0.f};
}
// Branch-free variant for the kernel prologue: `ld4_issue` only issues the four loads (always from valid
// addresses: `safe` stands in when there is nothing to read), `ld4_mask` zeroes what was not asked for.
// A branch around a load makes the compiler wait for it at once; issuing all prologue loads first and
// masking afterwards keeps ~30 loads in flight instead of 8 serial groups of 4.
__device__ __forceinline__ f32x4 ld4_issue(const float* p, int nvalid4, const float* safe) {
    const float* q = nvalid4 > 0 ? p : safe;
    const int i1 = nvalid4 > 1 ? 1 : 0, i2 = nvalid4 > 2 ? 2 : 0, i3 = nvalid4 > 3 ? 3 : 0;
    return f32x4{q[0], q[i1], q[i2], q[i3]};
}
// One instruction instead of four when every lane's count is <= 0 or >= 4 (`wide`, wave-uniform: the rows come in whole
// groups of four) -- the sample rows are only 4-byte aligned (D = n_in + 3 floats apart), which global memory allows.
// The texture-address path handles a wave's 64 lane addresses per instruction at a fixed rate whatever the width, so the
// prologue's ~36 dword gathers per wave cost it four times the cycles of 9 wide ones.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ f32x4 ld4_issue_w(const float* p, int nvalid4, const float* safe, bool wide) {
    if (wide) return *reinterpret_cast<const f32x4u*>(nvalid4 > 0 ? p : safe);
    return ld4_issue(p, nvalid4, safe);
}
__device__ __forceinline__ f32x4 ld3_issue(const float* p, int nvalid, const float* safe) {   // nvalid is 0 or 3
    const f32x3u v = *reinterpret_cast<const f32x3u*>(nvalid > 0 ? p : safe);
    return f32x4{v.x, v.y, v.z, 0.f};
}
__device__ __forceinline__ void st4_wide(float* p, const f32x4& v) { *reinterpret_cast<f32x4u*>(p) = v; }
__device__ __forceinline__ f32x4 ld4_mask(const f32x4& v, int nvalid4) {
    return f32x4{nvalid4 > 0 ? v.x : 0.f, nvalid4 > 1 ? v.y : 0.f, nvalid4 > 2 ? v.z : 0.f, nvalid4 > 3 ? v.w : 0.f};
}
__device__ __forceinline__ void st4(float* p, const f32x4& v, int nvalid4) {
    if (nvalid4 > 0) p[0] = v.x;
    if (nvalid4 > 1) p[1] = v.y;
    if (nvalid4 > 2) p[2] = v.z;
    if (nvalid4 > 3) p[3] = v.w;
}
__device__ __forceinline__ void err_acc(float& errsum, float& badcnt, const f32x4 (&k)[7], const f32x4& u,
                                        const f32x4& un, float h, float abstol, float reltol, int nvalid4) {
    f32x4 e = TS_BT1 * k[0] + TS_BT2 * k[1] + TS_BT3 * k[2] + TS_BT4 * k[3] + TS_BT5 * k[4] + TS_BT6 * k[5] +
              TS_BT7 * k[6];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < nvalid4) {
            const float sc = fmaf(fmaxf(fabsf(u[c]), fabsf(un[c])), reltol, abstol);
            const float x = h * e[c] / sc;
            errsum = fmaf(x, x, errsum);
            if (!(fabsf(un[c]) <= 3.0e38f)) badcnt += 1.f;
        }
    }
}

__device__ __forceinline__ void publish_mirror(const MfmaArgs& a, const StepState& z) { mirror_store(a.mirror, a.seq, z); }
