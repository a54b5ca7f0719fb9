// The resource split VERDICT round 3 (item 4) asked to be tried for the headline kernel: 4 waves x 512 registers per CU on
// v_mfma_f32_32x32x16_bf16 instead of k_solve3b's 8 waves x 256 registers on v_mfma_f32_16x16x32_bf16 -- as a timing model
// of ONE evaluation of the 32-128-128-32 network (split-bf16 products, six terms each), before committing to a rewrite of
// the 2 900-line kernel.
//
// Structure modelled (32 samples per workgroup = the N of a 32x32 tile; wave w owns feature rows 32 w .. 32 w + 31):
//   * the accumulator tile of a layer is the B operand of the next product with no lane movement (MFMA guide, "an
//     accumulator tile as the next MFMA's operand"), so a wave multiplies its OWN 32 rows of h (K = 32, in registers, split
//     in registers) against the matching 32 columns of the next weight matrix for all 128 output rows: no operand images in
//     LDS, no operand reads;  what crosses waves are fp32 PARTIAL output tiles (reduce-scatter through LDS: each wave writes
//     three 32 x 32 tiles and reads three);
//   * wide products (128 -> 128, both sweeps): 4 output tiles x 2 k-steps x 6 terms = 48 MFMAs per wave + reduce-scatter;
//     32 -> 128: output-stationary, 12 MFMAs, no reduction;  128 -> 32: K-split, 12 MFMAs + an all-reduce of one tile;
//     144 MFMAs x 32 cycles = 4.6 k cycles of matrix pipe per evaluation and SIMD: the same as k_solve3b's 288 x 16;
//   * epilogue per layer on the wave's own 32 x 32 tile: 16 values per lane: bias, tanh (exp2 / rcp), sigma', the three-piece
//     split (cvt_pk_bf16 pairs) straight into the next B operand registers.
// Timed: cycles per evaluation of this skeleton on random data, every CU busy (256 workgroups), against k_solve3b's measured
// 11.3-11.5 k cycles per evaluation.  The numerics are NOT checked here (weights and data are random, the k
// permutation of the accumulator-as-operand trick is not applied): a timing model, nothing else.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/eval3w.hip -o /tmp/e3w && /tmp/e3w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Op { bf16x8 h, m, l; };                              // one k-step (16 k's: 8 per lane half) of a split operand

__device__ __forceinline__ float tanh_fast(float a) {
    const float t = __builtin_amdgcn_exp2f(a * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}
// v = h + m + l with h, m, l bf16 (round to nearest): two values per conversion
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf2 hh = __builtin_convertvector(f2{a, b}, bf2);
    const f2 hf = __builtin_convertvector(hh, f2);
    const f2 r1 = f2{a, b} - hf;
    const bf2 mm = __builtin_convertvector(r1, bf2);
    const f2 r2 = r1 - __builtin_convertvector(mm, f2);
    const bf2 ll = __builtin_convertvector(r2, bf2);
    h = __builtin_bit_cast(unsigned, hh); m = __builtin_bit_cast(unsigned, mm); l = __builtin_bit_cast(unsigned, ll);
}
// registers 8 s .. 8 s + 7 of an accumulator tile -> the split operand of k-step s
__device__ __forceinline__ Op split8(const f32x16& x, int s) {
    u32x4 h, m, l;
    unsigned a, b, c;
    split2(x[8 * s + 0], x[8 * s + 1], a, b, c); h.x = a; m.x = b; l.x = c;
    split2(x[8 * s + 2], x[8 * s + 3], a, b, c); h.y = a; m.y = b; l.y = c;
    split2(x[8 * s + 4], x[8 * s + 5], a, b, c); h.z = a; m.z = b; l.z = c;
    split2(x[8 * s + 6], x[8 * s + 7], a, b, c); h.w = a; m.w = b; l.w = c;
    Op o; o.h = __builtin_bit_cast(bf16x8, h); o.m = __builtin_bit_cast(bf16x8, m); o.l = __builtin_bit_cast(bf16x8, l);
    return o;
}
__device__ __forceinline__ f32x16 mm6(const Op& a, const Op& b, f32x16 c) {      // six terms, smallest first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// MODE 0: the whole skeleton; 1: without the MFMAs; 2: without the epilogue arithmetic; 3: without the LDS reductions
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_eval(const float* __restrict__ wsrc, float* out, int evals, unsigned long long* stamp) {
    __shared__ __attribute__((aligned(16))) float part[2][4][4][16][64];        // [buffer][source wave][tile][register][lane]: 2 x 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // resident split weight fragments: wide layer forward and reverse (4 tiles x 2 k-steps each), four narrow products (2 k-steps)
    Op W2[4][2], W2T[4][2], N1[2], N2[2], N3[2], N4[2];
    {
        const float* p = wsrc + (size_t)(wave * 64 + lane) * 64;
        f32x16 r;
        for (int i = 0; i < 16; ++i) r[i] = p[i];
        for (int t = 0; t < 4; ++t) for (int s = 0; s < 2; ++s) { W2[t][s] = split8(r, s); W2T[t][s] = split8(r * 0.5f, s); r = r * 1.01f; }
        for (int s = 0; s < 2; ++s) { N1[s] = split8(r, s); N2[s] = split8(r * 0.3f, s); N3[s] = split8(r * 0.7f, s); N4[s] = split8(r * 0.9f, s); }
    }
    f32x16 zero;
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    f32x16 x = zero;
    for (int i = 0; i < 16; ++i) x[i] = 0.01f * (lane + i);
    auto epilogue = [&](f32x16 v, Op (&o)[2]) {
        if (MODE != 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = tanh_fast(v[i] + 0.1f);
        }
        o[0] = split8(v, 0); o[1] = split8(v, 1);
        return v;
    };
    int pbuf = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    Op b[2];
    b[0] = split8(x, 0); b[1] = split8(x, 1);
    for (int e = 0; e < evals; ++e) {
#pragma unroll
        for (int sweep = 0; sweep < 2; ++sweep) {          // forward, reverse: the same three product shapes
            // ---- narrow -> wide (32 -> 128): output-stationary, K = 32 ----
            f32x16 a1 = zero;
            if (MODE != 1) { a1 = mm6(sweep ? N3[0] : N1[0], b[0], a1); a1 = mm6(sweep ? N3[1] : N1[1], b[1], a1); }
            else a1 = x;
            Op h1[2];
            x = epilogue(a1, h1);
            // ---- wide (128 -> 128): K-split over the waves, four partial tiles, reduce-scatter ----
            f32x16 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = zero;
                if (MODE != 1) { acc[t] = mm6(sweep ? W2T[t][0] : W2[t][0], h1[0], acc[t]); acc[t] = mm6(sweep ? W2T[t][1] : W2[t][1], h1[1], acc[t]); }
                else acc[t] = x * (1.f + t);
            }
            f32x16 own = acc[0];
            if (MODE != 3) {
                // the partial of tile (wave + t) & 3 goes to that wave's inbox (the own tile, t = 0, stays in registers)
                // (conflict-free form: [register quad][lane][4])
#pragma unroll
                for (int t = 1; t < 4; ++t) {
                    const int dst = (wave + t) & 3;
                    const f32x16 v = t == 1 ? acc[1] : (t == 2 ? acc[2] : acc[3]);
                    float* base = &part[pbuf][dst][wave][0][0];
#pragma unroll
                    for (int i = 0; i < 4; ++i) *(f32x4*)(base + (i * 64 + lane) * 4) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                }
                bar();
#pragma unroll
                for (int src = 0; src < 4; ++src)
                    if (src != wave) {
                        const float* base = &part[pbuf][wave][src][0][0];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 v = *(const f32x4*)(base + (i * 64 + lane) * 4);
                            own[4 * i] += v.x; own[4 * i + 1] += v.y; own[4 * i + 2] += v.z; own[4 * i + 3] += v.w;
                        }
                    }
                pbuf ^= 1;
            }
            Op h2[2];
            x = epilogue(own, h2);
            // ---- wide -> narrow (128 -> 32): K-split, one partial tile per wave, all-reduce ----
            f32x16 a3 = zero;
            if (MODE != 1) { a3 = mm6(sweep ? N4[0] : N2[0], h2[0], a3); a3 = mm6(sweep ? N4[1] : N2[1], h2[1], a3); }
            else a3 = x;
            if (MODE != 3) {
                float* base = &part[pbuf][0][wave][0][0];
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(base + (i * 64 + lane) * 4) = f32x4{a3[4 * i], a3[4 * i + 1], a3[4 * i + 2], a3[4 * i + 3]};
                bar();
#pragma unroll
                for (int src = 0; src < 4; ++src)
                    if (src != wave) {
                        const float* bs = &part[pbuf][0][src][0][0];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 v = *(const f32x4*)(bs + (i * 64 + lane) * 4);
                            a3[4 * i] += v.x; a3[4 * i + 1] += v.y; a3[4 * i + 2] += v.z; a3[4 * i + 3] += v.w;
                        }
                    }
                pbuf ^= 1;
            }
            x = epilogue(a3, b);                            // (forward: zdot and the seed of the reverse sweep; reverse: eps^T J and the next stage state)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) { stamp[2 * blockIdx.x] = t1 - t0; stamp[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
static void run(const char* what, const float* dw, float* dout, unsigned long long* dst, int ncu) {
    const int evals = 400;
    hipLaunchKernelGGL((k_eval<MODE>), dim3(ncu), dim3(256), 0, 0, dw, dout, 4, dst);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k_eval<MODE>), dim3(ncu), dim3(256), 0, 0, dw, dout, evals, dst);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> st(2 * ncu);
    (void)hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < ncu; ++i) { cyc += st[2 * i]; rt += st[2 * i + 1]; }
    printf("%-44s %7.0f cycles per evaluation (%.2f GHz)\n", what, cyc / ncu / evals, cyc / rt / 10.0);
}

int main() {
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<float> h(256 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.02f * (float)((i * 37) % 101) - 1.f;
    float *dw, *dout; unsigned long long* dst;
    (void)hipMalloc(&dw, h.size() * 4); (void)hipMalloc(&dout, (size_t)ncu * 256 * 4); (void)hipMalloc(&dst, (size_t)ncu * 16);
    (void)hipMemcpy(dw, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("4 waves x 512 registers, v_mfma_f32_32x32x16_bf16, 32 samples per workgroup; matrix pipe alone: 144 x 32 = 4608 cycles\n");
    run<0>("the skeleton", dw, dout, dst, ncu);
    run<1>("without the MFMAs", dw, dout, dst, ncu);
    run<2>("without tanh (split kept)", dw, dout, dst, ncu);
    run<3>("without the LDS reductions and barriers", dw, dout, dst, ncu);
    printf("k_solve3b (8 waves x 256 registers, 16x16x32): 11.3-11.5 k cycles per evaluation (DESIGN 7.0)\n");
    return 0;
}
