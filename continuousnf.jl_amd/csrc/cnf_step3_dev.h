// Device helpers shared by the step / solve kernels of the headline shape (cnf_step3.hip): the Tsit5 table
// as a kernel argument, the LDS-only barrier, wave reductions, the global weight image of the split kernels and the
// three-piece bf16 operand algebra (split, images, six-term products).
#pragma once
#include "cnf_step3.h"
#include "cnf_split.h"

// Tsit5 rows a_{s+1, 1..6} (s = 1..6), passed BY VALUE as a kernel argument: the stage sums then take their coefficients
// by scalar loads from the argument segment, which every wave has just read (scalar-cache hits).  A __constant__ table
// costs a scalar-cache miss -- a memory round trip -- in front of the first stage of every launch, and select chains
// over immediates put the 21 literals into vector registers.
struct S3Tab { float a[7][8]; };
static const S3Tab kS3Tab = {{
    {0, 0, 0, 0, 0, 0, 0, 0},
    {TS_A21, 0, 0, 0, 0, 0, 0, 0},
    {TS_A31, TS_A32, 0, 0, 0, 0, 0, 0},
    {TS_A41, TS_A42, TS_A43, 0, 0, 0, 0, 0},
    {TS_A51, TS_A52, TS_A53, TS_A54, 0, 0, 0, 0},
    {TS_A61, TS_A62, TS_A63, TS_A64, TS_A65, 0, 0, 0},
    {TS_A71, TS_A72, TS_A73, TS_A74, TS_A75, TS_A76, 0, 0}}};

#define S3_SB() __builtin_amdgcn_sched_barrier(0)
// workgroup barrier for LDS traffic only: does not drain global loads / stores in flight
// (the scheduling fences keep register-only instructions -- MFMAs -- in the interval the source puts them in)
__device__ __forceinline__ void s3_bar() {
    __builtin_amdgcn_sched_barrier(0);
#ifdef S3_ABL_NOBAR
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ f32x4 s3_tanh4(const f32x4& a) {
#ifdef S3_ABL_NOEPI
    return a * 0.25f;
#endif
    return f32x4{tanh_fast(a.x), tanh_fast(a.y), tanh_fast(a.z), tanh_fast(a.w)};
}
__device__ __forceinline__ f32x4 s3_dtanh4(const f32x4& h) {       // sigma' from h
    return f32x4{fmaf(-h.x, h.x, 1.f), fmaf(-h.y, h.y, 1.f), fmaf(-h.z, h.z, 1.f), fmaf(-h.w, h.w, 1.f)};
}
__device__ __forceinline__ float s3_dot4(const f32x4& a, const f32x4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
// Sum over the 64 lanes of the wave, fixed tree, on the VALU: four DPP butterflies inside each row of 16 lanes (quad
// swaps, half-row mirror, row mirror: every lane ends with its row's total), then the four row totals through
// v_readlane.  The __shfl_down ladder does the same through the LDS crossbar: six dependent round trips per value.
__device__ __forceinline__ float s3_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}

// Global image of the two split kernels (bytes).  The register-resident fragments travel in fp32 and are split on arrival
// (2/3 of the bytes of three bf16 pieces; the split runs while the rest of the stream is in flight).  Fragment = the 8
// weights M[16 tile + x][32 k-block + 8q .. +7] of lane 16q + x, as two 16-byte halves: [half 2][lane 64] x 16 B.
namespace s3g {
constexpr int BIASB = 0;                                   // b1 (128), b2 (128), b3 (32) fp32
constexpr int F32 = (2 * 128 + 32) * 4;                    // [wave 8][W1 | W2 x4 | W3^T | W2^T x4], then W3: [tile 2][k-block 4]
constexpr int NFR = 8 * 10 + 2 * 4;
constexpr int W3I = F32 + NFR * 2048;                      // k_step3b's LDS images (three bf16 pieces, LDS layout): W3 rows, W1^T rows
constexpr int WI = 3 * 32 * 256;
constexpr int IMG_BYTES = W3I + 2 * WI;
static_assert(F32 % 16 == 0 && W3I % 16 == 0, "16-byte loads");
}  // namespace s3g

// 4 rows of one sample -> the three images (8 bytes each)
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s3b_store4(char* img, int piece_bytes, const f32x4& v) {
#ifdef S3_ABL_NOSTORE
    { asm volatile("" :: "v"(v), "v"((unsigned)(size_t)img)); return; }
#endif
    u32x2_ h, m, l;
    { unsigned h_, m_, l_; s3b_split2(v[0], v[1], h_, m_, l_); h.x = h_; m.x = m_; l.x = l_; }
    { unsigned h_, m_, l_; s3b_split2(v[2], v[3], h_, m_, l_); h.y = h_; m.y = m_; l.y = l_; }
    *(u32x2_*)img = h; *(u32x2_*)(img + piece_bytes) = m; *(u32x2_*)(img + 2 * piece_bytes) = l;
}
struct S3bOp { bf16x8 h, m, l; };
__device__ __forceinline__ S3bOp s3b_load(const char* img, int piece_bytes) {
    S3bOp o;
#ifdef S3_ABL_NOLOAD
    { typedef unsigned u4_ __attribute__((ext_vector_type(4))); u4_ z = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(z) : "v"((unsigned)(size_t)img));
      o.h = o.m = o.l = __builtin_bit_cast(bf16x8, z); return o; }
#endif
    o.h = *(const bf16x8*)img; o.m = *(const bf16x8*)(img + piece_bytes); o.l = *(const bf16x8*)(img + 2 * piece_bytes);
    return o;
}
// 8 consecutive fp32 values -> a split operand
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ S3bOp s3b_split8(const f32x4& lo4, const f32x4& hi4) {
    u32x4 h, m, l;
    { unsigned h_, m_, l_; s3b_split2(lo4[0], lo4[1], h_, m_, l_); h.x = h_; m.x = m_; l.x = l_; }
    { unsigned h_, m_, l_; s3b_split2(lo4[2], lo4[3], h_, m_, l_); h.y = h_; m.y = m_; l.y = l_; }
    { unsigned h_, m_, l_; s3b_split2(hi4[0], hi4[1], h_, m_, l_); h.z = h_; m.z = m_; l.z = l_; }
    { unsigned h_, m_, l_; s3b_split2(hi4[2], hi4[3], h_, m_, l_); h.w = h_; m.w = m_; l.w = l_; }
    S3bOp o;
    o.h = __builtin_bit_cast(bf16x8, h); o.m = __builtin_bit_cast(bf16x8, m); o.l = __builtin_bit_cast(bf16x8, l);
    return o;
}
// an ordered no-op that consumes and redefines the operand: pins its computation in program order
__device__ __forceinline__ void s3b_pin(S3bOp& o) {
    u32x4 a = __builtin_bit_cast(u32x4, o.h), b = __builtin_bit_cast(u32x4, o.m), c = __builtin_bit_cast(u32x4, o.l);
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    o.h = __builtin_bit_cast(bf16x8, a); o.m = __builtin_bit_cast(bf16x8, b); o.l = __builtin_bit_cast(bf16x8, c);
}
// term T (0..5, smallest first) of the product a x b into acc
template <int T>
__device__ __forceinline__ f32x4 s3b_term(const S3bOp& a, const S3bOp& b, const f32x4& acc) {
#ifdef S3_ABL_NOMFMA
    { u32x4 x = __builtin_bit_cast(u32x4, a.h), y = __builtin_bit_cast(u32x4, b.h); asm volatile("" :: "v"(x), "v"(y)); return acc; }
#endif
    if (T == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.l, b.h, acc, 0, 0, 0);
    if (T == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.l, acc, 0, 0, 0);
    if (T == 2) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.m, acc, 0, 0, 0);
    if (T == 3) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.m, b.h, acc, 0, 0, 0);
    if (T == 4) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.m, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, acc, 0, 0, 0);
}
// NC independent products that share the A operand: term-major order, so that the six terms of one accumulator are NC
// MFMAs apart
template <int NC>
__device__ __forceinline__ void s3b_mm(f32x4 (&acc)[NC], const S3bOp& a, const S3bOp (&b)[NC]) {
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = s3b_term<0>(a, b[n], acc[n]);
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = s3b_term<1>(a, b[n], acc[n]);
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = s3b_term<2>(a, b[n], acc[n]);
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = s3b_term<3>(a, b[n], acc[n]);
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = s3b_term<4>(a, b[n], acc[n]);
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[n] = s3b_term<5>(a, b[n], acc[n]);
}

// 4 rows of one sample back from the three images: the pieces sum to the fp32 value exactly
__device__ __forceinline__ f32x4 s3b_load4(const char* img, int piece_bytes) {
    const bf16x4 h = *(const bf16x4*)img, m = *(const bf16x4*)(img + piece_bytes), l = *(const bf16x4*)(img + 2 * piece_bytes);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)h[j] + ((float)m[j] + (float)l[j]);
    return v;
}

