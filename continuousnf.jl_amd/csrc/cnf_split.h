// The exact three-piece bf16 split behind the six-term products (cnf_step3.hip, cnf_grad.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// v = h + m + l exactly: pieces by TRUNCATION (one AND each: the upper 16 bits of an fp32 are a bf16), residuals by exact
// subtractions; the last residual has at most 8 significant bits, so its upper half is all of it.  Round 2's form, kept
// for -DCNF_SPLIT_TRUNC builds: the kernels split pairs by round-to-nearest now (s3b_split2 below), which is as cheap
// and keeps the dropped terms below 2^-23 |a b| in the worst case instead of 2^-21.
__device__ __forceinline__ void s3b_split(float v, __bf16& h, __bf16& m, __bf16& l) {
    const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb);
    h = __builtin_bit_cast(__bf16, (unsigned short)(hb >> 16));
    m = __builtin_bit_cast(__bf16, (unsigned short)(mb >> 16));
    l = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(r2) >> 16));
}

// The same split by ROUND-TO-NEAREST pieces, two values at a time (v_cvt_pk_bf16_f32 converts a pair and packs it):
// |m| <= 2^-8 ulp-units, |l| <= 2^-17, so the three products a six-term product leaves out (m l, l m, l l) stay below
// 2^-24 |a b| for EVERY input, where truncated pieces reach 2^-21 |a b| when every mantissa bit is set
// (tests/test_gpu_parity.py::test_split_product_error_bound).  Still exact: v - h has at most 16 significant bits,
// (v - h) - m at most 8.  Same instruction count as the truncating form (per pair: 3 converts, 4 widenings, 4 subtractions
// against 4 ANDs, 4 subtractions, 3 packs).  -DCNF_SPLIT_TRUNC selects the truncating form (A/B).
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s3b_split2(float v0, float v1, unsigned& h2, unsigned& m2, unsigned& l2) {
#ifdef S3_ABL_NOEPI
    h2 = (__float_as_uint(v0) >> 16) | (__float_as_uint(v1) & 0xFFFF0000u); m2 = 0u; l2 = 0u; return;
#endif
#ifdef CNF_SPLIT_TRUNC
    __bf16 a0, b0, c0, a1, b1, c1;
    s3b_split(v0, a0, b0, c0); s3b_split(v1, a1, b1, c1);
    h2 = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
    m2 = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
    l2 = (unsigned)__builtin_bit_cast(unsigned short, c0) | ((unsigned)__builtin_bit_cast(unsigned short, c1) << 16);
#else
    h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{v0, v1}, bf16x2_));
    const float r0 = v0 - __uint_as_float(h2 << 16), r1 = v1 - __uint_as_float(h2 & 0xFFFF0000u);
    m2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{r0, r1}, bf16x2_));
    const float q0 = r0 - __uint_as_float(m2 << 16), q1 = r1 - __uint_as_float(m2 & 0xFFFF0000u);
    l2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_{q0, q1}, bf16x2_));
#endif
}
