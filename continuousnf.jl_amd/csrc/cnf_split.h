// The exact three-piece bf16 split behind the six-term products (cnf_step3.hip, cnf_grad.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// v = h + m + l exactly: pieces by TRUNCATION (one AND each: the upper 16 bits of an fp32 are a bf16), residuals by exact
// subtractions; the last residual has at most 8 significant bits, so its upper half is all of it.  (Round-to-nearest
// pieces cost a convert and a widen each and buy nothing: the three products left out are below 2^-24 either way.)
__device__ __forceinline__ void s3b_split(float v, __bf16& h, __bf16& m, __bf16& l) {
    const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb);
    h = __builtin_bit_cast(__bf16, (unsigned short)(hb >> 16));
    m = __builtin_bit_cast(__bf16, (unsigned short)(mb >> 16));
    l = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(r2) >> 16));
}
