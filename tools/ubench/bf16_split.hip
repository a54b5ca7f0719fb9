// Microbenchmark for DESIGN.md section 8.6: an fp32 product emulated by six bf16 MFMA terms (operands split into three
// bf16 pieces each: exact), against v_mfma_f32_16x16x4_f32 -- accuracy against a float64 reference and issue rate.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/bf16_split.hip -o /tmp/bf16_split && /tmp/bf16_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // pieces by truncation (the upper half of an fp32 is a bf16), residuals by exact subtractions: as the kernels do
        const unsigned hb = __float_as_uint(x[j]) & 0xFFFF0000u;
        const float r1 = x[j] - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(mb);
        hi[j] = __builtin_bit_cast(__bf16, (unsigned short)(hb >> 16));
        mid[j] = __builtin_bit_cast(__bf16, (unsigned short)(mb >> 16));
        lo[j] = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(r2) >> 16));
    }
}

// one 16x16 tile, K = 128: A[16][128], B[128][16] -> C fp32 (both ways) ; lane (x = l & 15, q = l >> 4)
__global__ void k_acc(const float* A, const float* B, float* Cf32, float* Csplit) {
    const int l = threadIdx.x, x = l & 15, q = l >> 4;
    f32x4 c32 = {0, 0, 0, 0}, cs = {0, 0, 0, 0};
    for (int k0 = 0; k0 < 128; k0 += 32) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = A[x * 128 + k0 + 8 * q + j]; b[j] = B[(k0 + 8 * q + j) * 16 + x]; }
        // fp32 path: 16x16x4 takes k = 4 * step + q; feed the same 32 k in 8 steps
        for (int st = 0; st < 8; ++st) {
            const float av = A[x * 128 + k0 + 4 * st + q], bv = B[(k0 + 4 * st + q) * 16 + x];
            c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c32, 0, 0, 0);
        }
        bf16x8 ah, am, al, bh, bm, bl;
        split3(a, ah, am, al); split3(b, bh, bm, bl);
        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, cs, 0, 0, 0);      // smallest terms first
        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, cs, 0, 0, 0);
        cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, cs, 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) { Cf32[(4 * q + j) * 16 + x] = c32[j]; Csplit[(4 * q + j) * 16 + x] = cs[j]; }
}

template <int MODE>
__global__ void __launch_bounds__(512) k_rate(float* out, int iters) {
    const int l = threadIdx.x & 63;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    bf16x8 p, r;
    for (int j = 0; j < 8; ++j) { p[j] = (__bf16)(0.001f * (l + j)); r[j] = (__bf16)(0.5f + j); }
    const float fa = 0.001f * l, fb = 0.5f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {         // 8 fp32 MFMAs = K 32 of one tile
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, c1, 0, 0, 0);
            }
        } else {                 // 6 bf16 MFMAs = the same K 32 of one tile
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, r, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r, p, c1, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y;
}

int main() {
    std::vector<float> A(16 * 128), B(128 * 16), c32(256), cs(256);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    float *dA, *dB, *d32, *ds, *dout;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&d32, 1024); (void)hipMalloc(&ds, 1024);
    (void)hipMalloc(&dout, 256 * 512 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, d32, ds);
    (void)hipMemcpy(c32.data(), d32, 1024, hipMemcpyDeviceToHost);
    (void)hipMemcpy(cs.data(), ds, 1024, hipMemcpyDeviceToHost);
    double e32 = 0, es = 0, mag = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0, m = 0;
            for (int k = 0; k < 128; ++k) { ref += (double)A[i * 128 + k] * B[k * 16 + j]; m += std::fabs((double)A[i * 128 + k] * B[k * 16 + j]); }
            e32 = std::fmax(e32, std::fabs(c32[i * 16 + j] - ref) / m);
            es = std::fmax(es, std::fabs(cs[i * 16 + j] - ref) / m);
            mag = std::fmax(mag, m);
        }
    printf("max |error| / sum|a b| over a 16x16 tile, K = 128:  fp32 MFMA %.2e,  six bf16 terms %.2e\n", e32, es);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 20000;
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(512), 0, 0, dout, 10);
        else hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(512), 0, 0, dout, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(512), 0, 0, dout, iters);
        else hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(512), 0, 0, dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        // per iteration and wave: one 16x16 tile x K 32 = 16384 fp32-equivalent flops
        const double eq = (double)iters * 16384.0 * 8 * 256;
        printf("%s: %.3f ms -> %.1f fp32-equivalent TFLOP/s\n", mode == 0 ? "8 x v_mfma_f32_16x16x4_f32   " : "6 x v_mfma_f32_16x16x32_bf16 ", ms, eq / (ms * 1e9));
    }
    return 0;
}
