// Timing model: one evaluation of the 16-48-16 network (BASELINE config 2, VJP mode) by ONE wave per 16-sample tile -- what
// k_solve_wave does: 48 v_mfma_f32_16x16x4_f32 and 16 tanh per lane in one dependent chain, 3.37 k cycles measured in the
// kernel -- against the same evaluation by THREE waves of a workgroup (one per SIMD), wave w owning hidden tile w: 16 MFMAs and
// 8 tanh per wave, the two K = 48 products as per-wave partial tiles summed through LDS (one b128 per lane written, a
// barrier, three read).  Random data, the numerics are not checked: cycles per evaluation only, every CU busy.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/quadwave.hip -o /tmp/qw && /tmp/qw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tanh_fast(float a) {
    const float t = __builtin_amdgcn_exp2f(a * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}
__device__ __forceinline__ f32x4 mm4(const float (&A)[4], const f32x4& b, f32x4 acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], b[j], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ float quad_sum(float v) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ void act(const f32x4& pre, f32x4& h, f32x4& d) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = tanh_fast(pre[j]); d[j] = fmaf(-h[j], h[j], 1.f); }
}

// ---- one wave per tile: the evaluation of k_solve_wave<1, 3, VJP> ----
__global__ void __launch_bounds__(64) k_single(const float* W, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x;
    float fW1[3][4], fW2[3][4], fW2T[3][4], fW1T[3][4];
    for (int m = 0; m < 3; ++m)
        for (int j = 0; j < 4; ++j) {
            fW1[m][j] = W[(m * 4 + j) * 64 + lane]; fW2[m][j] = W[768 + (m * 4 + j) * 64 + lane];
            fW2T[m][j] = W[1536 + (m * 4 + j) * 64 + lane]; fW1T[m][j] = W[2304 + (m * 4 + j) * 64 + lane];
        }
    f32x4 z = {0.1f * lane, 0.2f, -0.1f, 0.05f}, ep = {0.3f, -0.2f, 0.1f, 0.7f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float sacc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 h1[3], d1[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { f32x4 a = mm4(fW1[m], z, zero4); act(a, h1[m], d1[m]); }
        f32x4 part[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) part[k] = mm4(fW2[k], h1[k], zero4);
        f32x4 zd, d2;
        act(part[0] + part[1] + part[2], zd, d2);
        const f32x4 g2 = ep * d2;
        f32x4 g1[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) g1[m] = mm4(fW2T[m], g2, zero4) * d1[m];
#pragma unroll
        for (int k = 0; k < 3; ++k) part[k] = mm4(fW1T[k], g1[k], zero4);
        const f32x4 eJ = part[0] + part[1] + part[2];
        float ld = 0.f, n2 = 0.f, e2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ld = fmaf(eJ[j], ep[j], ld); n2 = fmaf(eJ[j], eJ[j], n2); e2 = fmaf(zd[j], zd[j], e2); }
        ld = quad_sum(ld); n2 = quad_sum(n2); e2 = quad_sum(e2);
        sacc += ld + __builtin_amdgcn_sqrtf(n2) + __builtin_amdgcn_sqrtf(e2);
        z = z + 0.01f * zd;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 64 + lane] = z[0] + sacc;
}

// ---- three waves per tile: wave w owns hidden tile w ----
__global__ void __launch_bounds__(192) k_tri(const float* W, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) f32x4 red[2][3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float fW1[4], fW2[4], fW2T[4], fW1T[4];
    for (int j = 0; j < 4; ++j) {
        fW1[j] = W[(w * 4 + j) * 64 + lane]; fW2[j] = W[768 + (w * 4 + j) * 64 + lane];
        fW2T[j] = W[1536 + (w * 4 + j) * 64 + lane]; fW1T[j] = W[2304 + (w * 4 + j) * 64 + lane];
    }
    f32x4 z = {0.1f * lane, 0.2f, -0.1f, 0.05f}, ep = {0.3f, -0.2f, 0.1f, 0.7f};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float sacc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 h1, d1;
        act(mm4(fW1, z, zero4), h1, d1);
        red[0][w][lane] = mm4(fW2, h1, zero4);
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
        f32x4 zd, d2;
        act(red[0][0][lane] + red[0][1][lane] + red[0][2][lane], zd, d2);
        const f32x4 g1 = mm4(fW2T, ep * d2, zero4) * d1;
        red[1][w][lane] = mm4(fW1T, g1, zero4);
        __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier();
        const f32x4 eJ = red[1][0][lane] + red[1][1][lane] + red[1][2][lane];
        float ld = 0.f, n2 = 0.f, e2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { ld = fmaf(eJ[j], ep[j], ld); n2 = fmaf(eJ[j], eJ[j], n2); e2 = fmaf(zd[j], zd[j], e2); }
        ld = quad_sum(ld); n2 = quad_sum(n2); e2 = quad_sum(e2);
        sacc += ld + __builtin_amdgcn_sqrtf(n2) + __builtin_amdgcn_sqrtf(e2);
        z = z + 0.01f * zd;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 192 + threadIdx.x] = z[0] + sacc;
}

int main() {
    std::vector<float> hW(3072);
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = 0.05f * (float)((int)(i * 2654435761u % 2001) - 1000) / 1000.f;
    float *W, *out; unsigned long long* cyc;
    hipMalloc(&W, hW.size() * 4); hipMalloc(&out, 256 * 192 * 4); hipMalloc(&cyc, 8);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    unsigned long long c = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_single, dim3(256), dim3(64), 0, 0, W, out, cyc, iters);
        hipDeviceSynchronize(); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("one wave per tile   : %7.0f cycles per evaluation (s_memtime)\n", (double)c / iters);
        hipLaunchKernelGGL(k_tri, dim3(256), dim3(192), 0, 0, W, out, cyc, iters);
        hipDeviceSynchronize(); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("three waves per tile: %7.0f cycles per evaluation\n", (double)c / iters);
    }
    return 0;
}
