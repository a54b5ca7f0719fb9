// Microbenchmark: fp32 MFMA throughput with the B operand fed from LDS and the A operand resident in registers (the
// shape of the step kernels' inner loops), v_mfma_f32_16x16x4_f32 against v_mfma_f32_32x32x2_f32, 1 or 2 waves per SIMD.
// Reports flops per REAL shader cycle per SIMD (s_memtime) and the clock (s_memtime / s_memrealtime): ideal = 64.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>       // 0: 16x16x4, two accumulator chains; 1: 32x32x2, two accumulator chains
__global__ void __launch_bounds__(512) k(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = f32x4{0.5f + u, 0.25f, 2.f, 1.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 0) {
        const float* xb = lds + (lane & 15) * 136 + 4 * (lane >> 4);
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
        for (int it = 0; it < iters; ++it) {
            f32x4 b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = *(const f32x4*)(xb + 16 * u + ((it & 1) ? 2176 : 0));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, c1, 0, 0, 0);
            }
        }
        lds[threadIdx.x] = c0.x + c1.y;
    } else {
        const float* xb = lds + (lane & 31) * 136 + 4 * (lane >> 5);
        f32x16 c0 = {}, c1 = {};
        for (int it = 0; it < iters; ++it) {
            f32x4 b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = *(const f32x4*)(xb + 8 * u + ((it & 1) ? 4352 : 0));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, c1, 0, 0, 0);
            }
        }
        lds[threadIdx.x] = c0[0] + c1[5];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 5 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <class K>
static void run(const char* name, K kk, int threads, unsigned long long* d, double flops_per_mfma) {
    const int iters = 4000;
    (void)hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipLaunchKernelGGL(kk, dim3(256), dim3(threads), 140000, 0, d, 10);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kk, dim3(256), dim3(threads), 140000, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double mfmas_per_simd = (double)iters * 32 * (threads / 256);
    printf("%-34s %d wave(s)/SIMD: %6.2f flops/cycle/SIMD (ideal 64), %5.1f cycles per MFMA, clock %4.0f MHz; events: %.3f ms = %.1f TFLOP/s\n",
           name, threads / 256, mfmas_per_simd * flops_per_mfma / (double)h[0], (double)h[0] / mfmas_per_simd,
           (double)h[0] / (double)h[1] * 100.0, ms, mfmas_per_simd * flops_per_mfma * 1024.0 / (ms * 1e9));
}
int main() {
    unsigned long long* d; (void)hipMalloc(&d, 64);
    for (int t : {256, 512}) {
        run("16x16x4, B from LDS, A resident", k<0>, t, d, 2048.0);
        run("32x32x2, B from LDS, A resident", k<1>, t, d, 4096.0);
    }
    return 0;
}
